// Single-process, multi-device evaluation (include/gprhip.h: gprhip_ctx_*, gprhip_sharded_*).
//
// The reference's host is one OCaml process (bin/ocaml_gpr.ml:176-177, :340-342), so the library itself shards the
// training points over the devices it is given and does the exchange steps: one gprhip_problem per device, one host
// worker thread per device that enqueues that shard's passes on its own HIP stream, and between the passes one grouped
// ncclAllReduce(sum, fp64) of the packed exchange buffers ON THOSE SAME STREAMS -- the collective is stream-ordered
// behind pass 1 and in front of pass 2 on every device, the host never waits in between.  RCCL is loaded with dlopen at
// context creation (only when more than one device is named), so libgprhip.so itself does not link it.
//
// Row partition, exchange buffers and the replicated m x m work are exactly those of the one-process-per-GPU path
// (gpr_amd/dist.py): SURVEY.md section 8(e).
#include <dlfcn.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/gprhip.h"
#include "common.h"
#include "problem_internal.h"

namespace gprhip {
const std::string& last_error();
}
using namespace gprhip;

namespace {

// ---- RCCL through dlopen: the handful of entry points the exchange step needs (rccl.h: ncclDouble = 8, ncclSum = 0)
typedef void* rccl_comm_t;
constexpr int RCCL_DOUBLE = 8, RCCL_SUM = 0;
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*CommGetAsyncError)(rccl_comm_t, int*) = nullptr;  // optional (absent from very old builds): checked after a group
  std::string path;
};

void fail(int status, const std::string& msg) {
  set_error(msg);
  throw HipFail{status};
}

void load_rccl(Rccl& r) {
  std::vector<std::string> names;
  if (const char* e = getenv("GPRHIP_RCCL_LIB")) names.push_back(e);
  names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"});
  std::string tried;
  for (const auto& n : names) {
    r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (r.handle) {
      r.path = n;
      break;
    }
    tried += (tried.empty() ? "" : ", ") + n;
  }
  if (!r.handle) fail(GPRHIP_ECOMM, "gprhip_ctx_create: RCCL not found (tried " + tried + "): " + dlerror());
  auto sym = [&](const char* name) {
    void* f = dlsym(r.handle, name);
    if (!f) fail(GPRHIP_ECOMM, std::string("gprhip_ctx_create: ") + r.path + " lacks " + name);
    return f;
  };
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
  r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(r.handle, "ncclCommGetAsyncError"));
}

// ---- one worker thread per device: runs the stage calls of that device's shard so that the shards' launches are
// enqueued side by side (a single host thread would start device k's pass k enqueue-times late, and the exchange step
// makes every device wait for the last one)
class Worker {
 public:
  Worker() : th_([this] { loop(); }) {}
  ~Worker() {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  void post(std::function<int()> f) {
    {
      std::lock_guard<std::mutex> g(mu_);
      task_ = std::move(f);
      busy_ = true;
    }
    cv_.notify_all();
  }
  // status of the posted call; its message (the worker's thread-local gprhip_last_error) in *msg
  int wait(std::string* msg) {
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [this] { return !busy_; });
    if (msg) *msg = msg_;
    return status_;
  }

 private:
  void loop() {
    for (;;) {
      std::function<int()> f;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return quit_ || (busy_ && task_); });
        if (quit_) return;
        f = std::move(task_);
        task_ = nullptr;
      }
      int st = GPRHIP_EHIP;
      std::string m;
      try {
        st = f();
        if (st != GPRHIP_OK) m = last_error();
      } catch (...) {
        m = "gprhip: unexpected C++ exception in a device worker";
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        status_ = st;
        msg_ = m;
        busy_ = false;
      }
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::function<int()> task_;
  bool busy_ = false, quit_ = false;
  int status_ = GPRHIP_OK;
  std::string msg_;
  std::thread th_;
};

constexpr int MAX_SHARDS = 64;
struct SumPtrs {
  double* b[MAX_SHARDS];
};
// validation mode (all shards on one device): every buffer <- b[0] + b[1] + ... in that fixed order
__global__ __launch_bounds__(256) void sum_buffers_kernel(SumPtrs ptrs, int nb, int64_t len) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  double s = ptrs.b[0][i];
  for (int k = 1; k < nb; ++k) s += ptrs.b[k][i];
  for (int k = 0; k < nb; ++k) ptrs.b[k][i] = s;
}

}  // namespace

struct gprhip_ctx {
  std::vector<int> devices;
  int mode = GPRHIP_COMM_NONE;
  Rccl rccl;
  std::vector<rccl_comm_t> comms;
  std::vector<std::unique_ptr<Worker>> workers;
  // lifetime bookkeeping, guarded by g_lifetime_mu: a garbage-collected host may finalise a sharded problem and its
  // context on different threads (OCaml 5 domains), in either order
  int live_problems = 0;
  bool closed = false;  // gprhip_ctx_destroy was called while sharded problems were alive: the last of them frees the context
};

struct gprhip_sharded {
  gprhip_ctx* ctx = nullptr;
  int64_t n = 0;
  std::vector<gprhip_problem*> parts;
  std::vector<int64_t> lo, hi;
  // same-device mode: one event per shard ("my pass is enqueued") and one for the summed buffer
  std::vector<hipEvent_t> ev_ready;
  hipEvent_t ev_summed = nullptr;
  int timing = 0;
  hipEvent_t t0[2] = {nullptr, nullptr}, t1[2] = {nullptr, nullptr};
  int collectives = 0;
  int64_t bytes[2] = {0, 0};
  float ms[2] = {0.f, 0.f};
};

namespace {

std::mutex g_lifetime_mu;  // guards gprhip_ctx::live_problems / closed of every context

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return GPRHIP_OK;
  } catch (const HipFail& e) {
    return e.status;
  } catch (const std::bad_alloc&) {
    set_error("gprhip: host allocation failed");
    return GPRHIP_EOOM;
  } catch (...) {
    set_error("gprhip: unexpected C++ exception");
    return GPRHIP_EHIP;
  }
}

// run f(i) for every shard -- on the shard's worker thread when there are several -- and fail with the first error
void for_all(gprhip_sharded* sp, const std::function<int(int)>& f) {
  const int nd = (int)sp->parts.size();
  if (nd == 1) {
    const int st = f(0);
    if (st != GPRHIP_OK) throw HipFail{st};  // message already in this thread's slot
    return;
  }
  for (int i = 0; i < nd; ++i) sp->ctx->workers[i]->post([&f, i] { return f(i); });
  int first = GPRHIP_OK;
  std::string msg, m;
  for (int i = 0; i < nd; ++i) {
    const int st = sp->ctx->workers[i]->wait(&m);
    if (st != GPRHIP_OK && first == GPRHIP_OK) {
      first = st;
      msg = m + " (shard " + std::to_string(i) + " on device " + std::to_string(sp->ctx->devices[i]) + ")";
    }
  }
  if (first != GPRHIP_OK) fail(first, msg);
}

// The exchange step: every shard's buffer <- sum over shards, stream-ordered on the shards' own streams.
void exchange(gprhip_sharded* sp, int which, int64_t len, const std::function<double*(gprhip_problem*)>& buf) {
  gprhip_ctx* c = sp->ctx;
  const int nd = (int)sp->parts.size();
  if (c->mode == GPRHIP_COMM_NONE) return;
  sp->bytes[which] = len * (int64_t)sizeof(double);
  hipStream_t s0 = problem_hip_stream(sp->parts[0]);
  if (sp->timing) {
    GPR_HIP(hipSetDevice(c->devices[0]));
    GPR_HIP(hipEventRecord(sp->t0[which], s0));
  }
  if (c->mode == GPRHIP_COMM_RCCL) {
    int rc = c->rccl.GroupStart();
    bool dev_ok = true;
    for (int i = 0; i < nd && rc == 0 && dev_ok; ++i) {
      dev_ok = hipSetDevice(c->devices[i]) == hipSuccess;  // (no exception between GroupStart and GroupEnd)
      if (!dev_ok) break;
      double* b = buf(sp->parts[i]);
      rc = c->rccl.AllReduce(b, b, (size_t)len, RCCL_DOUBLE, RCCL_SUM, c->comms[i], problem_hip_stream(sp->parts[i]));
    }
    const int rc2 = c->rccl.GroupEnd();
    if (rc == 0) rc = rc2;
    if (!dev_ok) fail(GPRHIP_EHIP, "gprhip_sharded_eval: hipSetDevice failed inside the exchange step");
    if (rc != 0) fail(GPRHIP_ECOMM, std::string("gprhip_sharded_eval: ncclAllReduce failed: ") + c->rccl.GetErrorString(rc));
    // a rank that failed asynchronously (a lost peer, a fault inside the collective's kernel) is reported by its
    // communicator, not by the enqueue calls above: surface it here as GPRHIP_ECOMM instead of a hang in the next
    // stream synchronisation
    if (c->rccl.CommGetAsyncError)
      for (int i = 0; i < nd; ++i) {
        int async = 0;
        const int q = c->rccl.CommGetAsyncError(c->comms[i], &async);
        if (q != 0 || async != 0)
          fail(GPRHIP_ECOMM, "gprhip_sharded_eval: RCCL reports an asynchronous error on rank " + std::to_string(i) + " (device " +
                                 std::to_string(c->devices[i]) + ") after exchange step " + std::to_string(which + 1) + ": " +
                                 c->rccl.GetErrorString(q != 0 ? q : async));
      }
  } else {  // all shards on one device: a fixed-order sum on the first shard's stream, fenced by events
    GPR_HIP(hipSetDevice(c->devices[0]));
    SumPtrs ptrs;
    for (int i = 0; i < nd; ++i) {
      ptrs.b[i] = buf(sp->parts[i]);
      if (i > 0) {
        GPR_HIP(hipEventRecord(sp->ev_ready[i], problem_hip_stream(sp->parts[i])));
        GPR_HIP(hipStreamWaitEvent(s0, sp->ev_ready[i], 0));
      }
    }
    hipLaunchKernelGGL(sum_buffers_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s0, ptrs, nd, len);
    GPR_HIP(hipGetLastError());
    GPR_HIP(hipEventRecord(sp->ev_summed, s0));
    for (int i = 1; i < nd; ++i) GPR_HIP(hipStreamWaitEvent(problem_hip_stream(sp->parts[i]), sp->ev_summed, 0));
  }
  if (sp->timing) {
    GPR_HIP(hipSetDevice(c->devices[0]));
    GPR_HIP(hipEventRecord(sp->t1[which], s0));
  }
  ++sp->collectives;
}

}  // namespace

extern "C" {

int gprhip_shard_rows(int64_t n, int ndev, int idx, int64_t* row_lo, int64_t* row_hi) {
  return guarded([&] {
    if (n < 1 || ndev < 1 || idx < 0 || idx >= ndev || !row_lo || !row_hi) fail(GPRHIP_EBADARG, "gprhip_shard_rows: invalid arguments");
    // contiguous blocks whose sizes differ by at most one (the same rule as gpr_amd.dist.shard_rows)
    const int64_t base = n / ndev, rem = n % ndev;
    *row_lo = idx * base + std::min<int64_t>(idx, rem);
    *row_hi = *row_lo + base + (idx < rem ? 1 : 0);
  });
}

int gprhip_ctx_create(const int* devices, int ndev, gprhip_ctx** out) {
  return guarded([&] {
    if (!out || !devices || ndev < 1 || ndev > MAX_SHARDS) fail(GPRHIP_EBADARG, "gprhip_ctx_create: invalid arguments (1 <= ndev <= 64)");
    *out = nullptr;
    check_single_hip_runtime("gprhip_ctx_create");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
    if (count < 1) fail(GPRHIP_EHIP, "gprhip_ctx_create: no HIP device visible");
    bool all_same = true, distinct = true;
    for (int i = 0; i < ndev; ++i) {
      if (devices[i] < 0 || devices[i] >= count)
        fail(GPRHIP_EBADARG, "gprhip_ctx_create: device " + std::to_string(devices[i]) + " does not exist (" +
                                 std::to_string(count) + " visible)");
      if (devices[i] != devices[0]) all_same = false;
      for (int j = 0; j < i; ++j)
        if (devices[j] == devices[i]) distinct = false;
    }
    if (ndev > 1 && !all_same && !distinct)
      fail(GPRHIP_EBADARG, "gprhip_ctx_create: devices must be all distinct (RCCL) or all the same one (validation mode)");
    std::unique_ptr<gprhip_ctx> c(new gprhip_ctx());
    c->devices.assign(devices, devices + ndev);
    const char* force = getenv("GPRHIP_CTX_RCCL");
    if (ndev > 1 && all_same) c->mode = GPRHIP_COMM_SAME_DEVICE;
    else if (ndev > 1 || (force && atoi(force) != 0)) c->mode = GPRHIP_COMM_RCCL;
    if (c->mode == GPRHIP_COMM_RCCL) {
      load_rccl(c->rccl);
      c->comms.assign(ndev, nullptr);
      const int rc = c->rccl.CommInitAll(c->comms.data(), ndev, c->devices.data());
      if (rc != 0) {
        c->comms.clear();
        fail(GPRHIP_ECOMM, std::string("gprhip_ctx_create: ncclCommInitAll failed: ") + c->rccl.GetErrorString(rc));
      }
    }
    if (ndev > 1)
      for (int i = 0; i < ndev; ++i) c->workers.emplace_back(new Worker());
    *out = c.release();
  });
}

static void ctx_free(gprhip_ctx* c) {
  c->workers.clear();
  for (size_t i = 0; i < c->comms.size(); ++i)
    if (c->comms[i]) {
      hipSetDevice(c->devices[i]);
      c->rccl.CommDestroy(c->comms[i]);
    }
  // the RCCL handle stays loaded: its teardown at dlclose is not safe while HIP streams it used are alive
  delete c;
}

// Sharded problems keep their context alive: destroying the context first (a host whose garbage collector finalises
// the two handles in either order) only marks it, and the last sharded problem frees it.
void gprhip_ctx_destroy(gprhip_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> g(g_lifetime_mu);
    if (c->closed) return;  // (a second destroy of a context that is waiting for its problems)
    if (c->live_problems > 0) {
      c->closed = true;
      return;
    }
    c->closed = true;
  }
  ctx_free(c);
}

int gprhip_ctx_ndev(const gprhip_ctx* c) { return c ? (int)c->devices.size() : 0; }
int gprhip_ctx_comm_mode(const gprhip_ctx* c) { return c ? c->mode : GPRHIP_COMM_NONE; }

int gprhip_sharded_create(gprhip_ctx* c, int cov_kind, int precision, int64_t n, int D, int d, int m, int64_t chunk_rows,
                          gprhip_sharded** out) {
  return guarded([&] {
    if (!c || !out) fail(GPRHIP_EBADARG, "gprhip_sharded_create: NULL argument");
    *out = nullptr;
    {  // the problem counts as alive from here on, so that a concurrent gprhip_ctx_destroy cannot free the context under it
      std::lock_guard<std::mutex> g(g_lifetime_mu);
      if (c->closed) fail(GPRHIP_EBADARG, "gprhip_sharded_create: destroyed context");
      ++c->live_problems;
    }
    struct Reserve {
      gprhip_ctx* c;
      bool keep = false;
      ~Reserve() {
        if (keep) return;
        bool last;
        {
          std::lock_guard<std::mutex> g(g_lifetime_mu);
          last = --c->live_problems == 0 && c->closed;
        }
        if (last) ctx_free(c);
      }
    } reserve{c};
    const int nd = (int)c->devices.size();
    if (n < nd) fail(GPRHIP_EBADARG, "gprhip_sharded_create: fewer training points than devices");
    std::unique_ptr<gprhip_sharded> sp(new gprhip_sharded());
    sp->ctx = c;
    sp->n = n;
    sp->parts.assign(nd, nullptr);
    sp->lo.resize(nd);
    sp->hi.resize(nd);
    for (int i = 0; i < nd; ++i)
      if (gprhip_shard_rows(n, nd, i, &sp->lo[i], &sp->hi[i]) != GPRHIP_OK) throw HipFail{GPRHIP_EBADARG};
    {  // every shard's resident set against the free memory of its device (shards sharing a device: their sum), before
       // the first shard allocates anything: a partition that cannot hold the problem fails here with the figures
      std::vector<int64_t> need(nd, 0);
      for (int i = 0; i < nd; ++i) {
        gprhip_memory_plan_t plan;
        const int st = gprhip_memory_plan(cov_kind, precision, sp->hi[i] - sp->lo[i], D, d, m, chunk_rows, &plan);
        if (st != GPRHIP_OK) throw HipFail{st};
        int first = i;
        for (int j = 0; j < i; ++j)
          if (c->devices[j] == c->devices[i]) { first = j; break; }
        need[first] += plan.total;
      }
      for (int i = 0; i < nd; ++i) {
        if (need[i] == 0) continue;
        size_t free_b = 0, total_b = 0;
        GPR_HIP(hipSetDevice(c->devices[i]));
        GPR_HIP(hipMemGetInfo(&free_b, &total_b));
        if ((uint64_t)need[i] > (uint64_t)free_b) {
          char buf[256];
          snprintf(buf, sizeof buf, "gprhip_sharded_create: the shards on device %d need %.1f GB (n = %lld rows over %d shards, m = %d) "
                   "and %.1f GB are free of %.1f GB: use more devices or the fp32-bulk mode", c->devices[i], need[i] / 1e9,
                   (long long)n, nd, m, free_b / 1e9, total_b / 1e9);
          fail(GPRHIP_EOOM, buf);
        }
      }
    }
    auto destroy_parts = [&] {
      for (auto*& q : sp->parts) {
        gprhip_problem_destroy(q);
        q = nullptr;
      }
    };
    for (int i = 0; i < nd; ++i) {
      const int st = gprhip_problem_create_ex(c->devices[i], cov_kind, precision, sp->hi[i] - sp->lo[i], D, d, m,
                                              chunk_rows, &sp->parts[i]);
      if (st != GPRHIP_OK) {
        destroy_parts();
        throw HipFail{st};
      }
    }
    try {
      GPR_HIP(hipSetDevice(c->devices[0]));
      if (c->mode == GPRHIP_COMM_SAME_DEVICE) {
        sp->ev_ready.assign(nd, nullptr);
        for (int i = 1; i < nd; ++i) GPR_HIP(hipEventCreateWithFlags(&sp->ev_ready[i], hipEventDisableTiming));
        GPR_HIP(hipEventCreateWithFlags(&sp->ev_summed, hipEventDisableTiming));
      }
      for (int k = 0; k < 2; ++k) {
        GPR_HIP(hipEventCreate(&sp->t0[k]));
        GPR_HIP(hipEventCreate(&sp->t1[k]));
      }
    } catch (...) {
      destroy_parts();
      throw;
    }
    reserve.keep = true;
    *out = sp.release();
  });
}

void gprhip_sharded_destroy(gprhip_sharded* sp) {
  if (!sp) return;
  for (auto* q : sp->parts) gprhip_problem_destroy(q);
  hipSetDevice(sp->ctx->devices[0]);
  for (auto e : sp->ev_ready)
    if (e) hipEventDestroy(e);
  if (sp->ev_summed) hipEventDestroy(sp->ev_summed);
  for (int k = 0; k < 2; ++k) {
    if (sp->t0[k]) hipEventDestroy(sp->t0[k]);
    if (sp->t1[k]) hipEventDestroy(sp->t1[k]);
  }
  gprhip_ctx* const c = sp->ctx;
  delete sp;
  bool last;
  {
    std::lock_guard<std::mutex> g(g_lifetime_mu);
    last = --c->live_problems == 0 && c->closed;
  }
  if (last) ctx_free(c);
}

int gprhip_sharded_shard(const gprhip_sharded* sp, int idx, int* device, int64_t* row_lo, int64_t* row_hi) {
  return guarded([&] {
    if (!sp || idx < 0 || idx >= (int)sp->parts.size()) fail(GPRHIP_EBADARG, "gprhip_sharded_shard: invalid arguments");
    if (device) *device = sp->ctx->devices[idx];
    if (row_lo) *row_lo = sp->lo[idx];
    if (row_hi) *row_hi = sp->hi[idx];
  });
}

gprhip_problem* gprhip_sharded_problem(gprhip_sharded* sp, int idx) {
  return (sp && idx >= 0 && idx < (int)sp->parts.size()) ? sp->parts[idx] : nullptr;
}

int gprhip_sharded_set_inputs(gprhip_sharded* sp, const double* inputs, int64_t ld) {
  return guarded([&] {
    if (!sp || !inputs) fail(GPRHIP_EBADARG, "gprhip_sharded_set_inputs: invalid arguments");
    for_all(sp, [&](int i) { return gprhip_set_inputs(sp->parts[i], inputs + sp->lo[i] * ld, ld); });
  });
}

int gprhip_sharded_set_targets(gprhip_sharded* sp, const double* targets) {
  return guarded([&] {
    if (!sp || !targets) fail(GPRHIP_EBADARG, "gprhip_sharded_set_targets: invalid arguments");
    for_all(sp, [&](int i) { return gprhip_set_targets(sp->parts[i], targets + sp->lo[i]); });
  });
}

int gprhip_sharded_eval(gprhip_sharded* sp, const gprhip_hypers* h, int want_grad, gprhip_result* res, double* grad,
                        double* coeffs) {
  return guarded([&] {
    if (!sp || !res || (want_grad && !grad)) fail(GPRHIP_EBADARG, "gprhip_sharded_eval: NULL argument");
    const int nd = (int)sp->parts.size();
    sp->collectives = 0;
    sp->bytes[0] = sp->bytes[1] = 0;
    sp->ms[0] = sp->ms[1] = 0.f;
    for_all(sp, [&](int i) {
      return gprhip_eval_pass1(sp->parts[i], h, want_grad, sp->n, problem_ar1(sp->parts[i]));
    });
    exchange(sp, 0, gprhip_ar1_len(sp->parts[0]), problem_ar1);
    for_all(sp, [&](int i) {
      return gprhip_eval_pass2(sp->parts[i], problem_ar1(sp->parts[i]), problem_ar2(sp->parts[i]));
    });
    // an evidence-only evaluation carries nothing in the second buffer (pass 2 clears its scalar tail locally)
    if (want_grad) exchange(sp, 1, gprhip_ar2_len(sp->parts[0]), problem_ar2);
    // first shard: the m x m work of the gradient and the result copies go onto its stream now; the other shards only
    // report their factorisation flags (the reduced buffers, hence the factors, are identical everywhere)
    int st = problem_finish_enqueue(sp->parts[0], problem_ar2(sp->parts[0]), 0);
    if (st != GPRHIP_OK) throw HipFail{st};
    for (int i = 1; i < nd; ++i) {
      st = problem_finish_enqueue(sp->parts[i], problem_ar2(sp->parts[i]), 1);
      if (st != GPRHIP_OK) throw HipFail{st};
    }
    int first = GPRHIP_OK;
    std::string msg;
    for (int i = nd - 1; i >= 0; --i) {
      st = problem_finish_collect(sp->parts[i], res, grad, coeffs, i > 0);
      if (st != GPRHIP_OK) {
        first = st;
        msg = last_error();
      }
    }
    if (sp->timing && sp->collectives > 0) {
      GPR_HIP(hipSetDevice(sp->ctx->devices[0]));
      for (int k = 0; k < sp->collectives; ++k) hipEventElapsedTime(&sp->ms[k], sp->t0[k], sp->t1[k]);
    }
    if (first != GPRHIP_OK) fail(first, msg);
  });
}

int gprhip_sharded_predict(gprhip_sharded* sp, const double* test_inputs, int64_t ld, int64_t nt, int predictive,
                           double* means, double* variances) {
  return guarded([&] {
    if (!sp || !test_inputs || nt < 1) fail(GPRHIP_EBADARG, "gprhip_sharded_predict: invalid arguments");
    const int nd = (int)sp->parts.size();
    if (nt < nd) {  // fewer points than devices: the first shard serves them
      const int st = gprhip_predict(sp->parts[0], test_inputs, ld, nt, predictive, means, variances);
      if (st != GPRHIP_OK) throw HipFail{st};
      return;
    }
    for_all(sp, [&](int i) {
      int64_t lo = 0, hi = 0;
      int st = gprhip_shard_rows(nt, nd, i, &lo, &hi);
      if (st != GPRHIP_OK) return st;
      return gprhip_predict(sp->parts[i], test_inputs + lo * ld, ld, hi - lo, predictive, means ? means + lo : nullptr,
                            variances ? variances + lo : nullptr);
    });
  });
}

int gprhip_sharded_train_stats(gprhip_sharded* sp, double* means, double* sums) {
  return guarded([&] {
    if (!sp || !sums) fail(GPRHIP_EBADARG, "gprhip_sharded_train_stats: invalid arguments");
    const int nd = (int)sp->parts.size();
    std::vector<double> part((size_t)nd * 4, 0.0);
    for_all(sp, [&](int i) {
      return gprhip_train_stats(sp->parts[i], means ? means + sp->lo[i] : nullptr, part.data() + (size_t)i * 4);
    });
    sums[0] = sums[1] = sums[2] = sums[3] = 0.0;
    for (int i = 0; i < nd; ++i) {  // fixed order: reproducible
      sums[0] += part[(size_t)i * 4 + 0];
      sums[1] += part[(size_t)i * 4 + 1];
      sums[2] = std::max(sums[2], part[(size_t)i * 4 + 2]);
      sums[3] += part[(size_t)i * 4 + 3];
    }
  });
}

int gprhip_sharded_comm_stats(const gprhip_sharded* sp, int* collectives, int64_t bytes[2], float ms[2]) {
  return guarded([&] {
    if (!sp) fail(GPRHIP_EBADARG, "gprhip_sharded_comm_stats: NULL argument");
    if (collectives) *collectives = sp->collectives;
    for (int k = 0; k < 2; ++k) {
      if (bytes) bytes[k] = sp->bytes[k];
      if (ms) ms[k] = sp->ms[k];
    }
  });
}

int gprhip_sharded_set_timing(gprhip_sharded* sp, int level) {
  return guarded([&] {
    if (!sp || level < 0 || level > 2) fail(GPRHIP_EBADARG, "gprhip_sharded_set_timing: invalid arguments");
    sp->timing = level;
  });
}

}  // extern "C"
