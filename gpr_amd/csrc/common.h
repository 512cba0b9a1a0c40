// Shared declarations for the gprhip library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

namespace gprhip {

// status codes returned across the C ABI (include/gprhip.h)
enum : int {
  ST_OK = 0,
  ST_BAD_ARG = 1,
  ST_NOT_POSDEF = 2,
  ST_HIP_ERROR = 3,
  ST_OOM = 4,
  ST_STATE = 5,
};

void set_error(const std::string& msg);
// Distinct libamdhip64.so files mapped into the process (common.cpp); the creation entry points refuse to run with two
std::vector<std::string> mapped_hip_runtimes();
void check_single_hip_runtime(const char* who);

struct HipFail {
  int status;
};

#define GPR_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) {                                                            \
      ::gprhip::set_error(std::string("gprhip: HIP error: ") + hipGetErrorString(e__) + \
                          " at " __FILE__ ":" + std::to_string(__LINE__));              \
      throw ::gprhip::HipFail{e__ == hipErrorOutOfMemory ? ::gprhip::ST_OOM             \
                                                         : ::gprhip::ST_HIP_ERROR};     \
    }                                                                                   \
  } while (0)

// One-time setup per HIP device (function attributes are per device; a context drives several devices, possibly from
// one host thread per device): runs f() the first time the calling thread's current device is seen for `mask`.
std::mutex& device_once_mutex();
template <typename F>
static inline void once_per_device(uint64_t& mask, F&& f) {
  std::lock_guard<std::mutex> g(device_once_mutex());
  int dev = 0;
  GPR_HIP(hipGetDevice(&dev));
  if ((mask >> (dev & 63)) & 1) return;
  f();
  mask |= uint64_t(1) << (dev & 63);
}

constexpr int TILE = 128;  // MFMA engine block tile (rows and columns)
constexpr int BK = 16;     // fp64 MFMA engine k-depth per LDS stage (fp32: 32)

static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Exchange buffers carry a symmetric mp x mp accumulation as its upper 128-tiles only, tile after tile (tile (bm, bn),
// bm <= bn, at index bn(bn+1)/2 + bm, 128 x 128 row-major): half the all-reduce payload of the full square.
static inline int64_t packed_upper_len(int mp) { return (int64_t)(mp / TILE) * (mp / TILE + 1) / 2 * TILE * TILE; }
__host__ __device__ static inline int64_t packed_upper_off(int r, int c) {
  const int bm = r / TILE, bn = c / TILE;
  return ((int64_t)bn * (bn + 1) / 2 + bm) * (TILE * TILE) + (int64_t)(r % TILE) * TILE + (c % TILE);
}

}  // namespace gprhip
