#include "common.h"

#include <limits.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <vector>

namespace gprhip {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }
std::mutex& device_once_mutex() {
  static std::mutex m;
  return m;
}

// ---- one HIP runtime per process.
// libgprhip.so is linked against the system runtime (libamdhip64.so.N of /opt/rocm/lib); a host that ALSO loads a package
// bundling its own copy (PyTorch-ROCm ships torch/lib/libamdhip64.so) ends up with two runtimes mapped if this library was
// loaded first: the dynamic linker resolves the second request by path, not by soname.  Each copy then initialises the
// driver connection by itself, and the one that comes second finds no devices ("No HIP GPUs are available" from torch, or
// hipErrorNoDevice here) -- far from the cause.  Loaded in the other order (the bundling package first) this library's
// soname request is served by the copy already mapped and one runtime serves both.  The creation entry points therefore
// look at the process's link map and refuse to go on with two distinct runtime files mapped, naming both.
// GPRHIP_ALLOW_TWO_HIP_RUNTIMES=1 turns the refusal into silence (for hosts that know the second copy stays unused).
namespace {
int collect_hip_runtimes(struct dl_phdr_info* info, size_t, void* data) {
  auto* out = static_cast<std::vector<std::string>*>(data);
  const char* name = info->dlpi_name;
  if (!name || !*name) return 0;
  const char* base = std::strrchr(name, '/');
  base = base ? base + 1 : name;
  if (std::strncmp(base, "libamdhip64.so", 14) != 0) return 0;
  char real[PATH_MAX];
  std::string path = realpath(name, real) ? real : name;
  for (const auto& p : *out)
    if (p == path) return 0;
  out->push_back(path);
  return 0;
}
}  // namespace

std::vector<std::string> mapped_hip_runtimes() {
  std::vector<std::string> v;
  dl_iterate_phdr(collect_hip_runtimes, &v);
  return v;
}

void check_single_hip_runtime(const char* who) {
  const std::vector<std::string> v = mapped_hip_runtimes();
  if (v.size() < 2) return;
  if (const char* e = getenv("GPRHIP_ALLOW_TWO_HIP_RUNTIMES"))
    if (atoi(e) != 0) return;
  std::string msg = std::string(who) + ": two HIP runtimes are mapped into this process: ";
  for (size_t i = 0; i < v.size(); ++i) msg += (i ? " and " : "") + v[i];
  msg += " -- only one can own the devices.  Load the package that bundles its own runtime (e.g. `import torch`) BEFORE "
         "libgprhip.so, so that one copy serves both (INTEGRATION.md, \"hosts that also load torch\"); "
         "GPRHIP_ALLOW_TWO_HIP_RUNTIMES=1 overrides";
  set_error(msg);
  throw HipFail{ST_HIP_ERROR};
}
}  // namespace gprhip
