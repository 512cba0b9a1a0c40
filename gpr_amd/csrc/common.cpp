#include "common.h"

namespace gprhip {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }
std::mutex& device_once_mutex() {
  static std::mutex m;
  return m;
}
}  // namespace gprhip
