// Error string + device query shared by every entry point.
#include "common.h"

namespace ms {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int num_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    cus = p.multiProcessorCount;
  }
  return cus;
}
}  // namespace ms

extern "C" int ms_abi_version(void) { return MS_ABI_VERSION; }
extern "C" const char* ms_last_error(void) {
  static thread_local std::string copy;
  copy = ms::g_err;
  return copy.c_str();
}
