// Error string + device query shared by every entry point.
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace ms {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int num_cus() {
  static std::atomic<int> cus[64];  // per device ordinal; 0 = not queried yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int v = cus[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    v = p.multiProcessorCount;
    cus[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}
int precision_mode() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_PRECISION");
    v = PREC_BF16X3;
    if (e && strcmp(e, "f32") == 0) v = PREC_F32;
    if (e && strcmp(e, "fp16") == 0) v = PREC_F16;
  }
  return v;
}
}  // namespace ms

extern "C" int ms_abi_version(void) { return MS_ABI_VERSION; }
extern "C" const char* ms_last_error(void) {
  static thread_local std::string copy;
  copy = ms::g_err;
  return copy.c_str();
}
