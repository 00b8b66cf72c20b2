// Shared host/device helpers for libms_hotpath.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/ms_hotpath.h"

namespace ms {

void set_error(const std::string& s);

#define MS_REQUIRE(cond, msg)                                              \
  do {                                                                     \
    if (!(cond)) {                                                         \
      ms::set_error(std::string(__func__) + ": " + (msg));                 \
      return MS_ERR_INVALID;                                               \
    }                                                                      \
  } while (0)

#define MS_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      ms::set_error(std::string(__func__) + ": " #call " -> " + hipGetErrorString(e_));     \
      return MS_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define MS_LAUNCH_CHECK() MS_HIP(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Row of accumulator register `reg` of a 32x32 MFMA tile held by `lane`
// (column is lane & 31): cdna_hip_programming.md §3.
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

int num_cus();

// "Once per device" guard for hipFuncSetAttribute (function attributes belong to the device the module is loaded on;
// a process that drives more than one device must set them on each).  A race between two host threads at worst
// repeats an idempotent call.
struct DeviceOnce {
  std::atomic<unsigned long long> done_mask{0};
  static unsigned long long bit() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return 1ull << (d & 63);
  }
  bool need() const { return !(done_mask.load(std::memory_order_acquire) & bit()); }
  void done() { done_mask.fetch_or(bit(), std::memory_order_release); }
};

// Optional per-family launch timing (ms_prof_enable / ms_prof_read; api_common.cpp): while enabled, an entry point
// brackets its launches with HIP events on the caller's stream.  Kinds: MS_PROF_* in include/ms_hotpath.h.
struct ProfScope {
  hipEvent_t a = nullptr, b = nullptr;
  int kind;
  hipStream_t stream;
  bool on;
  ProfScope(int kind, hipStream_t st);
  ~ProfScope();
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
};

// Operand precision of the split kernels, from MS_PRECISION: 0 = "f32" (exact float32 MFMA),
// 1 = "bf16x3" (default: bf16 hi+lo, three MFMAs), 2 = "fp16" (single fp16 pass; ~2^-11 operands).
enum { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_F16 = 2 };
int precision_mode();

}  // namespace ms
