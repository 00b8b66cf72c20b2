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

// Operand precision of the split kernels, from MS_PRECISION: 0 = "f32" (exact float32 MFMA), 1 = "bf16x3" (bf16 hi + lo
// planes, three bf16 MFMAs: ~2^-17 per product), 2 = "fp16" (ONE fp16 plane, one pass: ~2^-11 operands), 3 = "f16x3"
// (default since round 6: fp16 hi + lo planes, three fp16 MFMAs).  f16x3 keeps 22 mantissa bits of every operand where
// bf16x3 keeps 16 -- the MFMA honours fp16 subnormal inputs (tools/micro/mfma_f16_denorm.hip), so a lo plane below 2^-14
// degrades gradually (absolute step 2^-24) instead of vanishing -- at the same MFMA rate; what it gives up is range: operands
// are clamped to +-65504 before the split (values up to 2 x 65504 still come out exact to ~2^-11 through the lo plane).
// bf16x3's 2^-17 was enough for a default-init network (logits of 0.02) and is NOT for a trained-scale one: 7.7e-3 on
// logits of 2.8 +- 17 with 4 of 32 greedy transcripts changed (tests/golden/ds2_cfg2_trained_summary.npz).
enum { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_F16 = 2, PREC_F16X3 = 3 };
int precision_mode();
int split_mode();   // the two-plane (or one-plane fp16) format of this process: precision_mode(), with f32 mapped to the default split
// ---- per-tensor power-of-two scale of a WEIGHT tensor's fp16 planes.  An fp16 lo plane is exact to 2^-24 absolute, which
// for weights of ~0.01 .. 0.03 (a default-init or a trained layer alike) is 2^-18 .. 2^-20 of the weight: the lo values sit
// in the subnormal range.  Scaling the tensor by 2^s with max |w| 2^s in [2^12, 2^13) before the split puts hi and lo of
// every weight within 2^-16 of the largest one in the normal range (22 mantissa bits), the products stay far inside float32
// (|x| <= 65504, K <= 2^20), and the kernel multiplies its accumulator by 2^-s -- exact -- in the epilogue.  The word
// {2^s, 2^-s} lives in device memory beside the packed planes (no host round trip at pack time); bf16 planes get {1, 1}.
// a contraction run whole, or as the first / second of two launches cut along K (gemm_split.hip: gemm4_body)
enum { GEMM_K_WHOLE = 0, GEMM_K_FIRST = 1, GEMM_K_SECOND = 2 };
int weight_scale_launch(const float* w, size_t n, float* scale_word, int prec, hipStream_t stream);

// the two-plane split modes' plane format: the kernels take it as a template parameter `P` (one of the PREC_* above)
constexpr bool prec_one_plane(int P) { return P == PREC_F16; }
constexpr bool prec_half_planes(int P) { return P == PREC_F16X3 || P == PREC_F16; }

// ---- plane element conversions and the matching MFMA opcodes (HM: the planes hold fp16 values, else bf16)
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
template <bool HM>
__device__ __forceinline__ unsigned plane_bits(float x) {
  if constexpr (HM) return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f));
  else return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
}
// (for values known to lie in the fp16 range -- a cell's h in [-1, 1]: no clamp on the recurrence's critical path)
template <bool HM>
__device__ __forceinline__ unsigned plane_bits_bounded(float x) {
  if constexpr (HM) return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x);
  else return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
}
template <bool HM>
__device__ __forceinline__ float plane_val(unsigned bits) {
  if constexpr (HM) return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
  else return __uint_as_float(bits << 16);
}
// hi / lo bits of one float32: hi = rn(x), lo = rn(x - hi)
template <bool HM>
__device__ __forceinline__ void plane_split(float x, unsigned& hi, unsigned& lo) {
  hi = plane_bits<HM>(x);
  lo = plane_bits<HM>(x - plane_val<HM>(hi));
}
template <bool HM>
__device__ __forceinline__ void plane_split_bounded(float x, unsigned& hi, unsigned& lo) {
  hi = plane_bits_bounded<HM>(x);
  lo = plane_bits_bounded<HM>(x - plane_val<HM>(hi));
}
template <bool HM>
__device__ __forceinline__ f32x16 mfma_32x32x16(u32x4_ a, u32x4_ b, f32x16 c) {
  if constexpr (HM) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_, a), __builtin_bit_cast(f16x8_, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
template <bool HM>
__device__ __forceinline__ f32x4 mfma_16x16x32(u32x4_ a, u32x4_ b, f32x4 c) {
  if constexpr (HM) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_, a), __builtin_bit_cast(f16x8_, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}

}  // namespace ms
