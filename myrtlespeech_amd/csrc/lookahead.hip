// DS2 lookahead convolution (model/lookahead.py:65-69): right-pad context-1 frames, then a
// depthwise conv over the future `ctx` frames:  y[n,f,t] = sum_k w[f,k] * x[n,f,t+k].
// HBM/L2-bound sliding dot product, no MFMA.  Element strides on x and y let the caller
// hand over the RNN output in its native [T,N,F] layout (deep_speech_2.py:119-121,161-164)
// and receive [N,T,F] for the fully-connected stack without materialising a permute.
//
// Two access patterns: time-contiguous (xs_t == 1: one workgroup stages a row segment of a
// single (n,f) in LDS) and feature-contiguous (xs_f == 1: lanes walk f, every tap is a
// coalesced row read that the L2 serves on re-use).
#include "common.h"

namespace {

constexpr int LA_TB = 256;  // output frames per workgroup (time-contiguous variant)

__global__ __launch_bounds__(256) void lookahead_tcontig_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ y, int F, int T, int To, int ctx,
                                                                long xs_n, long xs_f, long ys_n, long ys_f, long ys_t,
                                                                int act, float lo, float hi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [LA_TB + ctx - 1]
  float* ws = smem + LA_TB + ctx;   // [ctx]
  const int f = blockIdx.y, n = blockIdx.z, t0 = blockIdx.x * LA_TB, tid = threadIdx.x;
  const float* xr = x + n * xs_n + f * xs_f;
  for (int i = tid; i < LA_TB + ctx - 1; i += 256) xs[i] = (t0 + i < T) ? xr[t0 + i] : 0.f;
  for (int i = tid; i < ctx; i += 256) ws[i] = w[(size_t)f * ctx + i];
  __syncthreads();
  const int t = t0 + tid;
  if (t < To) {
    float acc = 0.f;
    for (int k = 0; k < ctx; ++k) acc += ws[k] * xs[tid + k];
    if (act == MS_ACT_CLAMP) acc = fminf(fmaxf(acc, lo), hi);
    y[n * ys_n + f * ys_f + t * ys_t] = acc;
  }
}

// Feature-contiguous layout (the RNN output [T, N, F]): lanes walk f, so every tap of every frame is a coalesced row
// read.  A thread keeps LA_TT consecutive output frames of its feature in registers and consumes the taps in chunks
// of LA_KC: per chunk the LA_KC weights sit in registers and each of the LA_TT + LA_KC - 1 input frames is loaded
// ONCE and fanned out to the outputs it contributes to (all indices static after unrolling) -- 7 loads per output
// instead of one weight + one input load per tap (160 at context 80).
// (round 4) The inputs of a chunk are requested TOGETHER, unconditionally (frames past the end: the last frame's address,
// the value replaced by zero afterwards) -- behind a per-load predicate the compiler waited for every load before issuing the
// next, one L2 round trip per input frame: 92 us for a streaming window of 16 output frames.  TT = 16 for such windows (half
// the loads and no outputs computed for nothing); the per-output order of the products is the same for every TT.
constexpr int LA_TT = 32, LA_KC = 16;

template <int TT>
__global__ __launch_bounds__(256) void lookahead_strided_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ y, int F, int T, int To, int ctx,
                                                                long xs_n, long xs_f, long xs_t, long ys_n, long ys_f,
                                                                long ys_t, int act, float lo, float hi) {
  const int f = blockIdx.x * 256 + threadIdx.x, t0 = blockIdx.y * TT, n = blockIdx.z;
  if (f >= F) return;
  const float* xr = x + n * xs_n + f * xs_f;
  const float* wr = w + (size_t)f * ctx;
  float acc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < ctx; k0 += LA_KC) {
    float wk[LA_KC], xv[TT + LA_KC - 1];
#pragma unroll
    for (int j = 0; j < TT + LA_KC - 1; ++j) xv[j] = xr[(long)min(t0 + k0 + j, T - 1) * xs_t];
#pragma unroll
    for (int k = 0; k < LA_KC; ++k) wk[k] = wr[min(k0 + k, ctx - 1)];
#pragma unroll
    for (int k = 0; k < LA_KC; ++k) wk[k] = (k0 + k < ctx) ? wk[k] : 0.f;
#pragma unroll
    for (int j = 0; j < TT + LA_KC - 1; ++j) xv[j] = (t0 + k0 + j < T) ? xv[j] : 0.f;
    // (round 6) output-major: two loops of constant trip count, every index static.  The input-major form (for j: for i:
    // k = j - i, guarded) left hipcc's unroller with a 47 x 32 body at TT = 32 that it did not flatten: xv[] was indexed at
    // run time and lived in scratch memory (192 bytes per lane).  Per output the products still arrive in tap order.
#pragma unroll
    for (int i = 0; i < TT; ++i) {
#pragma unroll
      for (int k = 0; k < LA_KC; ++k) acc[i] += wk[k] * xv[i + k];
    }
  }
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const int t = t0 + i;
    if (t < To) {
      float v = acc[i];
      if (act == MS_ACT_CLAMP) v = fminf(fmaxf(v, lo), hi);
      y[n * ys_n + f * ys_f + t * ys_t] = v;
    }
  }
}

}  // namespace

static int lookahead_launch(const float* x, const float* w, float* y, int N, int F, int T, int To, int ctx, long xs_n,
                            long xs_f, long xs_t, long ys_n, long ys_f, long ys_t, int act, float act_lo, float act_hi,
                            void* stream) {
  ms::ProfScope prof_span(MS_PROF_OTHER, (hipStream_t)stream);
  MS_REQUIRE(x && w && y, "null pointer");
  MS_REQUIRE(N > 0 && F > 0 && T > 0 && ctx > 0 && To > 0 && To <= T, "bad shape");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  MS_REQUIRE(N <= 65535 && F <= 65535 && T <= 65535, "dimension exceeds grid limits");
  if (xs_t == 1) {
    const size_t lds = (size_t)(LA_TB + 2 * ctx) * sizeof(float);
    MS_REQUIRE(lds <= 64 * 1024, "context too large");
    hipLaunchKernelGGL(lookahead_tcontig_kernel, dim3(ms::cdiv(To, LA_TB), F, N), dim3(256), lds, (hipStream_t)stream, x,
                       w, y, F, T, To, ctx, xs_n, xs_f, ys_n, ys_f, ys_t, act, act_lo, act_hi);
  } else {
    if (To <= 16)
      hipLaunchKernelGGL(lookahead_strided_kernel<16>, dim3(ms::cdiv(F, 256), 1, N), dim3(256), 0, (hipStream_t)stream, x, w, y, F, T, To,
                         ctx, xs_n, xs_f, xs_t, ys_n, ys_f, ys_t, act, act_lo, act_hi);
    else
      hipLaunchKernelGGL(lookahead_strided_kernel<LA_TT>, dim3(ms::cdiv(F, 256), ms::cdiv(To, LA_TT), N), dim3(256), 0,
                         (hipStream_t)stream, x, w, y, F, T, To, ctx, xs_n, xs_f, xs_t, ys_n, ys_f, ys_t, act, act_lo, act_hi);
  }
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_lookahead_forward(const float* x, const float* w, float* y, int N, int F, int T, int ctx, long xs_n,
                                    long xs_f, long xs_t, long ys_n, long ys_f, long ys_t, int act, float act_lo,
                                    float act_hi, void* stream) {
  return lookahead_launch(x, w, y, N, F, T, T, ctx, xs_n, xs_f, xs_t, ys_n, ys_f, ys_t, act, act_lo, act_hi, stream);
}

extern "C" int ms_lookahead_window_forward(const float* x, const float* w, float* y, int N, int F, int T_in, int T_out,
                                           int ctx, long xs_n, long xs_f, long xs_t, long ys_n, long ys_f, long ys_t,
                                           int act, float act_lo, float act_hi, void* stream) {
  return lookahead_launch(x, w, y, N, F, T_in, T_out, ctx, xs_n, xs_f, xs_t, ys_n, ys_f, ys_t, act, act_lo, act_hi, stream);
}
