// MaskConv1d with many input channels as im2col + split-bf16 GEMM (model/cnn.py:295-333 for the conv1d flavour of the DS2
// builder, builders/speech_to_text.py:69-83).  The channels-last implicit-GEMM kernel (conv_cl.hip) keeps a whole
// input patch of every channel in LDS, which does not fit once Cin * patch width grows (Cin = 512, k = 11: 565 KB), and
// the exact-f32 tap kernel then runs at f32-MFMA speed (80 TF).  Here the patches are written once as bf16 hi / lo
// planes [N * Tout, Cin * KT padded to 32] -- the time mask and the SAME padding are load predicates of that pass -- and the
// contraction is the 256 x 256 split GEMM with the bias + clamp epilogue; a tiled transpose returns the reference layout.
#include "common.h"

namespace ms {
int gemm_bf16x3_launch(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                       const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                       float hi, int prec, hipStream_t stream);
}

namespace {

// (HM: fp16 hi / lo planes -- the default f16x3 mode -- else bf16 pairs)
template <bool HM>
__device__ __forceinline__ void split_store(float x, unsigned short* hi, unsigned short* lo, size_t at) {
  unsigned h, l;
  ms::plane_split<HM>(x, h, l);
  hi[at] = (unsigned short)h;
  lo[at] = (unsigned short)l;
}

// weight [Cout, Cin*KT] f32 -> planes [Cout, Kp] (zero padded columns)
template <bool HM>
__global__ void conv1d_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi,
                                   unsigned short* __restrict__ lo, int K, int Kp) {
  const int row = blockIdx.x;
  for (int k = threadIdx.x; k < Kp; k += blockDim.x)
    split_store<HM>(k < K ? w[(size_t)row * K + k] : 0.f, hi, lo, (size_t)row * Kp + k);
}

// patches: row (n, t_out), column ci*KT + k  <-  x[n, ci, t_out*ST + k*DT - pad_l] if that frame exists and is < lens[n].
// A workgroup builds a 32-row x 64-column tile through LDS: the gather runs with lanes along t_out (consecutive lanes read
// consecutive frames of one channel: coalesced), the scatter with 8 lanes per row (one 16-byte store per plane each).
template <bool HM>
__global__ __launch_bounds__(256) void conv1d_im2col_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                            unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                            int Cin, int Tin, int Tout, int KT, int ST, int DT, int pad_l,
                                                            int K, int Kp) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) unsigned short th[32][72], tl[32][72];  // +8 pad: row stride 144 B
  const int n = blockIdx.z, t0 = blockIdx.y * 32, c_base = blockIdx.x * 64, tid = threadIdx.x;
  const int len = min(lens[n], Tin);
  const float* xn = x + (size_t)n * Cin * Tin;
  {
    const int tr = tid & 31, cg = tid >> 5;
    const int t_out = t0 + tr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cl = cg * 8 + e, c = c_base + cl;
      float v = 0.f;
      if (c < K && t_out < Tout) {
        const int ci = c / KT, k = c - ci * KT;
        const int t = t_out * ST + k * DT - pad_l;
        if (t >= 0 && t < len) v = xn[(size_t)ci * Tin + t];
      }
      unsigned hb, lb;
      ms::plane_split<HM>(v, hb, lb);
      th[tr][cl] = (unsigned short)hb;
      tl[tr][cl] = (unsigned short)lb;
    }
  }
  __syncthreads();
  {
    const int tr = tid >> 3, ch = tid & 7;
    const int t_out = t0 + tr;
    if (t_out < Tout && c_base + ch * 8 < Kp) {
      const size_t at = ((size_t)n * Tout + t_out) * Kp + c_base + ch * 8;
      *reinterpret_cast<u32x4*>(hi + at) = *reinterpret_cast<const u32x4*>(&th[tr][ch * 8]);
      *reinterpret_cast<u32x4*>(lo + at) = *reinterpret_cast<const u32x4*>(&tl[tr][ch * 8]);
    }
  }
}

// y [N, T, C] -> out [N, C, T]
__global__ void ntc_to_nct_kernel(const float* __restrict__ y, float* __restrict__ out, int T, int C) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (t < T && c < C) ? y[((size_t)n * T + t) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (c < C && t < T) out[((size_t)n * C + c) * T + t] = tile[tx][i];
  }
}

inline int padded_k(int Cin, int KT) { return ms::cdiv(Cin * KT, 32) * 32; }
// two planes always (MS_PRECISION=fp16 has no one-plane form of this path): fp16 pairs in f16x3 mode, bf16 pairs otherwise
inline bool half_planes() { return ms::precision_mode() == ms::PREC_F16X3; }

}  // namespace

extern "C" size_t ms_maskconv1d_gemm_packed_bytes(int Cout, int Cin, int KT) {
  if (Cout <= 0 || Cin <= 0 || KT <= 0) return 0;
  return (size_t)2 * Cout * padded_k(Cin, KT) * sizeof(unsigned short);
}

extern "C" int ms_maskconv1d_gemm_pack(const float* w, void* packed, int Cout, int Cin, int KT, void* stream) {
  MS_REQUIRE(w && packed, "null pointer");
  MS_REQUIRE(Cout > 0 && Cin > 0 && KT > 0, "bad shape");
  const int Kp = padded_k(Cin, KT);
  unsigned short* hi = (unsigned short*)packed;
  hipLaunchKernelGGL(half_planes() ? conv1d_pack_kernel<true> : conv1d_pack_kernel<false>, dim3(Cout), dim3(256), 0,
                     (hipStream_t)stream, w, hi, hi + (size_t)Cout * Kp, Cin * KT, Kp);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" size_t ms_maskconv1d_gemm_workspace_bytes(int N, int Cin, int Tout, int Cout, int KT) {
  if (N <= 0 || Cin <= 0 || Tout <= 0 || Cout <= 0 || KT <= 0) return 0;
  const size_t rows = (size_t)N * Tout;
  return ms::align_up(rows * padded_k(Cin, KT) * 2 * sizeof(unsigned short), 256) +
         ms::align_up(rows * Cout * sizeof(float), 256);
}

extern "C" int ms_maskconv1d_gemm_forward(const float* x, const int32_t* lens, const void* packed, const float* bias,
                                          float* y, int N, int Cin, int Tin, int Cout, int Tout, int KT, int ST, int DT,
                                          int pad_l, int act, float act_lo, float act_hi, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
  ms::ProfScope prof_span(MS_PROF_CONV, (hipStream_t)stream_);
  MS_REQUIRE(x && lens && packed && y && workspace, "null pointer");
  MS_REQUIRE(N > 0 && Cin > 0 && Tin > 0 && Cout > 0 && Tout > 0 && KT > 0 && ST > 0 && DT > 0 && pad_l >= 0, "bad shape");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  MS_REQUIRE(N <= 65535 && Tout <= 65535 * 32 && (long long)N * Tout <= 2147483647LL, "N / Tout exceed grid limits");
  MS_REQUIRE(workspace_bytes >= ms_maskconv1d_gemm_workspace_bytes(N, Cin, Tout, Cout, KT), "workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int K = Cin * KT, Kp = padded_k(Cin, KT);
  const size_t rows = (size_t)N * Tout;
  unsigned short* ph = (unsigned short*)workspace;
  unsigned short* pl = ph + rows * Kp;
  float* yt = (float*)((char*)workspace + ms::align_up(rows * Kp * 2 * sizeof(unsigned short), 256));
  const unsigned short* wh = (const unsigned short*)packed;
  const unsigned short* wl = wh + (size_t)Cout * Kp;
  hipLaunchKernelGGL(half_planes() ? conv1d_im2col_kernel<true> : conv1d_im2col_kernel<false>, dim3(ms::cdiv(Kp, 64)
                     , ms::cdiv(Tout, 32), N), dim3(256), 0, stream, x, lens, ph, pl, Cin, Tin, Tout, KT, ST, DT, pad_l, K, Kp);
  MS_LAUNCH_CHECK();
  int rc = ms::gemm_bf16x3_launch(ph, pl, wh, wl, bias, yt, (int)rows, Kp, Cout, act, act_lo, act_hi,
                                  half_planes() ? ms::PREC_F16X3 : ms::PREC_BF16X3, stream);
  if (rc != MS_OK) return rc;
  hipLaunchKernelGGL(ntc_to_nct_kernel, dim3(ms::cdiv(Tout, 32), ms::cdiv(Cout, 32), N), dim3(32, 8), 0, stream, yt, y, Tout,
                     Cout);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
