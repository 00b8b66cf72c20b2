// Masked convolution over (feature, time) as an implicit GEMM on the f32 matrix cores.
//
// Replaces MaskConv2d.forward / MaskConv1d.forward (model/cnn.py:445-483, 295-333):
//   _mask_  (cnn.py:425-443)  -> predicate t_in < lens[n] while staging the input patch
//   _pad    (cnn.py:391-423)  -> index arithmetic with the LEFT pads of pad_same (cnn.py:148-163)
//   Conv2d  (cnn.py:481)      -> v_mfma_f32_32x32x2_f32 (exact f32), bias + clamp epilogue
//
// Decomposition: one workgroup (4 waves) owns 32 output channels x 128 output frames of
// one (n, f_out) row; wave w owns frames [32w, 32w+32).  The reduction axis is walked as
// rows r = (cin, kf) x kernel-time taps kt (padded to an even count KT2 with zero taps).
// Per chunk of RC rows the workgroup stages
//   Wl[r][kt][32 cout]      (contiguous copy out of the pre-packed filter bank)
//   Pl[r][p]                (the input row f_in = f_out*SF - pad_f + kf*DF, frames
//                            t0*ST - pad_t + p, already masked / zero padded)
// and each MFMA takes A = Wl[r][2m+h][cout = lane&31], B = Pl[r][(32w + lane&31)*ST + (2m+h)*DT]
// (h = lane>>5 picks the tap of the k-pair).  Output columns are time, so the NCHW store
// is 128 contiguous bytes per accumulator register.
#include <algorithm>

#include "common.h"

namespace {

constexpr int CO_T = 32;     // output channels per workgroup
constexpr int T_WG = 128;    // output frames per workgroup
constexpr int RC_MAX = 16;   // reduction rows per LDS chunk

struct ConvP {
  int N, Cin_g, Fin, Tin, Cout_g, Fout, Tout, KF, KT, KT2, SF, ST, DF, DT, pad_f, pad_t;
  int R;         // Cin_g * KF
  int PW;        // staged patch width = (T_WG-1)*ST + (KT2-1)*DT + 1
  int PWS;       // padded row stride of Pl
  int co_tiles;  // ceil(Cout_g / 32)
  long x_nstride, y_nstride;
  int act;
  float lo, hi;
};

__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int R, int KT,
                                 int KT2, int co_tiles) {
  // packed[tile][r][kt2][32]
  const size_t total = (size_t)co_tiles * R * KT2 * CO_T;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % CO_T;
    const int kt = (i / CO_T) % KT2;
    const int r = (i / ((size_t)CO_T * KT2)) % R;
    const int tile = i / ((size_t)CO_T * KT2 * R);
    const int co = tile * CO_T + c;
    packed[i] = (co < Cout && kt < KT) ? w[((size_t)co * R + r) * KT + kt] : 0.f;
  }
}

__global__ __launch_bounds__(256) void maskconv_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                       const float* __restrict__ wp, const float* __restrict__ bias,
                                                       float* __restrict__ y, ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                                    // [RC][KT2][32]
  float* Pl = smem + RC_MAX * p.KT2 * CO_T;            // [RC][PWS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int t0 = blockIdx.x * T_WG;
  const int fo = blockIdx.y;
  const int n = blockIdx.z / p.co_tiles;
  const int tile = blockIdx.z % p.co_tiles;
  const int len = lens ? min(lens[n], p.Tin) : p.Tin;
  const float* xn = x + (size_t)n * p.x_nstride;
  const float* wt = wp + (size_t)tile * p.R * p.KT2 * CO_T;

  ms::f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int tin0 = t0 * p.ST - p.pad_t;
  const int fin0 = fo * p.SF - p.pad_f;
  const int bcol = (wave * 32 + l31) * p.ST + half * p.DT;

  for (int r0 = 0; r0 < p.R; r0 += RC_MAX) {
    const int rc = min(RC_MAX, p.R - r0);
    // stage the filter chunk: contiguous rc*KT2*32 floats
    {
      const int nf = rc * p.KT2 * CO_T;  // multiple of 64 (KT2 even)
      const ms::f32x4* src = reinterpret_cast<const ms::f32x4*>(wt + (size_t)r0 * p.KT2 * CO_T);
      ms::f32x4* dst = reinterpret_cast<ms::f32x4*>(Wl);
      for (int i = tid; i < nf / 4; i += 256) dst[i] = src[i];
    }
    // stage the input patch rows
    for (int rr = wave; rr < rc; rr += 4) {
      const int r = r0 + rr;
      const int cin = r / p.KF, kf = r - cin * p.KF;
      const int fin = fin0 + kf * p.DF;
      const bool frow = (fin >= 0 && fin < p.Fin);
      const float* xr = xn + ((size_t)cin * p.Fin + (frow ? fin : 0)) * p.Tin;
      for (int q = lane; q < p.PW; q += 64) {
        const int tin = tin0 + q;
        float v = 0.f;
        if (frow && tin >= 0 && tin < len) v = xr[tin];
        Pl[rr * p.PWS + q] = v;
      }
    }
    __syncthreads();
    for (int rr = 0; rr < rc; ++rr) {
      const float* wrow = Wl + rr * p.KT2 * CO_T + half * CO_T + l31;
      const float* prow = Pl + rr * p.PWS + bcol;
      for (int m = 0; m < p.KT2 / 2; ++m) {
        const float a = wrow[2 * m * CO_T];
        const float b = prow[2 * m * p.DT];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int t = t0 + wave * 32 + l31;
  if (t < p.Tout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = tile * CO_T + ms::mfma32_row(r, lane);
      if (co < p.Cout_g) {
        float v = acc[r] + (bias ? bias[co] : 0.f);
        if (p.act == MS_ACT_CLAMP) v = fminf(fmaxf(v, p.lo), p.hi);
        y[(size_t)n * p.y_nstride + ((size_t)co * p.Fout + fo) * p.Tout + t] = v;
      }
    }
  }
}

}  // namespace

extern "C" size_t ms_maskconv_packed_bytes(int Cout, int Cin_g, int KF, int KT, int groups) {
  if (Cout <= 0 || Cin_g <= 0 || KF <= 0 || KT <= 0 || groups <= 0 || Cout % groups) return 0;
  const int KT2 = (KT + 1) & ~1;
  return (size_t)groups * ms::cdiv(Cout / groups, CO_T) * Cin_g * KF * KT2 * CO_T * sizeof(float);
}

extern "C" int ms_maskconv_pack(const float* w, void* packed, int Cout, int Cin_g, int KF, int KT, int groups,
                                void* stream) {
  MS_REQUIRE(w && packed, "null pointer");
  MS_REQUIRE(Cout > 0 && Cin_g > 0 && KF > 0 && KT > 0 && groups > 0 && Cout % groups == 0, "bad shape");
  const int KT2 = (KT + 1) & ~1;
  const int R = Cin_g * KF;
  const int Cout_g = Cout / groups;
  const int tiles = ms::cdiv(Cout_g, CO_T);
  const size_t total = (size_t)tiles * R * KT2 * CO_T;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  for (int g = 0; g < groups; ++g) {
    hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       w + (size_t)g * Cout_g * R * KT, (float*)packed + g * total, Cout_g, R, KT, KT2, tiles);
    MS_LAUNCH_CHECK();
  }
  return MS_OK;
}

extern "C" int ms_maskconv_forward(const float* x, const int32_t* lens, const void* packed_w, const float* bias,
                                   float* y, int N, int Cin, int Fin, int Tin, int Cout, int Fout, int Tout, int KF,
                                   int KT, int SF, int ST, int DF, int DT, int pad_f_l, int pad_t_l, int groups, int act,
                                   float act_lo, float act_hi, void* stream) {
  ms::ProfScope prof_span(MS_PROF_CONV, (hipStream_t)stream);
  MS_REQUIRE(x && packed_w && y, "null pointer");
  MS_REQUIRE(N > 0 && Cin > 0 && Fin > 0 && Tin > 0 && Cout > 0 && Fout > 0 && Tout > 0, "bad shape");
  MS_REQUIRE(KF > 0 && KT > 0 && SF > 0 && ST > 0 && DF > 0 && DT > 0, "bad kernel/stride/dilation");
  MS_REQUIRE(pad_f_l >= 0 && pad_t_l >= 0, "negative padding");
  MS_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "groups must divide channels");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  ConvP p;
  p.N = N; p.Cin_g = Cin / groups; p.Fin = Fin; p.Tin = Tin; p.Cout_g = Cout / groups; p.Fout = Fout; p.Tout = Tout;
  p.KF = KF; p.KT = KT; p.KT2 = (KT + 1) & ~1; p.SF = SF; p.ST = ST; p.DF = DF; p.DT = DT;
  p.pad_f = pad_f_l; p.pad_t = pad_t_l;
  p.R = p.Cin_g * KF;
  p.PW = (T_WG - 1) * ST + (p.KT2 - 1) * DT + 1;
  p.PWS = p.PW + 1;
  p.co_tiles = ms::cdiv(p.Cout_g, CO_T);
  p.x_nstride = (long)Cin * Fin * Tin;
  p.y_nstride = (long)Cout * Fout * Tout;
  p.act = act; p.lo = act_lo; p.hi = act_hi;
  const size_t lds = ((size_t)RC_MAX * p.KT2 * CO_T + (size_t)RC_MAX * p.PWS) * sizeof(float);
  MS_REQUIRE(lds <= 160 * 1024, "kernel_time/stride/dilation too large for the LDS patch");
  MS_REQUIRE(Fout <= 65535 && (long)N * p.co_tiles <= 65535, "Fout or N*cout_tiles exceed grid limits");
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)maskconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  const size_t per_group_packed = (size_t)p.co_tiles * p.R * p.KT2 * CO_T;
  for (int g = 0; g < groups; ++g) {
    const float* xg = x + (size_t)g * p.Cin_g * Fin * Tin;
    float* yg = y + (size_t)g * p.Cout_g * Fout * Tout;
    const float* wg = (const float*)packed_w + g * per_group_packed;
    const float* bg = bias ? bias + g * p.Cout_g : nullptr;
    dim3 grid(ms::cdiv(Tout, T_WG), Fout, N * p.co_tiles);
    hipLaunchKernelGGL(maskconv_kernel, grid, dim3(256), lds, (hipStream_t)stream, xg, lens, wg, bg, yg, p);
    MS_LAUNCH_CHECK();
  }
  return MS_OK;
}
