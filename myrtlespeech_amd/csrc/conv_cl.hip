// Masked 2-D convolution for multi-channel inputs as a split-bf16 implicit GEMM over CHANNELS.
//
// The second DS2 convolution (32 -> 32 channels, 21 x 11 taps: 152 GFLOP of the encoder's 170)
// spends 1.9 ms in the exact-f32 kernel of conv.hip.  Here the reduction runs over input channels
// (k-groups of 8 channels = one MFMA operand granule), so it needs no tap padding and every
// LDS read is an aligned 16-byte granule:
//
//   input   : channels-last bf16 planes  X_hi, X_lo [N][Fin][Tin][Cin]   (made from the f32 NCHW
//             input by nchw_to_cl_split_kernel: one tiled transpose + split, ~40 us)
//   filters : packed planes [hi|lo][KF][KT][Cin/8][Cout 32][8]            (ms_maskconv_cl_pack)
//   per workgroup (4 waves): 32 output channels x 128 output frames x 2 output feature rows;
//   wave w owns frames [32w, 32w+32).  For each kernel-feature row kf the workgroup stages the
//   two input rows it touches as [Cin/8][frame][8] granules and that row's KT filter taps; per tap
//   and 16-channel k-step: A = filter granule (rows = cout), B = input granule (cols = frames),
//   3 MFMAs (hi*hi + lo*hi + hi*lo) per output row.
//   _mask_ / _pad of cnn.py:391-443 are load predicates; bias + clamp in the epilogue; the
//   output is f32 NCHW (time contiguous) like conv.hip.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int CL_F = 2;    // output feature rows per wave (accumulator tiles)

struct ClP {
  int N, Cin, Fin, Tin, Cout, Fout, Tout, KF, KT, SF, ST, DF, DT, pad_f, pad_t;
  int KG;        // Cin / 8
  int PW;        // staged frames per input row = (frames per workgroup - 1)*ST + (KT-1)*DT + 1
  int co_tiles;
  int act;
  float lo, hi;
  // feature-window instantiation (single-channel input, ms_maskconv_fwin_*): the "channels" of output feature row fo are
  // the KF input feature rows of its window, read from planes [N][Tin][FP] at element offset fo * SF (KF = 1 in this struct)
  int FP;
};

__device__ __forceinline__ unsigned bf16b(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned f16b(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }
// hi / lo bits of x in plane format `prec` (ms::PREC_BF16X3 | PREC_F16 | PREC_F16X3; PREC_F16: lo = 0): layout kernels only
__device__ __forceinline__ void split_by_prec(float x, int prec, unsigned& hi, unsigned& lo) {
  if (prec == ms::PREC_F16) { hi = f16b(x); lo = 0u; }
  else if (prec == ms::PREC_F16X3) ms::plane_split<true>(x, hi, lo);
  else ms::plane_split<false>(x, hi, lo);
}

// packed[plane][kf][kt][kg][cout_pad][8] <- w[cout][cin][kf][kt]
__global__ void conv_cl_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int Cout, int Cin,
                                    int KF, int KT, int cout_pad, int prec, const float* __restrict__ scale_word) {
  const int KG = Cin / 8;
  const float wscale = scale_word[0];
  const size_t plane = (size_t)KF * KT * KG * cout_pad * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 7;
    const int co = (i >> 3) % cout_pad;
    const int kg = (i / ((size_t)8 * cout_pad)) % KG;
    const int kt = (i / ((size_t)8 * cout_pad * KG)) % KT;
    const int kf = i / ((size_t)8 * cout_pad * KG * KT);
    float x = 0.f;
    if (co < Cout) x = w[(((size_t)co * Cin + kg * 8 + e) * KF + kf) * KT + kt] * wscale;
    unsigned h, l;
    split_by_prec(x, prec, h, l);
    packed[i] = (unsigned short)h;
    packed[plane + i] = (unsigned short)l;
  }
}

// WIN: feature-window mode (ms_maskconv_fwin_*), a separate instantiation so that the multi-channel kernel's code is
// untouched by it.
// WF: waves along the feature axis.  WF = 1: the four waves own four consecutive 32-frame blocks (128 frames x CL_F rows
// per workgroup).  WF = 4: they own the SAME 32 frames of four consecutive row groups (32 frames x 4 CL_F rows) -- for short
// inputs (a 320 ms streaming chunk is 16 output frames) a 128-frame tile would be seven-eighths padding.
template <int P, bool WIN = false, int WF = 1>
__global__ __launch_bounds__(256, 2) void maskconv_cl_kernel(const unsigned short* __restrict__ xh,
                                                             const unsigned short* __restrict__ xl,
                                                             const int32_t* __restrict__ lens,
                                                             const unsigned short* __restrict__ wp,
                                                             const float* __restrict__ bias, float* __restrict__ y, ClP p,
                                                             const float* __restrict__ scale_word) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  constexpr int WT = 4 / WF;        // waves along time
  constexpr int TT = 32 * WT;       // output frames per workgroup
  constexpr int RF = CL_F * WF;     // output feature rows per workgroup
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // [plane 2][row RF][kg][PW][16 B]  then filters [plane 2][KT][kg][32][16 B]
  const int row_bytes = p.KG * p.PW * 16;
  char* Ph = lds;
  char* Pl = lds + RF * row_bytes;
  char* Wh = lds + 2 * RF * row_bytes;
  const int wtap = p.KG * 32 * 16;            // bytes of one tap's filters (one plane)
  char* Wl = Wh + ((p.KT + 1) / 2) * wtap;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wt = wave % WT, wf = wave / WT;   // this wave's frame block and row group
  const int t0 = blockIdx.x * TT;
  const int fo0 = blockIdx.y * RF;
  const int n = blockIdx.z / p.co_tiles, tile = blockIdx.z % p.co_tiles;
  const int len = lens ? min(lens[n], p.Tin) : p.Tin;
  const int cout_pad = p.co_tiles * 32;
  const size_t wplane = (size_t)p.KF * p.KT * p.KG * cout_pad * 8;  // elements

  ms::f32x16 acc[CL_F];
#pragma unroll
  for (int f = 0; f < CL_F; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  const int tin0 = t0 * p.ST - p.pad_t;
  const int ngran = p.KG * p.PW;  // granules per staged input row
  const unsigned plane_bytes = WIN ? (unsigned)((size_t)p.N * p.Tin * p.FP * 2) : 0u;
  const __amdgpu_buffer_rsrc_t xh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xh), 0, plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xl_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xl), 0, plane_bytes, 0x00020000);

  for (int kf = 0; kf < p.KF; ++kf) {
    // ---- stage the input rows this kf touches (masked, zero padded)
    for (int f = 0; f < RF; ++f) {
      const int fin = (fo0 + f) * p.SF - p.pad_f + kf * p.DF;
      const bool frow = (fo0 + f) < p.Fout && (WIN || (fin >= 0 && fin < p.Fin));
      const size_t rbase = ((size_t)n * p.Fin + (frow && !WIN ? fin : 0)) * p.Tin;
      for (int i = tid; i < ngran; i += 256) {
        const int q = i / p.KG, kg = i - q * p.KG;  // frame-major so 4 lanes cover one frame's 64 B
        const int tin = tin0 + q;
        u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
        if (frow && tin >= 0 && tin < len) {
          if (WIN) {
            // window of output row fo: FP-strided frame, element offset fo * SF (even => 4-byte aligned 16-byte loads)
            // (buffer loads: a 16-byte global load would be split into four dwords at this alignment)
            const int boff = (int)((((size_t)n * p.Tin + tin) * p.FP + (size_t)(fo0 + f) * p.SF + kg * 8) * 2);
            vh = __builtin_amdgcn_raw_buffer_load_b128(xh_rsrc, boff, 0, 0);
            if (!F16) vl = __builtin_amdgcn_raw_buffer_load_b128(xl_rsrc, boff, 0, 0);
          } else {
            const size_t off = (rbase + tin) * p.Cin + kg * 8;
            vh = *reinterpret_cast<const u32x4*>(xh + off);
            if (!F16) vl = *reinterpret_cast<const u32x4*>(xl + off);
          }
        }
        *reinterpret_cast<u32x4*>(Ph + f * row_bytes + (kg * p.PW + q) * 16) = vh;
        if (!F16) *reinterpret_cast<u32x4*>(Pl + f * row_bytes + (kg * p.PW + q) * 16) = vl;
      }
    }
    // ---- this kf row's filters go through LDS in two halves of the KT taps (keeps the
    // workgroup under 80 KiB so two of them share a CU and hide each other's staging)
    const int taps_per_stage = (p.KT + 1) / 2;
    for (int kt0 = 0; kt0 < p.KT; kt0 += taps_per_stage) {
      const int ntap = min(taps_per_stage, p.KT - kt0);
      if (kt0 > 0) __syncthreads();  // the previous half's reads are done
      {
        const int ng = ntap * p.KG * 32;
        for (int i = tid; i < ng; i += 256) {
          const int co = i & 31, rest = i >> 5;  // rest = kt_local*KG + kg
          const size_t src = ((((size_t)kf * p.KT + kt0) * p.KG + rest) * cout_pad + tile * 32 + co) * 8;
          *reinterpret_cast<u32x4*>(Wh + i * 16) = *reinterpret_cast<const u32x4*>(wp + src);
          if (!F16) *reinterpret_cast<u32x4*>(Wl + i * 16) = *reinterpret_cast<const u32x4*>(wp + wplane + src);
        }
      }
      __syncthreads();
      for (int ktl = 0; ktl < ntap; ++ktl) {
        const int tq = (wt * 32 + l31) * p.ST + (kt0 + ktl) * p.DT;
        for (int s = 0; s < p.KG / 2; ++s) {
          const int kg = 2 * s + half;
          const int woff = ((ktl * p.KG + kg) * 32 + l31) * 16;
          const u32x4 ah = *reinterpret_cast<const u32x4*>(Wh + woff);
          u32x4 al = ah;
          if (!F16) al = *reinterpret_cast<const u32x4*>(Wl + woff);
#pragma unroll
          for (int f = 0; f < CL_F; ++f) {
            const int poff = (wf * CL_F + f) * row_bytes + (kg * p.PW + tq) * 16;
            const u32x4 bh = *reinterpret_cast<const u32x4*>(Ph + poff);
            if (F16) {
              acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc[f], 0, 0, 0);
            } else {
              const u32x4 bl = *reinterpret_cast<const u32x4*>(Pl + poff);
              acc[f] = ms::mfma_32x32x16<HM>(ah, bh, acc[f]);
              acc[f] = ms::mfma_32x32x16<HM>(al, bh, acc[f]);
              acc[f] = ms::mfma_32x32x16<HM>(ah, bl, acc[f]);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  const int t = t0 + wt * 32 + l31;
  const float winv = scale_word[1];
  if (t < p.Tout) {
#pragma unroll
    for (int f = 0; f < CL_F; ++f) {
      const int fo = fo0 + wf * CL_F + f;
      if (fo >= p.Fout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = tile * 32 + ms::mfma32_row(r, lane);
        if (co < p.Cout) {
          float v = acc[f][r] * winv + (bias ? bias[co] : 0.f);   // (winv: the packed weights' 2^-s, exact)
          if (p.act == MS_ACT_CLAMP) v = fminf(fmaxf(v, p.lo), p.hi);
          y[(((size_t)n * p.Cout + co) * p.Fout + fo) * p.Tout + t] = v;
        }
      }
    }
  }
}


// ---- short inputs (round 4): a 320 ms streaming chunk reaches the second DS2 convolution as 16 frames x 40 rows x 32 channels
// per utterance.  The tiled kernel above then runs 192 workgroups whose 32-frame MFMA tiles are half padding and whose
// per-kernel-row staging (three barriers per kf, nothing in flight across them) is fully exposed at less than one workgroup
// per CU: 177 us for 29 GFLOP (profiles/r04e_cfg5_kernel_stats.csv).  This form: Cin = 32 -- one v_mfma_f32_16x16x32 contracts
// all channels of a tap --, a 16-frame x 16-cout MFMA tile per output feature row, every input row a workgroup needs staged
// ONCE (all kf rows: (RB - 1) SF + (KF - 1) DF + 1 rows x (16 + KT - 1) frames x 64 B per plane), no barrier in the tap
// loop: a wave owns every fourth kernel row for all of the block's output rows and streams its filter granules (the MFMA's
// A operand, 16 couts x 32 channels = 1 KB per plane and tap) straight from L2 into registers, a kernel row ahead.
// Workgroup = (utterance, 16 output channels, RB output feature rows); grid sized to about one workgroup per CU.
constexpr int SH_RB = 12;     // output feature rows per workgroup, at most
constexpr int SH_KT = 11;     // time taps held in registers per kernel row (DS2: 11)

template <int P>
__global__ __launch_bounds__(256, 1) void maskconv_cl_short_kernel(const unsigned short* __restrict__ xh,
                                                                   const unsigned short* __restrict__ xl,
                                                                   const int32_t* __restrict__ lens,
                                                                   const unsigned short* __restrict__ wp,
                                                                   const float* __restrict__ bias, float* __restrict__ y, ClP p,
                                                                   int RB, int RIN, const float* __restrict__ scale_word) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  const float winv = scale_word[1];
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int fo0 = blockIdx.x * RB, cb = blockIdx.y, n = blockIdx.z;
  const int PW = p.PW;                                   // staged frames per row: Tout + KT - 1 (ST = DT = 1)
  const int row_bytes = PW * 64;
  char* Ph = lds;
  char* Pl = lds + (size_t)RIN * row_bytes;
  const int len = lens ? min(lens[n], p.Tin) : p.Tin;
  const int fin_lo = fo0 * p.SF - p.pad_f;

  // ---- stage every input row of this block once (masked, zero padded): granule (row, frame, kg), 4 lanes = one frame's 64 B
  for (int i = tid; i < RIN * PW * 4; i += 256) {
    const int kg = i & 3, rq = i >> 2;
    const int r = rq / PW, fq = rq - r * PW;
    const int fin = fin_lo + r, tin = fq - p.pad_t;
    u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
    if (fin >= 0 && fin < p.Fin && tin >= 0 && tin < len) {
      const size_t off = (((size_t)n * p.Fin + fin) * p.Tin + tin) * 32 + kg * 8;
      vh = *reinterpret_cast<const u32x4*>(xh + off);
      if (!F16) vl = *reinterpret_cast<const u32x4*>(xl + off);
    }
    *reinterpret_cast<u32x4*>(Ph + r * row_bytes + fq * 64 + kg * 16) = vh;
    if (!F16) *reinterpret_cast<u32x4*>(Pl + r * row_bytes + fq * 64 + kg * 16) = vl;
  }
  __syncthreads();

  // A wave owns every fourth KERNEL ROW (kf = wave, wave + 4, ...) for ALL output rows of the block: its filter granules are
  // then nobody else's, so the workgroup reads each filter byte once (when the waves split the OUTPUT rows instead, every
  // wave streamed all 462 KB of a cout-half's filters: 473 MB out of L2 per call, and the call ran at the L2's pace, 115 us).
  // The four partial sums per output meet in LDS at the end and are added in wave order.
  ms::f32x4 acc[SH_RB];
#pragma unroll
  for (int i = 0; i < SH_RB; ++i) acc[i] = ms::f32x4{0.f, 0.f, 0.f, 0.f};
  const int cout_pad = p.co_tiles * 32;
  const size_t wplane = (size_t)p.KF * p.KT * 4 * cout_pad * 8;            // elements
  const unsigned short* wa = wp + (size_t)q * cout_pad * 8 + (cb * 16 + c16) * 8;   // this lane's granule of tap 0
  const size_t tap_stride = (size_t)4 * cout_pad * 8;
  // the filters of one kernel row (KT taps x hi, lo) sit in registers and the wave's NEXT row is requested before this row's
  // MFMAs (up to 12 output rows x KT taps x 3: thousands of cycles, an L2 round trip is a few hundred)
  u32x4 fh[2][SH_KT], fl[2][SH_KT];
  auto load_row = [&](auto set, int kf) {
#pragma unroll
    for (int kt = 0; kt < SH_KT; ++kt)
      if (kt < p.KT) {
        const unsigned short* src = wa + (size_t)(kf * p.KT + kt) * tap_stride;
        fh[set][kt] = *reinterpret_cast<const u32x4*>(src);
        if (!F16) fl[set][kt] = *reinterpret_cast<const u32x4*>(src + wplane);
      }
  };
  auto compute_row = [&](auto set, int kf) {
#pragma unroll
    for (int i = 0; i < SH_RB; ++i) {
      const int fo = fo0 + i;
      const int fin = fo * p.SF - p.pad_f + kf * p.DF;
      if (i < RB && fo < p.Fout && fin >= 0 && fin < p.Fin) {     // wave-uniform
        const int rbase = (fin - fin_lo) * row_bytes + c16 * 64 + q * 16;
#pragma unroll
        for (int kt = 0; kt < SH_KT; ++kt)
          if (kt < p.KT) {
            const u32x4 bh = *reinterpret_cast<const u32x4*>(Ph + rbase + kt * 64);
            if (F16) {
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fh[set][kt]), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
            } else {
              const u32x4 bl = *reinterpret_cast<const u32x4*>(Pl + rbase + kt * 64);
              acc[i] = ms::mfma_16x16x32<HM>(fh[set][kt], bh, acc[i]);
              acc[i] = ms::mfma_16x16x32<HM>(fl[set][kt], bh, acc[i]);
              acc[i] = ms::mfma_16x16x32<HM>(fh[set][kt], bl, acc[i]);
            }
          }
      }
    }
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  if (wave < p.KF) load_row(S0, wave);
  for (int kf = wave; kf < p.KF; kf += 8) {
    if (kf + 4 < p.KF) load_row(S1, kf + 4);
    compute_row(S0, kf);
    if (kf + 8 < p.KF) load_row(S0, kf + 8);
    if (kf + 4 < p.KF) compute_row(S1, kf + 4);
  }

  // ---- the four waves' partial sums: [wave][row][lane] float4 in the (now free) staging area, added in wave order
  __syncthreads();
  ms::f32x4* part = reinterpret_cast<ms::f32x4*>(lds);
#pragma unroll
  for (int i = 0; i < SH_RB; ++i)
    if (i < RB) part[(wave * SH_RB + i) * 64 + lane] = acc[i];
  __syncthreads();
  // D[cout = 4 q + r][frame = c16]; wave w finishes rows w, w + 4, w + 8
#pragma unroll
  for (int j = 0; j < SH_RB / 4; ++j) {
    const int i = wave + 4 * j, fo = fo0 + i;
    if (i >= RB || fo >= p.Fout) continue;
    ms::f32x4 v = part[(0 * SH_RB + i) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += part[(w * SH_RB + i) * 64 + lane];
    if (c16 < p.Tout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cb * 16 + 4 * q + r;
        if (co < p.Cout) {
          float o = v[r] * winv + (bias ? bias[co] : 0.f);
          if (p.act == MS_ACT_CLAMP) o = fminf(fmaxf(o, p.lo), p.hi);
          y[(((size_t)n * p.Cout + co) * p.Fout + fo) * p.Tout + c16] = o;
        }
      }
    }
  }
}

// Does the short-input form serve this problem, and with which block of output rows?  (MS_CONV_SHORT=0: never; tests and A/B runs)
int conv_cl_short_plan(const ClP& p, bool f16, int* RB, int* RIN, size_t* lds) {
  static const bool off = getenv("MS_CONV_SHORT") && getenv("MS_CONV_SHORT")[0] == '0';
  if (off || p.Cin != 32 || p.Tout > 16 || p.ST != 1 || p.DT != 1 || p.FP != 0 || p.N > 65535 || p.KT > SH_KT) return 0;
  const int cbs = ms::cdiv(p.Cout, 16);
  int nblk = std::max(1, std::min(ms::cdiv(p.Fout, 4), ms::cdiv(ms::num_cus(), p.N * cbs)));
  while (ms::cdiv(p.Fout, nblk) > SH_RB) ++nblk;
  for (; nblk <= p.Fout; ++nblk) {
    const int rb = ms::cdiv(p.Fout, nblk);
    const int rin = (rb - 1) * p.SF + (p.KF - 1) * p.DF + 1;
    const size_t bytes = std::max((size_t)(f16 ? 1 : 2) * rin * (p.Tout + p.KT - 1) * 64, (size_t)4 * SH_RB * 64 * 16);
    if (bytes <= 156 * 1024) { *RB = rb; *RIN = rin; *lds = bytes; return 1; }
  }
  return 0;
}

// f32 NCHW [N][C][F][T] -> channels-last bf16 hi / lo planes [N][F][T][C]  (C % 8 == 0)
__global__ void nchw_to_cl_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi,
                                        unsigned short* __restrict__ lo, int C, int F, int T, int prec) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z / F, f = blockIdx.z % F;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < T) ? x[(((size_t)n * C + c) * F + f) * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < C) {
      const float v = tile[tx][i];
      const size_t o = (((size_t)n * F + f) * T + t) * C + c;
      unsigned h, l;
      split_by_prec(v, prec, h, l);
      hi[o] = (unsigned short)h;
      if (prec != ms::PREC_F16) lo[o] = (unsigned short)l;
    }
  }
}

// single-channel input x [N][Fin][T] f32 -> bf16 hi / lo planes [N][T][FP], element j = feature j - pad (zeros outside)
__global__ void ft_to_tf_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi,
                                      unsigned short* __restrict__ lo, int Fin, int T, int FP, int pad, int prec) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int t0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  for (int i = ty; i < 32; i += 8) {
    const int fidx = j0 + i - pad, t = t0 + tx;
    tile[i][tx] = (fidx >= 0 && fidx < Fin && t < T) ? x[((size_t)n * Fin + fidx) * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, j = j0 + tx;
    if (t < T && j < FP) {
      const float v = tile[tx][i];
      const size_t o = ((size_t)n * T + t) * FP + j;
      unsigned h, l;
      split_by_prec(v, prec, h, l);
      hi[o] = (unsigned short)h;
      if (prec != ms::PREC_F16) lo[o] = (unsigned short)l;
    }
  }
}

// packed[plane][kt][kg][cout_pad][8] <- w[cout][0][kf = 8 kg + e][kt]  (zero for kf >= KF)
__global__ void conv_fwin_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int Cout, int KF,
                                      int KT, int KG, int cout_pad, int prec, const float* __restrict__ scale_word) {
  const size_t plane = (size_t)KT * KG * cout_pad * 8;
  const float wscale = scale_word[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 7;
    const int co = (i >> 3) % cout_pad;
    const int kg = (i / ((size_t)8 * cout_pad)) % KG;
    const int kt = i / ((size_t)8 * cout_pad * KG);
    const int kf = kg * 8 + e;
    float x = 0.f;
    if (co < Cout && kf < KF) x = w[((size_t)co * KF + kf) * KT + kt] * wscale;
    unsigned h, l;
    split_by_prec(x, prec, h, l);
    packed[i] = (unsigned short)h;
    packed[plane + i] = (unsigned short)l;
  }
}

// Tile shape of maskconv_cl_kernel for this problem: 4 = 32 frames x 8 rows per workgroup (short inputs), 1 = 128 frames x 2
// rows, 0 = neither fits LDS / the grid.  Fills p.PW and *lds.
int conv_cl_plan(ClP& p, size_t* lds) {
  const size_t wbytes = (size_t)2 * ((p.KT + 1) / 2) * p.KG * 32 * 16;
  if ((long)p.N * p.co_tiles > 65535) return 0;
  for (int wf : {p.Tout <= 48 ? 4 : 1, 1}) {
    const int tt = 128 / wf, rf = CL_F * wf;
    p.PW = (tt - 1) * p.ST + (p.KT - 1) * p.DT + 1;
    *lds = (size_t)2 * rf * p.KG * p.PW * 16 + wbytes;
    if (*lds <= 160 * 1024 && ms::cdiv(p.Fout, rf) <= 65535) return wf;
  }
  return 0;
}

template <bool WIN>
int conv_cl_launch(const ClP& p, int wf, size_t lds, const unsigned short* xh, const unsigned short* xl, const int32_t* lens,
                   const void* packed_w, const float* bias, float* y, hipStream_t stream, const float* scale_word) {
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
#define MS_CL_ATTR(PP)                                                                                                                     \
    MS_HIP(hipFuncSetAttribute((const void*)maskconv_cl_kernel<PP, WIN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   \
    MS_HIP(hipFuncSetAttribute((const void*)maskconv_cl_kernel<PP, WIN, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MS_CL_ATTR(ms::PREC_BF16X3) MS_CL_ATTR(ms::PREC_F16) MS_CL_ATTR(ms::PREC_F16X3)
#undef MS_CL_ATTR
    attr_once.done();
  }
  const int prec = ms::split_mode();
  const int tt = 128 / wf, rf = CL_F * wf;
  dim3 grid(ms::cdiv(p.Tout, tt), ms::cdiv(p.Fout, rf), p.N * p.co_tiles);
  const unsigned short* wq = (const unsigned short*)packed_w;
#define MS_CL_LAUNCH(WF_)                                                                                                             \
  if (prec == ms::PREC_F16) hipLaunchKernelGGL((maskconv_cl_kernel<ms::PREC_F16, WIN, WF_>), grid, dim3(256), lds, stream, xh, xl, lens, wq, bias, y, p, scale_word); \
  else if (prec == ms::PREC_F16X3) hipLaunchKernelGGL((maskconv_cl_kernel<ms::PREC_F16X3, WIN, WF_>), grid, dim3(256), lds, stream, xh, xl, lens, wq, bias, y, p, scale_word); \
  else hipLaunchKernelGGL((maskconv_cl_kernel<ms::PREC_BF16X3, WIN, WF_>), grid, dim3(256), lds, stream, xh, xl, lens, wq, bias, y, p, scale_word);
  if (wf == 4) { MS_CL_LAUNCH(4) } else { MS_CL_LAUNCH(1) }
#undef MS_CL_LAUNCH
  MS_LAUNCH_CHECK();
  return MS_OK;
}

inline int fwin_kfp(int KF) { return ms::cdiv(KF, 16) * 16; }
inline int fwin_fp(int KF, int SF, int Fout) { return ms::cdiv((Fout - 1) * SF + fwin_kfp(KF), 32) * 32; }

}  // namespace

// ---- single-channel convolutions (DS2 conv1: 1 -> 32 channels, 41 x 11 taps) as the same split-bf16 implicit GEMM: the
// KF feature rows under an output row play the role of the input channels (padded to a multiple of 16), the time taps
// stay taps.  Input planes are [N][Tin][FP] (feature-contiguous, zero feature padding materialised).
// (the two planes, then the weights' scale word {2^s, 2^-s}: common.h "per-tensor power-of-two scale")
static size_t fwin_plane_bytes(int Cout, int KF, int KT) {
  return ms::align_up((size_t)2 * KT * (fwin_kfp(KF) / 8) * (ms::cdiv(Cout, 32) * 32) * 8 * sizeof(unsigned short), 256);
}
extern "C" size_t ms_maskconv_fwin_packed_bytes(int Cout, int KF, int KT) {
  if (Cout <= 0 || KF <= 0 || KT <= 0) return 0;
  return fwin_plane_bytes(Cout, KF, KT) + 256;
}

extern "C" int ms_maskconv_fwin_pack(const float* w, void* packed, int Cout, int KF, int KT, void* stream) {
  MS_REQUIRE(w && packed, "null pointer");
  MS_REQUIRE(Cout > 0 && KF > 0 && KT > 0, "bad shape");
  const int cout_pad = ms::cdiv(Cout, 32) * 32, KG = fwin_kfp(KF) / 8;
  const size_t plane = (size_t)KT * KG * cout_pad * 8;
  const int blocks = (int)std::min<size_t>((plane + 255) / 256, 2048);
  float* scale_word = (float*)((char*)packed + fwin_plane_bytes(Cout, KF, KT));
  const int rc = ms::weight_scale_launch(w, (size_t)Cout * KF * KT, scale_word, ms::split_mode(), (hipStream_t)stream);
  if (rc != MS_OK) return rc;
  hipLaunchKernelGGL(conv_fwin_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed, Cout,
                     KF, KT, KG, cout_pad, ms::split_mode(), scale_word);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" size_t ms_maskconv_fwin_workspace_bytes(int N, int Tin, int KF, int SF, int Fout) {
  if (N <= 0 || Tin <= 0 || KF <= 0 || SF <= 0 || Fout <= 0) return 0;
  return ms::align_up((size_t)N * Tin * fwin_fp(KF, SF, Fout) * 4, 256);  // hi + lo planes
}

extern "C" int ms_maskconv_fwin_forward(const float* x, const int32_t* lens, const void* packed_w, const float* bias, float* y,
                                        int N, int Fin, int Tin, int Cout, int Fout, int Tout, int KF, int KT, int SF, int ST,
                                        int DT, int pad_f_l, int pad_t_l, int act, float act_lo, float act_hi, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
  ms::ProfScope prof_span(MS_PROF_CONV, (hipStream_t)stream_);
  MS_REQUIRE(x && packed_w && y && workspace, "null pointer");
  MS_REQUIRE(N > 0 && Fin > 0 && Tin > 0 && Cout > 0 && Fout > 0 && Tout > 0, "bad shape");
  MS_REQUIRE(KF > 0 && KT > 0 && SF > 0 && ST > 0 && DT > 0 && pad_f_l >= 0 && pad_t_l >= 0, "bad kernel");
  MS_REQUIRE(SF % 2 == 0, "feature stride must be even (4-byte aligned window loads)");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  if (workspace_bytes < ms_maskconv_fwin_workspace_bytes(N, Tin, KF, SF, Fout)) {
    ms::set_error("ms_maskconv_fwin_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  ClP p;
  p.N = N; p.Cin = fwin_kfp(KF); p.Fin = 1; p.Tin = Tin; p.Cout = Cout; p.Fout = Fout; p.Tout = Tout; p.KF = 1; p.KT = KT;
  p.SF = SF; p.ST = ST; p.DF = 1; p.DT = DT; p.pad_f = 0; p.pad_t = pad_t_l; p.KG = p.Cin / 8;
  p.co_tiles = ms::cdiv(Cout, 32);
  p.act = act; p.lo = act_lo; p.hi = act_hi;
  p.FP = fwin_fp(KF, SF, Fout);
  size_t lds = 0;
  const int wf = conv_cl_plan(p, &lds);
  if (wf == 0 || (size_t)N * Tin * p.FP * 2 >= ((size_t)1 << 31)) {
    ms::set_error("ms_maskconv_fwin_forward: shape outside the LDS / grid budget");
    return MS_ERR_UNSUPPORTED;
  }
  unsigned short* xh = (unsigned short*)workspace;
  unsigned short* xl = xh + (size_t)N * Tin * p.FP;
  const int prec = ms::split_mode();
  const bool f16 = prec == ms::PREC_F16;
  hipLaunchKernelGGL(ft_to_tf_split_kernel, dim3(ms::cdiv(Tin, 32), ms::cdiv(p.FP, 32), N), dim3(32, 8), 0, stream, x, xh, xl,
                     Fin, Tin, p.FP, pad_f_l, prec);
  MS_LAUNCH_CHECK();
  return conv_cl_launch<true>(p, wf, lds, xh, xl, lens, packed_w, bias, y, stream,
                              (const float*)((const char*)packed_w + fwin_plane_bytes(Cout, KF, KT)));
}

static size_t cl_plane_bytes(int Cout, int Cin, int KF, int KT) {
  return ms::align_up((size_t)2 * KF * KT * (Cin / 8) * (ms::cdiv(Cout, 32) * 32) * 8 * sizeof(unsigned short), 256);
}
extern "C" size_t ms_maskconv_cl_packed_bytes(int Cout, int Cin, int KF, int KT) {
  if (Cout <= 0 || Cin <= 0 || Cin % 16 || KF <= 0 || KT <= 0) return 0;
  return cl_plane_bytes(Cout, Cin, KF, KT) + 256;      // + the weights' scale word
}

extern "C" int ms_maskconv_cl_pack(const float* w, void* packed, int Cout, int Cin, int KF, int KT, void* stream) {
  MS_REQUIRE(w && packed, "null pointer");
  MS_REQUIRE(Cout > 0 && Cin > 0 && Cin % 16 == 0 && KF > 0 && KT > 0, "bad shape (Cin must be a multiple of 16)");
  const int cout_pad = ms::cdiv(Cout, 32) * 32;
  const size_t plane = (size_t)KF * KT * (Cin / 8) * cout_pad * 8;
  const int blocks = (int)std::min<size_t>((plane + 255) / 256, 2048);
  float* scale_word = (float*)((char*)packed + cl_plane_bytes(Cout, Cin, KF, KT));
  const int rc = ms::weight_scale_launch(w, (size_t)Cout * Cin * KF * KT, scale_word, ms::split_mode(), (hipStream_t)stream);
  if (rc != MS_OK) return rc;
  hipLaunchKernelGGL(conv_cl_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed, Cout,
                     Cin, KF, KT, cout_pad, ms::split_mode(), scale_word);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" size_t ms_maskconv_cl_workspace_bytes(int N, int Cin, int Fin, int Tin) {
  if (N <= 0 || Cin <= 0 || Fin <= 0 || Tin <= 0) return 0;
  return ms::align_up((size_t)N * Cin * Fin * Tin * 4, 256);  // hi + lo planes of the input
}

extern "C" int ms_maskconv_cl_forward(const float* x, const int32_t* lens, const void* packed_w, const float* bias, float* y,
                                      int N, int Cin, int Fin, int Tin, int Cout, int Fout, int Tout, int KF, int KT,
                                      int SF, int ST, int DF, int DT, int pad_f_l, int pad_t_l, int act, float act_lo,
                                      float act_hi, void* workspace, size_t workspace_bytes, void* stream_) {
  ms::ProfScope prof_span(MS_PROF_CONV, (hipStream_t)stream_);
  MS_REQUIRE(x && packed_w && y && workspace, "null pointer");
  MS_REQUIRE(N > 0 && Cin > 0 && Cin % 16 == 0 && Fin > 0 && Tin > 0 && Cout > 0 && Fout > 0 && Tout > 0, "bad shape");
  MS_REQUIRE(KF > 0 && KT > 0 && SF > 0 && ST > 0 && DF > 0 && DT > 0 && pad_f_l >= 0 && pad_t_l >= 0, "bad kernel");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  if (workspace_bytes < ms_maskconv_cl_workspace_bytes(N, Cin, Fin, Tin)) {
    ms::set_error("ms_maskconv_cl_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  ClP p;
  p.N = N; p.Cin = Cin; p.Fin = Fin; p.Tin = Tin; p.Cout = Cout; p.Fout = Fout; p.Tout = Tout; p.KF = KF; p.KT = KT;
  p.SF = SF; p.ST = ST; p.DF = DF; p.DT = DT; p.pad_f = pad_f_l; p.pad_t = pad_t_l; p.KG = Cin / 8;
  p.co_tiles = ms::cdiv(Cout, 32);
  p.act = act; p.lo = act_lo; p.hi = act_hi;
  p.FP = 0;
  size_t lds = 0;
  const int wf = conv_cl_plan(p, &lds);
  if (wf == 0) {
    ms::set_error("ms_maskconv_cl_forward: shape outside the LDS / grid budget");
    return MS_ERR_UNSUPPORTED;
  }
  const float* scale_word = (const float*)((const char*)packed_w + cl_plane_bytes(Cout, Cin, KF, KT));
  unsigned short* xh = (unsigned short*)workspace;
  unsigned short* xl = xh + (size_t)N * Cin * Fin * Tin;
  MS_REQUIRE(N * Fin <= 65535, "N*Fin exceeds grid limits");
  const int prec = ms::split_mode();
  const bool f16 = prec == ms::PREC_F16;
  hipLaunchKernelGGL(nchw_to_cl_split_kernel, dim3(ms::cdiv(Tin, 32), ms::cdiv(Cin, 32), N * Fin), dim3(32, 8), 0, stream, x,
                     xh, xl, Cin, Fin, Tin, prec);
  MS_LAUNCH_CHECK();
  {
    int RB = 0, RIN = 0;
    size_t slds = 0;
    if (conv_cl_short_plan(p, f16, &RB, &RIN, &slds)) {
      static ms::DeviceOnce attr_once;
      if (attr_once.need()) {
        MS_HIP(hipFuncSetAttribute((const void*)maskconv_cl_short_kernel<ms::PREC_BF16X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MS_HIP(hipFuncSetAttribute((const void*)maskconv_cl_short_kernel<ms::PREC_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MS_HIP(hipFuncSetAttribute((const void*)maskconv_cl_short_kernel<ms::PREC_F16X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done();
      }
      ClP ps = p;
      ps.PW = Tout + KT - 1;
      const dim3 grid(ms::cdiv(Fout, RB), ms::cdiv(Cout, 16), N);
      const unsigned short* wq = (const unsigned short*)packed_w;
      auto kern = prec == ms::PREC_F16 ? maskconv_cl_short_kernel<ms::PREC_F16>
                  : prec == ms::PREC_F16X3 ? maskconv_cl_short_kernel<ms::PREC_F16X3> : maskconv_cl_short_kernel<ms::PREC_BF16X3>;
      hipLaunchKernelGGL(kern, grid, dim3(256), slds, stream, xh, xl, lens, wq, bias, y, ps, RB, RIN, scale_word);
      MS_LAUNCH_CHECK();
      return MS_OK;
    }
  }
  return conv_cl_launch<false>(p, wf, lds, xh, xl, lens, packed_w, bias, y, stream, scale_word);
}
