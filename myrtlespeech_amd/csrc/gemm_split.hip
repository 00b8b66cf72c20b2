// y[M,N] = x[M,K] . w[N,K]^T + bias[N] with float32 operands split into bf16 (hi, lo) pairs:
//   x.w ~= x_hi.w_hi + x_lo.w_hi + x_hi.w_lo      (f32 accumulate, relative error ~2^-17 per
// product instead of 2^-24; measured end to end in DESIGN.md).  Three v_mfma_f32_32x32x16_bf16
// per 16-deep k-step run at 3/16 of the cost of the eight exact-f32 MFMAs they replace.
//
// Used for the LSTM input projection (rnn.hip), the dominant GEMM of the encoder
// (torch.nn.LSTM's x.W_ih^T inside rnn.py:177).
//
//   split_planes_kernel : x f32 [M,K]  ->  hi, lo bf16 [M,K]                (HBM-bound)
//   gemm_nt_bf16x3_kernel: 256x128 block tile, BK = 32, 4 waves (each 64 rows x 128 cols =
//       2x4 MFMA tiles), all four operand planes of a K-block staged once in LDS as
//       [k/8][row][8 bf16] granules (one conflict-free ds_read_b128 = one MFMA operand),
//       register prefetch of the next K-block, XCD-aware tile order, bias epilogue.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace ms {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_hi_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

__device__ __forceinline__ unsigned f16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }

// fp16 mode: one plane of round-to-nearest fp16 values
__global__ void to_f16_plane_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    u32x2 ho = {f16_bits(v[0]) | (f16_bits(v[1]) << 16), f16_bits(v[2]) | (f16_bits(v[3]) << 16)};
    reinterpret_cast<u32x2*>(hi)[i] = ho;
  }
}

template <bool HM>
__global__ void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi,
                                    unsigned short* __restrict__ lo, size_t n4) {
  // n4 = number of 4-element groups
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) plane_split<HM>(v[e], h[e], l[e]);
    u32x2 ho = {h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
    u32x2 lo2 = {l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    reinterpret_cast<u32x2*>(hi)[i] = ho;
    reinterpret_cast<u32x2*>(lo)[i] = lo2;
  }
}

constexpr int SB_M = 256, SB_N = 128, SB_K = 32;
constexpr int A_PLANE = SB_M * 16 + 32;  // bytes per k-group plane (+32 B: conflict-free ds_write_b128)
constexpr int B_PLANE = SB_N * 16 + 32;
constexpr int A_TILE = 4 * A_PLANE, B_TILE = 4 * B_PLANE;
constexpr int SPLIT_LDS = 2 * A_TILE + 2 * B_TILE;

// Rows past the matrix edge are clamped to the last row (their products are never stored), so
// the loads need no predicate; K % 32 == 0 is a launch precondition.
// (ld: elements between consecutive rows, >= K: an operand may be a column block of a wider matrix -- the forward / backward
// half of a bidirectional layer's output planes and of the next layer's W_ih, rnn.hip "K-halves")
__device__ __forceinline__ u32x4 ld_granule(const unsigned short* __restrict__ p, int rows, int ld, int row, int k) {
  return *reinterpret_cast<const u32x4*>(p + (size_t)min(row, rows - 1) * ld + k);
}

template <bool HM>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x3_kernel(const unsigned short* __restrict__ Ah,
                                                                const unsigned short* __restrict__ Al,
                                                                const unsigned short* __restrict__ Wh,
                                                                const unsigned short* __restrict__ Wl,
                                                                const float* __restrict__ bias, float* __restrict__ Y,
                                                                int M, int K, int N, int act, float lo, float hi, int lda, int ldw) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* sAh = lds;
  char* sAl = lds + A_TILE;
  char* sBh = lds + 2 * A_TILE;
  char* sBl = lds + 2 * A_TILE + B_TILE;

  const int nbn = (N + SB_N - 1) / SB_N, nbm = (M + SB_M - 1) / SB_M;
  const int nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int bn = bid % nbn, bm = bid / nbn;
  const int m0 = bm * SB_M, n0 = bn * SB_N;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int kg = tid & 3, r_in = tid >> 2;  // loader mapping: 4 k-groups x 64 rows per pass

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 rah[4], ral[4], rbh[2], rbl[2];
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rah[i] = ld_granule(Ah, M, lda, m0 + r_in + 64 * i, k0 + kg * 8);
      ral[i] = ld_granule(Al, M, lda, m0 + r_in + 64 * i, k0 + kg * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rbh[i] = ld_granule(Wh, N, ldw, n0 + r_in + 64 * i, k0 + kg * 8);
      rbl[i] = ld_granule(Wl, N, ldw, n0 + r_in + 64 * i, k0 + kg * 8);
    }
  };
  auto store_regs = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(sAh + kg * A_PLANE + (r_in + 64 * i) * 16) = rah[i];
      *reinterpret_cast<u32x4*>(sAl + kg * A_PLANE + (r_in + 64 * i) * 16) = ral[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(sBh + kg * B_PLANE + (r_in + 64 * i) * 16) = rbh[i];
      *reinterpret_cast<u32x4*>(sBl + kg * B_PLANE + (r_in + 64 * i) * 16) = rbl[i];
    }
  };

  const int nk = (K + SB_K - 1) / SB_K;
  load_regs(0);
  store_regs();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_regs((kt + 1) * SB_K);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int g = 2 * s + half;
      u32x4 ah[2], al[2], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = g * A_PLANE + (wave * 64 + i * 32 + l31) * 16;
        ah[i] = *reinterpret_cast<const u32x4*>(sAh + off);
        al[i] = *reinterpret_cast<const u32x4*>(sAl + off);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int off = g * B_PLANE + (j * 32 + l31) * 16;
        bh[j] = *reinterpret_cast<const u32x4*>(sBh + off);
        bl[j] = *reinterpret_cast<const u32x4*>(sBl + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma_32x32x16<HM>(ah[i], bh[j], acc[i][j]);
          acc[i][j] = mfma_32x32x16<HM>(al[i], bh[j], acc[i][j]);
          acc[i][j] = mfma_32x32x16<HM>(ah[i], bl[j], acc[i][j]);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_regs();
      __syncthreads();
    }
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j * 32 + l31;
    const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 64 + i * 32 + mfma32_row(r, lane);
        if (m < M && n < N) {
          float v = acc[i][j][r] + bv;
          if (act == MS_ACT_CLAMP) v = fminf(fmaxf(v, lo), hi);
          Y[(size_t)m * N + n] = v;
        }
      }
  }
}

// ---- 64 x 64 x 32, 4 waves of 32 x 32 (round 4): outputs too small to give the 256-row tiles a workgroup on most CUs -- a
// streaming chunk's hidden FC layer is 512 x 1 024 (16 tiles of 256 x 128: 144 us on the kernel above, 9 % of a chunk of the
// shipped architecture) -- get 64 x 64 tiles: 128 .. 512 workgroups, each K-block 6 MFMAs per wave between ONE barrier.
// Operands go global -> registers -> LDS; the registers form a ring of four K-blocks (a block's loads are issued three
// blocks before it is consumed: an L2 round trip is longer than a block's 192 MFMA cycles) and the LDS has two stages.
// Fragment layout, operand roles and the per-element order of the products (hi.hi, lo.hi, hi.lo per 16-deep half, halves and
// blocks in k order) are gemm_nt_bf16x3_kernel's: bit-identical results (tests/test_gpu_gemm.py).
constexpr int T6 = 64;
constexpr int T6_PLANE = T6 * 16 + 32;            // bytes per k-group plane (+32 B: conflict-free ds_write_b128)
constexpr int T6_TILE = 4 * T6_PLANE;             // one operand plane (hi or lo) of a K-block
constexpr int T6_STAGE = 4 * T6_TILE;             // x hi, x lo, W hi, W lo
constexpr int T6_LDS = 2 * T6_STAGE;

template <int P = PREC_BF16X3>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x3_tile64_kernel(const unsigned short* __restrict__ Ah,
                                                                       const unsigned short* __restrict__ Al,
                                                                       const unsigned short* __restrict__ Wh,
                                                                       const unsigned short* __restrict__ Wl,
                                                                       const float* __restrict__ bias, float* __restrict__ Y,
                                                                       int M, int K, int N, int act, float lo, float hi, int lda, int ldw) {
  constexpr bool F16 = prec_one_plane(P), HM = P == PREC_F16X3;
  __shared__ __attribute__((aligned(16))) char lds[T6_LDS];
  const int nbn = (N + T6 - 1) / T6, nbm = (M + T6 - 1) / T6;
  const int nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int bn = bid % nbn, bm = bid / nbn;     // consecutive workgroups of an XCD share the x rows
  const int m0 = bm * T6, n0 = bn * T6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int kg = tid & 3, r_in = tid >> 2;      // loader: thread = (row r_in of the tile, k-group kg), both operands, both planes

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const unsigned short* pa_h = Ah + (size_t)min(m0 + r_in, M - 1) * lda + kg * 8;
  const unsigned short* pa_l = Al + (size_t)min(m0 + r_in, M - 1) * lda + kg * 8;
  const unsigned short* pb_h = Wh + (size_t)min(n0 + r_in, N - 1) * ldw + kg * 8;
  const unsigned short* pb_l = Wl + (size_t)min(n0 + r_in, N - 1) * ldw + kg * 8;
  const int nk = K / SB_K;
  u32x4 ring[4][4];                             // [block % 4][x hi, x lo, W hi, W lo]
  auto load_block = [&](int kb, u32x4 (&r)[4]) {
    const int k0 = min(kb, nk - 1) * SB_K;      // past the end: the last block again (never stored)
    r[0] = *reinterpret_cast<const u32x4*>(pa_h + k0);
    r[2] = *reinterpret_cast<const u32x4*>(pb_h + k0);
    if (!F16) {                                  // fp16 mode (MS_PRECISION=fp16): one plane per operand, one MFMA per half
      r[1] = *reinterpret_cast<const u32x4*>(pa_l + k0);
      r[3] = *reinterpret_cast<const u32x4*>(pb_l + k0);
    }
  };
  auto store_block = [&](int stage, const u32x4 (&r)[4]) {
    char* st = lds + stage * T6_STAGE + kg * T6_PLANE + r_in * 16;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (!F16 || (p & 1) == 0) *reinterpret_cast<u32x4*>(st + p * T6_TILE) = r[p];
  };
  auto compute = [&](int stage) {
    const char* st = lds + stage * T6_STAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int g = 2 * s2 + half;
      const int offa = g * T6_PLANE + (wm * 32 + l31) * 16, offb = g * T6_PLANE + (wn * 32 + l31) * 16;
      if (F16) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + offa));
        const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + 2 * T6_TILE + offb));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        continue;
      }
      const u32x4 ah = *reinterpret_cast<const u32x4*>(st + offa);
      const u32x4 al = *reinterpret_cast<const u32x4*>(st + T6_TILE + offa);
      const u32x4 bh = *reinterpret_cast<const u32x4*>(st + 2 * T6_TILE + offb);
      const u32x4 bl = *reinterpret_cast<const u32x4*>(st + 3 * T6_TILE + offb);
      acc = mfma_32x32x16<HM>(ah, bh, acc);
      acc = mfma_32x32x16<HM>(al, bh, acc);
      acc = mfma_32x32x16<HM>(ah, bl, acc);
    }
  };
  load_block(0, ring[0]);
  load_block(1, ring[1]);
  load_block(2, ring[2]);
  store_block(0, ring[0]);
  __syncthreads();
  // iteration kt: request block kt + 3, compute block kt from stage kt & 1, put block kt + 1 into the other stage (free since
  // the barrier that ended iteration kt - 1), barrier.  Unrolled by four so that the ring's slots are static registers.
  for (int kb = 0; kb < nk; kb += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kt = kb + u;
      if (kt < nk) {
        load_block(kt + 3, ring[(u + 3) & 3]);
        compute(kt & 1);
        if (kt + 1 < nk) store_block((kt + 1) & 1, ring[(u + 1) & 3]);
        __syncthreads();
      }
    }
  }
  const int n = n0 + wn * 32 + l31;
  const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + mfma32_row(r, lane);
    if (m < M && n < N) {
      float v = acc[r] + bv;
      if (act == MS_ACT_CLAMP) v = fminf(fmaxf(v, lo), hi);
      Y[(size_t)m * N + n] = v;
    }
  }
}

// ---- 256x256 block tile, 8 waves (4 along M x 2 along N, each 64x128), BK = 32, LDS double buffered:
// per K-block one barrier; the next block's ds_writes and the block-after-next's global loads are
// issued ahead of the MFMAs of the current block.
constexpr int S2_M = 256, S2_N = 256;
constexpr int S2_PLANE = 256 * 16 + 32;
constexpr int S2_TILE = 4 * S2_PLANE;           // one operand plane-set (hi or lo) of one matrix
constexpr int S2_BUF = 4 * S2_TILE;             // A_hi, A_lo, B_hi, B_lo
constexpr int SPLIT2_LDS = 2 * S2_BUF;

template <int P>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16x3_kernel2(const unsigned short* __restrict__ Ah,
                                                                 const unsigned short* __restrict__ Al,
                                                                 const unsigned short* __restrict__ Wh,
                                                                 const unsigned short* __restrict__ Wl,
                                                                 const float* __restrict__ bias, float* __restrict__ Y,
                                                                 int M, int K, int N, int act, float lo, float hi, int lda, int ldw) {
  constexpr bool F16 = prec_one_plane(P), HM = P == PREC_F16X3;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int nbn = (N + S2_N - 1) / S2_N, nbm = (M + S2_M - 1) / S2_M;
  const int nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  // grouped order: 32 consecutive tiles (one XCD's CUs) form a 4 x 8 patch, so every A row-block is
  // shared by 8 tiles and every W column-block by 4 while they sweep K together (L2 reuse)
  constexpr int GM = 4;
  const int per_group = GM * nbn;
  const int first_m = (bid / per_group) * GM;
  const int gm = min(GM, nbm - first_m);
  const int bm = first_m + (bid % per_group) % gm;
  const int bn = (bid % per_group) / gm;
  const int m0 = bm * S2_M, n0 = bn * S2_N;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int kg = tid & 3, r_in = tid >> 2;  // 4 k-groups x 128 rows per pass

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 rah[2], ral[2], rbh[2], rbl[2];
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rah[i] = ld_granule(Ah, M, lda, m0 + r_in + 128 * i, k0 + kg * 8);
      rbh[i] = ld_granule(Wh, N, ldw, n0 + r_in + 128 * i, k0 + kg * 8);
      if (!F16) {
        ral[i] = ld_granule(Al, M, lda, m0 + r_in + 128 * i, k0 + kg * 8);
        rbl[i] = ld_granule(Wl, N, ldw, n0 + r_in + 128 * i, k0 + kg * 8);
      }
    }
  };
  auto store_regs = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = kg * S2_PLANE + (r_in + 128 * i) * 16;
      *reinterpret_cast<u32x4*>(buf + off) = rah[i];
      *reinterpret_cast<u32x4*>(buf + 2 * S2_TILE + off) = rbh[i];
      if (!F16) {
        *reinterpret_cast<u32x4*>(buf + S2_TILE + off) = ral[i];
        *reinterpret_cast<u32x4*>(buf + 3 * S2_TILE + off) = rbl[i];
      }
    }
  };

  const int nk = K / SB_K;
  load_regs(0);
  store_regs(lds);
  if (nk > 1) load_regs(SB_K);
  __syncthreads();

  u32x4 ah[2], al[2], bh[4], bl[4];
  auto read_ops = [&](const char* cur, int s) {
    const int g = 2 * s + half;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = g * S2_PLANE + (wm * 64 + i * 32 + l31) * 16;
      ah[i] = *reinterpret_cast<const u32x4*>(cur + off);
      if (!F16) al[i] = *reinterpret_cast<const u32x4*>(cur + S2_TILE + off);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = g * S2_PLANE + (wn * 128 + j * 32 + l31) * 16;
      bh[j] = *reinterpret_cast<const u32x4*>(cur + 2 * S2_TILE + off);
      if (!F16) bl[j] = *reinterpret_cast<const u32x4*>(cur + 3 * S2_TILE + off);
    }
  };
  auto mfma_ops = [&]() {
    if (F16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bh[j]),
                                                             acc[i][j], 0, 0, 0);
    } else {
      // pass-major order: the three dependent updates of one accumulator are 8 MFMAs apart
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma_32x32x16<HM>(ah[i], bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma_32x32x16<HM>(al[i], bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma_32x32x16<HM>(ah[i], bl[j], acc[i][j]);
    }
  };
  {
    // One basic block per K-block (no conditionals: the tail re-loads the last block and re-stores it into the idle
    // buffer), so the scheduler can be told to thread the 8 LDS stores through the first 24 MFMAs and the 8 global
    // loads through the second 24 instead of issuing them in bursts that stall every wave at once.
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = lds + (kt & 1) * S2_BUF;
      read_ops(cur, 0);
      store_regs(lds + ((kt + 1) & 1) * S2_BUF);
      mfma_ops();
      __builtin_amdgcn_sched_group_barrier(0x100, F16 ? 6 : 12, 0);
#pragma unroll
      for (int i = 0; i < (F16 ? 4 : 8); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, F16 ? 2 : 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      read_ops(cur, 1);
      load_regs(min(kt + 2, nk - 1) * SB_K);
      mfma_ops();
      __builtin_amdgcn_sched_group_barrier(0x100, F16 ? 6 : 12, 0);
#pragma unroll
      for (int i = 0; i < (F16 ? 4 : 8); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, F16 ? 2 : 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 128 + j * 32 + l31;
    const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
        if (m < M && n < N) {
          float v = acc[i][j][r] + bv;
          if (act == MS_ACT_CLAMP) v = fminf(fmaxf(v, lo), hi);
          Y[(size_t)m * N + n] = v;
        }
      }
  }
}

// ---- 256 x 256 x 32, 8 waves, operands staged by LDS-DMA (`buffer_load_dwordx4 ... lds`), fragments double-buffered in
// registers.  What kernel2 leaves on the table (profiles/r02b: MFMA pipe 69 % busy): every half K-block its waves issue
// 12 fragment reads and then wait for them, both waves of a SIMD at the same moment, and 32 VGPRs per lane are tied up
// staging the next block's operands.  Here the DMA needs no staging registers, which pays for a second fragment set, so
// the reads of the NEXT half block are in flight under the MFMAs of the current one; one barrier per K-block at mid-block:
//
//   top : ds_read frags(b, half 1) -> set B      || 24 MFMA on set A = frags(b, half 0)
//         s_waitcnt lgkmcnt(0), vmcnt(0); s_barrier      (every wave's DMA of block b+1 has landed; nobody reads stage b&1 any more)
//   mid : DMA block b+2 -> stage b&1; ds_read frags(b+1, half 0) -> set A   || 24 MFMA on set B
//
// LDS image per stage and plane: 256 rows x 64 B, row-major -- what the DMA writes (wave-uniform base + lane * 16: a
// 1-KB piece = 16 rows, 4 lanes fetch one row's 64 contiguous bytes) -- with the 16-byte granule kg of row r stored at
// position kg ^ ((r >> 2) & 3): the source address is permuted, the destination stays linear, and the fragment read
// (row = lane & 31, kg = 2 half + (lane >> 5)) applies the same involution, which spreads every 16-lane group of a
// ds_read_b128 over all 64 banks.  Accumulation order per output element is kernel2's, so results are bit-identical.
constexpr int G4_PLANE = 256 * 64;
constexpr int G4_STAGE = 4 * G4_PLANE;
constexpr int G4_LDS = 2 * G4_STAGE;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// one LDS-DMA piece: 64 lanes x 16 B from (rsrc, voff + soff) to LDS [dst, dst + 1 KiB) in lane order.  (A __device__ helper:
// the generic -> LDS pointer cast does not exist in the host pass that emits the kernel's launch stub.)
__device__ __forceinline__ void lds_dma_16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}

// WN = wave columns: 2 = 256 x 256 tile, 8 waves, two per SIMD (the shipped form); 1 = 256 x 128 tile, 4 waves, one per
// SIMD: 224 VGPRs and 128 KB of LDS per workgroup, i.e. a workgroup that fits on a CU BESIDE a workgroup of the persistent
// LSTM kernel (272 VGPRs, 20 KB) -- the co-residency experiment of DESIGN 7.0 (tools/overlap_probe.py)
// NJ = 32-column W fragments per wave: 4 (64 x 128 per wave: the forms above) or, with WN = 2, 2 -- a 256 x 128 tile on EIGHT
// waves of 64 x 64 (round 4): twice the workgroups of the 256 x 256 form for outputs that would otherwise leave CUs idle
// (a streaming chunk's projection is 1 024 x 8 192: 128 tiles of 256 x 256 on 256 CUs), every output still the same
// k-ordered sum (bit-identical), 8 fragment reads per 12 MFMAs instead of 12 per 24.
// TM = tile rows: 256 (the forms above) or, with WN = 2 and NJ = 2, 128 -- a 128 x 128 tile on FOUR waves of 64 x 64 (round 6):
// outputs of ~500 rows (one utterance's projection, a 32-stream chunk's) get 120 .. 128 workgroups of 256 x 128, half of the
// chip; 128-row tiles give them 240 .. 256.  A wave's work per K-block is the 256 x 128 form's (same loop, same k-ordered sums:
// bit-identical); its DMA share is 4 x-pieces + 4 W-pieces (F16: 2 + 2).
template <int P, int WN, int PER_STEP_ = 3, bool SPACED = false, int NJ = 4, int TM = 256>
__device__ __forceinline__ void gemm4_body(const unsigned short* __restrict__ Ah, const unsigned short* __restrict__ Al,
                                           const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                           const float* __restrict__ bias, float* __restrict__ Y, int M, int K, int N, int act,
                                           float lo, float hi, const int* __restrict__ m_eff, int lda, int ldw, int kmode) {
  // kmode (GEMM_K_*): a contraction cut in two along K and run as two launches -- the first (K_FIRST) writes its accumulators as
  // they are (no bias, no activation), the second (K_SECOND) starts from them and finishes with the epilogue -- gives every
  // output element the k-ordered chain of the single launch, i.e. its bits: the accumulators travel through memory as exact
  // float32 copies (rnn.hip, ms_rnn_stack_forward: the two halves become available at different times).
  constexpr bool F16 = prec_one_plane(P), HM = P == PREC_F16X3;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int TN = 32 * NJ * WN;
  static_assert(NJ == 4 || (NJ == 2 && WN == 2), "the narrow form is the 8-wave kernel");
  static_assert(TM == 256 || (TM == 128 && NJ == 2 && WN == 2), "128-row tiles: the four-wave 128 x 128 form");
  // m_eff (optional): the number of rows that exist is a word in device memory (<= the M the grid was sized for): the
  // recurrent layers' packed projection, whose row count is the sum of the batch's lengths (rnn.hip).  Workgroups past
  // the tiles of that many rows leave at once.
  if (m_eff != nullptr) {
    M = __builtin_amdgcn_readfirstlane(*m_eff);
    if ((int)blockIdx.x >= ((N + TN - 1) / TN) * ((M + TM - 1) / TM)) return;
  }
  const int nbn = (N + TN - 1) / TN, nbm = (M + TM - 1) / TM;
  const int nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  constexpr int GM = 4;
  const int per_group = GM * nbn;
  const int first_m = (bid / per_group) * GM;
  const int gm = min(GM, nbm - first_m);
  const int bm = first_m + (bid % per_group) % gm;
  const int bn = (bid % per_group) / gm;
  const int m0 = bm * TM, n0 = bn * TN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- this wave's DMA pieces (16 rows each): NPA pieces of x plane PA from tile row RA, NPB pieces of W plane PB from RB.
  // WN = 2: 8 waves x 8 pieces of ONE plane (F16: 4); WN = 1: 4 waves x (8 of an x plane + 4 of a W plane) (F16: 4 + 2).
  // NJ = 2 (8 waves, 256 x 128): 32 x-pieces + 16 W-pieces per K-block = 4 + 2 per wave (F16: 2 + 1)
  constexpr int NPA = NJ == 2 ? (F16 ? 2 : 4) : WN == 2 ? (F16 ? 4 : 8) : (F16 ? 4 : 8);
  constexpr int NPB = TM == 128 ? NPA : NJ == 2 ? (F16 ? 1 : 2) : WN == 2 ? 0 : (F16 ? 2 : 4);
  constexpr int NP = NPA + NPB;
  int PA, RA, PB = 2, RB = 0;
  if (TM == 128) {
    // four waves, 128 x-rows and 128 W-rows per plane: a wave takes 4 (F16: 2) pieces of 16 rows from each
    if (F16) { PA = 0; RA = wave * 32; PB = 2; RB = wave * 32; }
    else { PA = wave >> 1; RA = (wave & 1) * 64; PB = 2 + (wave >> 1); RB = (wave & 1) * 64; }
  } else if (NJ == 2) {
    if (F16) { PA = 0; RA = wave * 32; PB = 2; RB = wave * 16; }
    else { PA = wave >> 2; RA = (wave & 3) * 64; PB = 2 + (wave >> 2); RB = (wave & 3) * 32; }
  } else if (WN == 2) {
    const int g0 = wave * NPA;                               // pieces numbered plane-major, 16 per plane
    PA = F16 ? (g0 >> 4) * 2 : (g0 >> 4);                    // 0 = x hi, 1 = x lo, 2 = W hi, 3 = W lo
    RA = (g0 & 15) * 16;
  } else if (F16) {
    PA = 0; RA = wave * 64; PB = 2; RB = wave * 32;
  } else {
    PA = wave >> 1; RA = (wave & 1) * 128; PB = 2 + (wave >> 1); RB = (wave & 1) * 64;
  }
  auto plane_rsrc = [&](int pl) {
    const unsigned short* src = pl == 0 ? Ah : pl == 1 ? Al : pl == 2 ? Wh : Wl;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, (unsigned)((size_t)(pl < 2 ? M : N) * (pl < 2 ? lda : ldw) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = plane_rsrc(PA), rsrc_b = plane_rsrc(PB);
  int voff[NP];                                              // byte offset of this lane's granule at k0 = 0
  {
    const int kgs = (lane & 3) ^ ((lane >> 4) & 3);          // granule fetched into position lane & 3 of row lane >> 2
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int pl = p < NPA ? PA : PB;
      const int rows = pl < 2 ? M : N, row = (pl < 2 ? m0 : n0) + (p < NPA ? RA + p * 16 : RB + (p - NPA) * 16) + (lane >> 2);
      voff[p] = min(row, rows - 1) * ((pl < 2 ? lda : ldw) * 2) + kgs * 16;
    }
  }
  // LDS stages.  bf16x3: two stages of four planes (x hi, x lo, W hi, W lo).  fp16 on the 8-wave forms (round 4): the lo planes
  // are never written, so their space holds two MORE stages -- stage s lives at (s & 1) * G4_STAGE + (s >> 1) * G4_PLANE -- and
  // a K-block's DMA is issued FOUR blocks ahead instead of two: an fp16 block is 8 MFMAs per wave (~260 cycles), less than
  // an L2 round trip, so with two stages every block waited for its successor's operands (0.73 PF/s at the chunk's projection).
  constexpr int RING = (F16 && WN == 2) ? 4 : 2;
  auto stage_off = [&](int stage) { return RING == 4 ? (stage & 1) * G4_STAGE + (stage >> 1) * G4_PLANE : stage * G4_STAGE; };
  auto dma_block = [&](int kb, int stage) {                  // all pieces of K-block kb
    const int soff = __builtin_amdgcn_readfirstlane(kb * (SB_K * 2));
    char* dst_a = lds + stage_off(stage) + PA * G4_PLANE + RA * 64;
    char* dst_b = lds + stage_off(stage) + PB * G4_PLANE + RB * 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p < NPA) lds_dma_16(rsrc_a, dst_a + p * 1024, voff[p], soff);
      else lds_dma_16(rsrc_b, dst_b + (p - NPA) * 1024, voff[p], soff);
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (kmode == GEMM_K_SECOND) {   // the first half's accumulators, in the epilogue's own element mapping (N % 4 == 0: launcher)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 64 + i * 32 + l31;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * (32 * NJ) + j * 32 + 8 * g + 4 * half;
          if (n + 3 < N) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Y + (size_t)m * N + n);
            acc[i][j][4 * g] = v[0]; acc[i][j][4 * g + 1] = v[1]; acc[i][j][4 * g + 2] = v[2]; acc[i][j][4 * g + 3] = v[3];
          }
        }
    }
  }

  // fragment addresses: row (lane & 31) of a 32-row group, granule kg = 2 h + half at position kg ^ ((row >> 2) & 3)
  const int sw = (l31 >> 2) & 3;
  const int foff0 = l31 * 64 + ((half ^ sw) * 16);           // h = 0: kg = half
  const int foff1 = l31 * 64 + (((2 + half) ^ sw) * 16);     // h = 1: kg = 2 + half
  const int a_base = wm * 64 * 64, b_base = 2 * G4_PLANE + wn * (32 * NJ) * 64;

  u32x4 fa[WN == 1 ? 1 : 2][6 + 6];   // [set][ah0 ah1 al0 al1 | bh0..3 bl0..3]: 12 granules per set
  auto read_frags = [&](int set, int stage, int h) {
    const char* st = lds + stage_off(stage) + (h ? foff1 : foff0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[set][i] = *reinterpret_cast<const u32x4*>(st + a_base + i * 32 * 64);
      if (!F16) fa[set][2 + i] = *reinterpret_cast<const u32x4*>(st + G4_PLANE + a_base + i * 32 * 64);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      fa[set][4 + j] = *reinterpret_cast<const u32x4*>(st + b_base + j * 32 * 64);
      if (!F16) fa[set][8 + j] = *reinterpret_cast<const u32x4*>(st + G4_PLANE + b_base + j * 32 * 64);
    }
  };
  // operand roles: the W fragment is the MFMA's A operand and the x fragment its B operand, so a lane's accumulator
  // registers r = 4 g .. 4 g + 3 are four CONSECUTIVE output columns n of one row m (one 16-byte store instead of four
  // 4-byte ones); every output element is the same k-ordered sum of the same products as in kernel2 (bit-identical)
  auto mfma_set = [&](int set) {
    if (F16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][4 + j]),
                                                             __builtin_bit_cast(f16x8, fa[set][i]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = mfma_32x32x16<HM>(fa[set][4 + j], fa[set][i], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = mfma_32x32x16<HM>(fa[set][4 + j], fa[set][2 + i], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = mfma_32x32x16<HM>(fa[set][8 + j], fa[set][i], acc[i][j]);
    }
  };

  const int nk = K / SB_K;
  dma_block(0, 0);
  dma_block(min(1, nk - 1), 1);
  if (RING == 4) {
    dma_block(min(2, nk - 1), 2);
    dma_block(min(3, nk - 1), 3);
  }
  // (waits are the builtin, not inline asm: the compiler's own wait insertion then knows the fragment sets are complete
  // and does not wait again at their first use, which would also cover the reads issued in between)
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  if (WN == 1) {
    // One wave per SIMD and at most 208 registers (the two-stream LSTM kernel holds 304 of the SIMD's 512): one x fragment
    // set per half block and a two-deep ring of W fragment pairs, each fetched under the six MFMAs of the pair before it;
    // the barrier sits in front of the block's last MFMA group (all reads of the stage are complete by then), so the next
    // block's first fragments are fetched under that group too.
    u32x4 xa[2][4], wb[2][2];      // [ring][x hi0 hi1 lo0 lo1], [ring][W hi, W lo]
    auto rd_x = [&](int ring, int stage, int h) {
      const char* st = lds + stage * G4_STAGE + (h ? foff1 : foff0) + a_base;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xa[ring][i] = *reinterpret_cast<const u32x4*>(st + i * 32 * 64);
        if (!F16) xa[ring][2 + i] = *reinterpret_cast<const u32x4*>(st + G4_PLANE + i * 32 * 64);
      }
    };
    auto rd_w = [&](int ring, int stage, int h, int j) {
      const char* st = lds + stage * G4_STAGE + (h ? foff1 : foff0) + b_base + j * 32 * 64;
      wb[ring][0] = *reinterpret_cast<const u32x4*>(st);
      if (!F16) wb[ring][1] = *reinterpret_cast<const u32x4*>(st + G4_PLANE);
    };
    // (issue(k): the k-th DMA piece of this step, placed in front of MFMA group k when the pieces are SPACED)
    auto mm = [&](int xr, int wr, int j, auto&& issue) {
      issue(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (F16) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wb[wr][0]), __builtin_bit_cast(f16x8, xa[xr][i]), acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = mfma_32x32x16<HM>(wb[wr][0], xa[xr][i], acc[i][j]);
        }
      }
      if (!F16) {
        issue(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = mfma_32x32x16<HM>(wb[wr][0], xa[xr][2 + i], acc[i][j]);
        issue(2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = mfma_32x32x16<HM>(wb[wr][1], xa[xr][i], acc[i][j]);
      }
    };
    rd_x(0, 0, 0);
    rd_w(0, 0, 0, 0);
    // (block 1 was issued in the prologue with block 0; from block 2 on a block's DMA pieces are issued a few at a time in
    // the first steps of the block BEFORE it -- a single wave cannot hide a burst of 12 DMA issues behind six MFMAs)
    constexpr int PER_STEP = PER_STEP_, DMA_STEPS = (NP + PER_STEP - 1) / PER_STEP;   // shipped: 3 pieces in each of 4 steps
    for (int b = 0; b < nk; ++b) {
      const int cur = b & 1;
      const int soff_next = __builtin_amdgcn_readfirstlane(min(b + 1, nk - 1) * (SB_K * 2));
      char* dst_a = lds + (cur ^ 1) * G4_STAGE + PA * G4_PLANE + RA * 64;
      char* dst_b = lds + (cur ^ 1) * G4_STAGE + PB * G4_PLANE + RB * 64;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int step = h * 4 + j;               // 0..7: W pair `step` lives in ring slot step & 1, x set h in ring slot h
          auto piece = [&](int p) {                  // block b+1 -> the other stage (free since the barrier of block b-1)
            if (p < NPA) lds_dma_16(rsrc_a, dst_a + p * 1024, voff[p], soff_next);
            else lds_dma_16(rsrc_b, dst_b + (p - NPA) * 1024, voff[p], soff_next);
          };
          if (!SPACED && b > 0 && step < DMA_STEPS) {
#pragma unroll
            for (int p = step * PER_STEP; p < (step + 1) * PER_STEP && p < NP; ++p) piece(p);
          }
          auto issue = [&](int k) {                  // SPACED: piece k of this step in front of MFMA group k
            if (SPACED && b > 0 && step < DMA_STEPS && k < PER_STEP && step * PER_STEP + k < NP) piece(step * PER_STEP + k);
          };
          if (step == 7) {
            // every read of this stage has been issued and (below) waited for; the next block has landed (this wave's pieces)
            __builtin_amdgcn_s_waitcnt(0x0070);     // lgkmcnt(0) + vmcnt(0)
            __builtin_amdgcn_s_barrier();
            rd_x(0, cur ^ 1, 0);
            rd_w(0, cur ^ 1, 0, 0);
          } else {
            if (j == 3) rd_x(1, cur, 1);            // h = 0, j = 3: the second half's x set
            rd_w((step + 1) & 1, cur, (step + 1) >> 2, (step + 1) & 3);
          }
          mm(h, step & 1, j, issue);
        }
    }
  } else {
  read_frags(0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  __builtin_amdgcn_sched_barrier(0);

  constexpr int NR = NJ == 2 ? (F16 ? 4 : 8) : (F16 ? 6 : 12);          // fragment reads per half block (NJ = 2: 12 MFMAs, 8 reads)
  constexpr int MPR = F16 ? 8 / 6 + 1 : 2;  // MFMAs issued per read in the interleave (F16: 8 MFMAs, 6 reads)
  for (int b = 0; b < nk; ++b) {
    const int cur = b & (RING - 1), nxt = (b + 1) & (RING - 1);
    // ---- top: frags(b, half 1) -> set 1 under the MFMAs on set 0
    read_frags(1, cur, 1);
    mfma_set(0);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // lgkmcnt(0): set 1 complete; this wave's pieces of block b+1 landed: vmcnt(0) with two stages, vmcnt(2 NP) with four
    // (blocks b+2 and b+3, issued later, may still be in flight; DMA pieces complete in issue order)
    if (RING == 4) __builtin_amdgcn_s_waitcnt(0x0070 | (2 * NP));
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- mid: block b+RING -> the stage just drained; frags(b+1, half 0) -> set 0 under the MFMAs on set 1
    // (the tail re-loads the last block into a stage nobody reads again and reads fragments nobody uses: no branches)
    dma_block(min(b + RING, nk - 1), cur);
    read_frags(0, nxt, 0);
    mfma_set(1);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i < NP) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): set 0 complete
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  __builtin_amdgcn_s_waitcnt(0x0070);     // the tail's DMA must have landed before the LDS is released

  // D[n][m]: lane = column m (lane & 31), register r -> row n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const bool n_vec = (N % 4) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * (32 * NJ) + j * 32 + 8 * g + 4 * half;
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (n_vec && n + 3 < N) {
          if (bias != nullptr && kmode != GEMM_K_FIRST) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
            v += bv;
          }
          if (act == MS_ACT_CLAMP && kmode != GEMM_K_FIRST) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], lo), hi);
          }
          *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < N) {
              float x = v[e] + (bias != nullptr ? bias[n + e] : 0.f);
              if (act == MS_ACT_CLAMP) x = fminf(fmaxf(x, lo), hi);
              Y[(size_t)m * N + n + e] = x;
            }
        }
      }
  }
}

template <int P>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16x3_kernel4(const unsigned short* __restrict__ Ah,
                                                                 const unsigned short* __restrict__ Al,
                                                                 const unsigned short* __restrict__ Wh,
                                                                 const unsigned short* __restrict__ Wl,
                                                                 const float* __restrict__ bias, float* __restrict__ Y,
                                                                 int M, int K, int N, int act, float lo, float hi,
                                                                 const int* __restrict__ m_eff, int lda, int ldw, int kmode) {
  gemm4_body<P, 2>(Ah, Al, Wh, Wl, bias, Y, M, K, N, act, lo, hi, m_eff, lda, ldw, kmode);
}

// 256 x 128 tiles on eight waves (NJ = 2): outputs whose 256 x 256 tiles would leave CUs without a workgroup
template <int P>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16x3_kernel4h(const unsigned short* __restrict__ Ah,
                                                                  const unsigned short* __restrict__ Al,
                                                                  const unsigned short* __restrict__ Wh,
                                                                  const unsigned short* __restrict__ Wl,
                                                                  const float* __restrict__ bias, float* __restrict__ Y,
                                                                  int M, int K, int N, int act, float lo, float hi,
                                                                  const int* __restrict__ m_eff, int lda, int ldw, int kmode) {
  gemm4_body<P, 2, 3, false, 2>(Ah, Al, Wh, Wl, bias, Y, M, K, N, act, lo, hi, m_eff, lda, ldw, kmode);
}

// 128 x 128 tiles on four waves (NJ = 2, TM = 128): outputs whose 256 x 128 tiles would fill at most half of the CUs
template <int P>
__global__ __launch_bounds__(256, 1) void gemm_nt_bf16x3_kernel4q(const unsigned short* __restrict__ Ah,
                                                                  const unsigned short* __restrict__ Al,
                                                                  const unsigned short* __restrict__ Wh,
                                                                  const unsigned short* __restrict__ Wl,
                                                                  const float* __restrict__ bias, float* __restrict__ Y,
                                                                  int M, int K, int N, int act, float lo, float hi,
                                                                  const int* __restrict__ m_eff, int lda, int ldw, int kmode) {
  gemm4_body<P, 2, 3, false, 2, 128>(Ah, Al, Wh, Wl, bias, Y, M, K, N, act, lo, hi, m_eff, lda, ldw, kmode);
}

// the 4-wave form, capped at 208 VGPRs (the two-stream LSTM kernel allocates 304 of a SIMD's 512): experiment only
template <int P, int PER_STEP = 3, bool SPACED = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x3_kernel4n(
    const unsigned short* __restrict__ Ah, const unsigned short* __restrict__ Al, const unsigned short* __restrict__ Wh,
    const unsigned short* __restrict__ Wl, const float* __restrict__ bias, float* __restrict__ Y, int M, int K, int N, int act,
    float lo, float hi, const int* __restrict__ m_eff, int lda, int ldw, int kmode) {
  gemm4_body<P, 1, PER_STEP, SPACED>(Ah, Al, Wh, Wl, bias, Y, M, K, N, act, lo, hi, m_eff, lda, ldw, kmode);
}

// tuning switch (tools/gemm_probe.py A/B runs in one process): 0 = default choice (kernel4, LDS-DMA), 2 = kernel2 (register staging), 7 = kernel4 with 256 x 128 tiles / 4 waves
static std::atomic<int> g_gemm_variant{0};

// hi/lo planes of an f32 matrix [rows, K] (K % 4 == 0, 16-byte aligned)
int split_planes_launch(const float* x, unsigned short* hi, unsigned short* lo, size_t elems, int prec, hipStream_t stream) {
  const size_t n4 = elems / 4;
  const int blocks = (int)std::min<size_t>((n4 + 255) / 256, 4096);
  if (prec == PREC_F16)
    hipLaunchKernelGGL(to_f16_plane_kernel, dim3(blocks), dim3(256), 0, stream, x, hi, n4);
  else if (prec == PREC_F16X3)
    hipLaunchKernelGGL(split_planes_kernel<true>, dim3(blocks), dim3(256), 0, stream, x, hi, lo, n4);
  else
    hipLaunchKernelGGL(split_planes_kernel<false>, dim3(blocks), dim3(256), 0, stream, x, hi, lo, n4);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

// the split mode of this process for callers that have no layer-specific choice to make (linear layers, convolutions)
int split_mode() {
  const int m = precision_mode();
  return m == PREC_F32 ? PREC_F16X3 : m;
}

// requires K % 32 == 0 and 16-byte aligned planes
// whether a GEMM of this size runs on the LDS-DMA kernels, which take the row count from device memory (gemm_bf16x3_launch_rows)
bool gemm_rows_from_device_ok(int M, int K, int N) {
  static const bool small_tile = getenv("MS_GEMM_TILE128") && getenv("MS_GEMM_TILE128")[0] == '1';
  static const bool regstage = getenv("MS_GEMM_REGSTAGE") && getenv("MS_GEMM_REGSTAGE")[0] == '1';
  const long tiles256 = (long)cdiv(M, S2_M) * cdiv(N, S2_N);
  const bool starved = tiles256 * 4 < (long)num_cus() * 3 && (long)cdiv(M, SB_M) * cdiv(N, SB_N) > tiles256;
  return !small_tile && !regstage && !starved && (long)M * N >= 4L * 1024 * 1024 && K % 32 == 0 &&
         (size_t)std::max(M, N) * K * 2 < ((size_t)1 << 31) && g_gemm_variant.load(std::memory_order_relaxed) != 2;
}

int gemm_bf16x3_launch_rows(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                            const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                            float hi, int prec, hipStream_t stream, const int* m_eff);
int gemm_bf16x3_launch_ld(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                          const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                          float hi, int prec, hipStream_t stream, const int* m_eff, int lda, int ldw, int kmode);

int gemm_bf16x3_launch(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                       const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                       float hi, int prec, hipStream_t stream) {
  return gemm_bf16x3_launch_rows(ah, al, wh, wl, bias, y, M, K, N, act, lo, hi, prec, stream, nullptr);
}

// one of a kernel template's three plane-format instances (`prec`: PREC_BF16X3 | PREC_F16 | PREC_F16X3)
#define MS_BY_PREC(K) (prec == PREC_F16 ? K<PREC_F16> : prec == PREC_F16X3 ? K<PREC_F16X3> : K<PREC_BF16X3>)

int gemm_bf16x3_launch_rows(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                            const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                            float hi, int prec, hipStream_t stream, const int* m_eff) {
  return gemm_bf16x3_launch_ld(ah, al, wh, wl, bias, y, M, K, N, act, lo, hi, prec, stream, m_eff, K, K, GEMM_K_WHOLE);
}

// m_eff: optional device word holding the number of rows that exist (<= M); only where gemm_rows_from_device_ok(M, K, N).
// lda / ldw: elements between consecutive rows of the x / W planes (>= K, multiples of 8; K itself for dense operands).
// The names say bf16x3 for history's sake: `prec` selects the plane format (bf16 hi + lo, fp16 hi + lo, one fp16 plane).
// kmode: GEMM_K_WHOLE, or one of the two launches of a contraction cut along K (gemm4_body) -- only on the LDS-DMA kernels:
// gemm_k_halves_ok(M, N) says whether this output runs on them.
bool gemm_k_halves_ok(int M, int N) {
  static const bool small_tile = getenv("MS_GEMM_TILE128") && getenv("MS_GEMM_TILE128")[0] == '1';
  static const bool regstage = getenv("MS_GEMM_REGSTAGE") && getenv("MS_GEMM_REGSTAGE")[0] == '1';
  static const bool half_off = getenv("MS_GEMM_HALF_TILE") && getenv("MS_GEMM_HALF_TILE")[0] == '0';
  if (small_tile || regstage || half_off || N % 4 != 0 || g_gemm_variant.load(std::memory_order_relaxed) != 0) return false;
  const long tiles256 = (long)cdiv(M, S2_M) * cdiv(N, S2_N);
  const bool starved = tiles256 * 4 < (long)num_cus() * 3 && (long)cdiv(M, SB_M) * cdiv(N, SB_N) > tiles256;
  const bool t64 = (long)cdiv(M, T6) * cdiv(N, T6) <= 2L * num_cus() && (long)cdiv(M, S2_M) * cdiv(N, 128) * 4 <= (long)num_cus();
  if (t64) return false;
  return starved ? (long)M * N >= 1024L * 1024 : (long)M * N >= 4L * 1024 * 1024;
}

int gemm_bf16x3_launch_ld(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                          const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                          float hi, int prec, hipStream_t stream, const int* m_eff, int lda, int ldw, int kmode) {
  if (kmode != GEMM_K_WHOLE && !gemm_k_halves_ok(M, N)) {
    set_error("gemm_bf16x3_launch_ld: a contraction cut along K needs an output that runs on the LDS-DMA kernels");
    return MS_ERR_UNSUPPORTED;
  }
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)gemm_nt_bf16x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_LDS));
    MS_HIP(hipFuncSetAttribute((const void*)gemm_nt_bf16x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_LDS));
#define MS_ATTR3(K, bytes)                                                                                         \
    MS_HIP(hipFuncSetAttribute((const void*)K<PREC_BF16X3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));   \
    MS_HIP(hipFuncSetAttribute((const void*)K<PREC_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));      \
    MS_HIP(hipFuncSetAttribute((const void*)K<PREC_F16X3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    MS_ATTR3(gemm_nt_bf16x3_kernel2, SPLIT2_LDS)
    MS_ATTR3(gemm_nt_bf16x3_kernel4, G4_LDS)
    MS_ATTR3(gemm_nt_bf16x3_kernel4n, G4_LDS)
    MS_ATTR3(gemm_nt_bf16x3_kernel4h, G4_LDS)
    MS_ATTR3(gemm_nt_bf16x3_kernel4q, G4_LDS)
#undef MS_ATTR3
    MS_HIP(hipFuncSetAttribute((const void*)gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
    MS_HIP(hipFuncSetAttribute((const void*)gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
    MS_HIP(hipFuncSetAttribute((const void*)gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
    MS_HIP(hipFuncSetAttribute((const void*)gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
    attr_once.done();
  }
  static const bool small_tile = getenv("MS_GEMM_TILE128") && getenv("MS_GEMM_TILE128")[0] == '1';
  const bool f16 = prec == PREC_F16;
  // 256 x 256 tiles unless they would leave a quarter of the CUs without a workgroup while 256 x 128 tiles would not
  // (narrow outputs, e.g. N = 512: 2 column tiles); both kernels accumulate in the same order, results are identical
  const long tiles256 = (long)cdiv(M, S2_M) * cdiv(N, S2_N);
  const bool starved = tiles256 * 4 < (long)num_cus() * 3 && (long)cdiv(M, SB_M) * cdiv(N, SB_N) > tiles256;
  // round 4: a starved output of at least a million elements takes 256 x 128 tiles on the LDS-DMA kernel (8 waves of 64 x 64)
  // instead of falling back to the round-1 register-staged kernel (or, in fp16 mode, half-filling the chip with 256 x 256
  // tiles): a streaming chunk's projection, 1 024 x 8 192.  MS_GEMM_HALF_TILE=0 keeps the old routing (A/B runs).
  static const bool half_off = getenv("MS_GEMM_HALF_TILE") && getenv("MS_GEMM_HALF_TILE")[0] == '0';
  // ... and one whose 256 x 128 tiles would still leave three quarters of the CUs idle takes 64 x 64 tiles (a chunk's hidden
  // FC layer at 64 streams: 1 024 x 1 024 = 32 tiles of 256 x 128, 256 of 64 x 64)
  auto tile64 = [&]() {
    const char* e = getenv("MS_GEMM_TILE64");
    return !(e && e[0] == '0') && !small_tile && m_eff == nullptr && g_gemm_variant.load(std::memory_order_relaxed) == 0 &&
           (long)cdiv(M, T6) * cdiv(N, T6) <= 2L * num_cus();
  };
  auto launch_tile64 = [&]() {
    hipLaunchKernelGGL(MS_BY_PREC(gemm_nt_bf16x3_tile64_kernel), dim3(cdiv(M, T6) * cdiv(N, T6)), dim3(256), 0, stream, ah, al, wh, wl,
                       bias, y, M, K, N, act, lo, hi, lda, ldw);
  };
  if (tile64() && (long)cdiv(M, S2_M) * cdiv(N, 128) * 4 <= (long)num_cus()) {
    launch_tile64();
    MS_LAUNCH_CHECK();
    return MS_OK;
  }
  if (!half_off && !small_tile && starved && m_eff == nullptr && (long)M * N >= 1024L * 1024 &&
      g_gemm_variant.load(std::memory_order_relaxed) == 0 && std::max((size_t)M * lda, (size_t)N * ldw) * 2 < ((size_t)1 << 31)) {
    // ... and 128 x 128 tiles on four waves where those 256 x 128 tiles would fill at most half of the CUs (round 6: a
    // 32-stream chunk's projection, 512 x 7 680: 120 -> 240 workgroups; one utterance's, 501 x 8 192: 128 -> 256).  The same
    // k-ordered sums (bit-identical).  MS_GEMM_QUARTER_TILE=0 (read per call) keeps the 256 x 128 form (A/B runs).
    const char* qe = getenv("MS_GEMM_QUARTER_TILE");
    if (!(qe && qe[0] == '0') && (long)cdiv(M, S2_M) * cdiv(N, 128) * 2 <= (long)num_cus() && cdiv(M, 128) > cdiv(M, S2_M)) {
      hipLaunchKernelGGL(MS_BY_PREC(gemm_nt_bf16x3_kernel4q), dim3(cdiv(M, 128) * cdiv(N, 128)), dim3(256), G4_LDS, stream, ah, al, wh, wl,
                         bias, y, M, K, N, act, lo, hi, m_eff, lda, ldw, kmode);
      MS_LAUNCH_CHECK();
      return MS_OK;
    }
    hipLaunchKernelGGL(MS_BY_PREC(gemm_nt_bf16x3_kernel4h), dim3(cdiv(M, S2_M) * cdiv(N, 128)), dim3(512), G4_LDS, stream, ah, al, wh, wl,
                       bias, y, M, K, N, act, lo, hi, m_eff, lda, ldw, kmode);
    MS_LAUNCH_CHECK();
    return MS_OK;
  }
  if (f16 || (!small_tile && !starved && (long)M * N >= 4L * 1024 * 1024)) {
    const int nwg2 = cdiv(M, S2_M) * cdiv(N, S2_N);
    const int variant = g_gemm_variant.load(std::memory_order_relaxed);
    // the LDS-DMA kernel (default) addresses its planes with 32-bit byte offsets; MS_GEMM_REGSTAGE=1 / variant 2 keep kernel2
    static const bool regstage = getenv("MS_GEMM_REGSTAGE") && getenv("MS_GEMM_REGSTAGE")[0] == '1';
    if (variant != 2 && !regstage && std::max((size_t)M * lda, (size_t)N * ldw) * 2 < ((size_t)1 << 31)) {
      if (variant >= 7 && variant <= 11) {   // 256 x 128 tiles, 4 waves: co-resident with the persistent LSTM (tools/overlap_probe.py)
        // how a wave issues the next block's 12 DMA pieces (bf16x3; experiments of tools/cotenant_variants.py): 7 = 3 per step
        // over four steps, 8 = 2 per step over six, 9 = 6 per step over two, 10 / 11 = as 7 / 8 with a step's pieces spaced
        // between its three MFMA groups
        auto k41 = MS_BY_PREC(gemm_nt_bf16x3_kernel4n);
        if (prec == PREC_BF16X3) {
          if (variant == 8) k41 = gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 2>;
          if (variant == 9) k41 = gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 6>;
          if (variant == 10) k41 = gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 3, true>;
          if (variant == 11) k41 = gemm_nt_bf16x3_kernel4n<PREC_BF16X3, 2, true>;
        }
        hipLaunchKernelGGL(k41, dim3(cdiv(M, S2_M) * cdiv(N, 128)), dim3(256), G4_LDS, stream, ah, al, wh, wl, bias, y, M, K, N,
                           act, lo, hi, m_eff, lda, ldw, kmode);
        MS_LAUNCH_CHECK();
        return MS_OK;
      }
      hipLaunchKernelGGL(MS_BY_PREC(gemm_nt_bf16x3_kernel4), dim3(nwg2), dim3(512), G4_LDS, stream, ah, al, wh, wl, bias, y, M, K, N, act,
                         lo, hi, m_eff, lda, ldw, kmode);
      MS_LAUNCH_CHECK();
      return MS_OK;
    }
    if (m_eff != nullptr) {
      set_error("gemm_bf16x3_launch_rows: this shape does not run on the kernels that take the row count from the device");
      return MS_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(MS_BY_PREC(gemm_nt_bf16x3_kernel2), dim3(nwg2), dim3(512), SPLIT2_LDS, stream, ah, al, wh, wl, bias, y, M, K, N,
                       act, lo, hi, lda, ldw);
    MS_LAUNCH_CHECK();
    return MS_OK;
  }
  if (m_eff != nullptr) {
    set_error("gemm_bf16x3_launch_rows: this shape does not run on the kernels that take the row count from the device");
    return MS_ERR_UNSUPPORTED;
  }
  // small outputs (round 4): 64 x 64 tiles when they give at most two workgroups per CU -- the same sums, bit for bit.
  // MS_GEMM_TILE64=0 (read per call: the tests' A/B switch) keeps the 256 x 128 register-staged kernel
  if (tile64()) {
    launch_tile64();
    MS_LAUNCH_CHECK();
    return MS_OK;
  }
  const int nwg = cdiv(M, SB_M) * cdiv(N, SB_N);
  if (prec == PREC_F16X3)
    hipLaunchKernelGGL(gemm_nt_bf16x3_kernel<true>, dim3(nwg), dim3(256), SPLIT_LDS, stream, ah, al, wh, wl, bias, y, M, K, N, act, lo, hi, lda, ldw);
  else
    hipLaunchKernelGGL(gemm_nt_bf16x3_kernel<false>, dim3(nwg), dim3(256), SPLIT_LDS, stream, ah, al, wh, wl, bias, y, M, K, N, act, lo, hi, lda, ldw);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
#undef MS_BY_PREC

}  // namespace ms

extern "C" int ms_gemm_set_variant(int v) {
  ms::g_gemm_variant.store(v, std::memory_order_relaxed);
  return MS_OK;
}

extern "C" size_t ms_linear_split_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  return ms::align_up((size_t)M * K * 4, 256) + ms::align_up((size_t)N * K * 4, 256);
}

// The weight planes of ms_linear_split_forward, made once (a streaming chunk's hidden FC layer re-split its
// 2.6 M weights on every call): [hi plane | lo plane] of N x K 16-bit words each, in the current precision mode's format.
extern "C" size_t ms_linear_split_packed_bytes(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  return ms::align_up((size_t)N * K * 4, 256);
}

extern "C" int ms_linear_split_pack(const float* w, void* packed, int K, int N, void* stream_) {
  MS_REQUIRE(w && packed, "null pointer");
  MS_REQUIRE(K > 0 && N > 0 && K % 32 == 0, "bad shape (K must be a multiple of 32)");
  MS_REQUIRE(((uintptr_t)w & 15) == 0, "w must be 16-byte aligned");
  unsigned short* wh = (unsigned short*)packed;
  const int prec = ms::split_mode();
  return ms::split_planes_launch(w, wh, wh + (size_t)N * K, (size_t)N * K, prec, (hipStream_t)stream_);
}

// ms_linear_split_forward with the weights as ms_linear_split_pack left them; the workspace holds the x planes only
// (M * K * 4 bytes).  Same kernels on the same plane values: the same bits.
extern "C" int ms_linear_split_forward_packed(const float* x, const void* packed_w, const float* bias, float* y, int M, int K, int N,
                                              int act, float act_lo, float act_hi, void* workspace, size_t workspace_bytes,
                                              void* stream_) {
  ms::ProfScope prof_span(MS_PROF_LINEAR, (hipStream_t)stream_);
  MS_REQUIRE(x && packed_w && y && workspace, "null pointer");
  MS_REQUIRE(M > 0 && K > 0 && N > 0, "bad shape");
  MS_REQUIRE(K % 32 == 0, "K must be a multiple of 32 (use ms_linear_forward otherwise)");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  MS_REQUIRE(((uintptr_t)x & 15) == 0, "x must be 16-byte aligned");
  if (workspace_bytes < ms::align_up((size_t)M * K * 4, 256)) {
    ms::set_error("ms_linear_split_forward_packed: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  unsigned short* xh = (unsigned short*)workspace;
  unsigned short* xl = xh + (size_t)M * K;
  const unsigned short* wh = (const unsigned short*)packed_w;
  const unsigned short* wl = wh + (size_t)N * K;
  const int prec = ms::split_mode();
  // (MS_TIMING_SKIP_SPLIT=1, read per call: timing experiments that stand in for a GEMM whose operand planes already exist --
  // tools/overlap_emulation.py; the planes are then whatever the workspace holds)
  const char* skip = getenv("MS_TIMING_SKIP_SPLIT");
  int rc = (skip && skip[0] == '1') ? MS_OK : ms::split_planes_launch(x, xh, xl, (size_t)M * K, prec, stream);
  if (rc == MS_OK) rc = ms::gemm_bf16x3_launch(xh, xl, wh, wl, bias, y, M, K, N, act, act_lo, act_hi, prec, stream);
  return rc;
}

extern "C" int ms_linear_split_forward(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                                       int act, float act_lo, float act_hi, void* workspace, size_t workspace_bytes,
                                       void* stream_) {
  ms::ProfScope prof_span(MS_PROF_LINEAR, (hipStream_t)stream_);
  MS_REQUIRE(x && w && y && workspace, "null pointer");
  MS_REQUIRE(M > 0 && K > 0 && N > 0, "bad shape");
  MS_REQUIRE(K % 32 == 0, "K must be a multiple of 32 (use ms_linear_forward otherwise)");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  MS_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "x and w must be 16-byte aligned");
  if (workspace_bytes < ms_linear_split_workspace_bytes(M, K, N)) {
    ms::set_error("ms_linear_split_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  unsigned short* xh = (unsigned short*)workspace;
  unsigned short* xl = xh + (size_t)M * K;
  unsigned short* wh = (unsigned short*)((char*)workspace + ms::align_up((size_t)M * K * 4, 256));
  unsigned short* wl = wh + (size_t)N * K;
  const int prec = ms::split_mode();
  int rc = ms::split_planes_launch(x, xh, xl, (size_t)M * K, prec, stream);
  if (rc == MS_OK) rc = ms::split_planes_launch(w, wh, wl, (size_t)N * K, prec, stream);
  if (rc == MS_OK) rc = ms::gemm_bf16x3_launch(xh, xl, wh, wl, bias, y, M, K, N, act, act_lo, act_hi, prec, stream);
  return rc;
}
