// y[M,N] = act(x[M,K] . w[N,K]^T + bias[N]) in exact float32 on the gfx950 matrix
// cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, no reduced precision).
//
// Replaces torch.nn.Linear at fully_connected.py:164 / deep_speech_1.py:124-136 and the
// x.W_ih^T input projection inside torch.nn.LSTM/GRU/RNN (rnn.py:177).
//
// Tiling (64-wide waves): 256 threads = 4 waves; block tile BM x BN with BK = 32.
// Both operands are K-contiguous, so a lane's 16-byte global load IS a [row][4 k]
// granule; LDS keeps granules as [k/4][row][4] so one ds_read_b128 feeds four
// consecutive MFMAs (lane half h = lane>>5 takes the odd/even k-quad).
#include "common.h"

namespace ms {

template <int WAVES_M, int WAVES_N, int TM, int TN>
struct GemmCfg {
  static constexpr int BM = WAVES_M * TM * 32;
  static constexpr int BN = WAVES_N * TN * 32;
  static constexpr int BK = 32;
  static constexpr int KQ = BK / 4;                 // k-quads per tile
  static constexpr int A_STRIDE = BM * 4 + 4;       // floats per k-quad plane (+16 B pad: ds_write conflicts)
  static constexpr int B_STRIDE = BN * 4 + 4;
  static constexpr int A_LD = (BM * KQ) / 256;      // float4 loads per thread
  static constexpr int B_LD = (BN * KQ) / 256;
  static constexpr int LDS_FLOATS = KQ * (A_STRIDE + B_STRIDE);
};

template <typename C, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ g, int rows_total, int K, int row0, int k0,
                                          f32x4* regs, int nld) {
  const int tid = threadIdx.x;
  const int kq = tid & 7;
  const int r_in = tid >> 3;  // 0..31
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < nld) {
      const int row = row0 + r_in + 32 * i;
      const int k = k0 + 4 * kq;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < rows_total) {
        const float* p = g + (size_t)row * K + k;
        if (VEC) {
          if (k < K) v = *reinterpret_cast<const f32x4*>(p);  // K % 4 == 0 => whole quad in range
        } else {
          if (k + 0 < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
          if (k + 3 < K) v.w = p[3];
        }
      }
      regs[i] = v;
    }
  }
}

__device__ __forceinline__ void store_tile(float* lds, int stride, const f32x4* regs, int nld) {
  const int tid = threadIdx.x;
  const int kq = tid & 7;
  const int r_in = tid >> 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < nld) *reinterpret_cast<f32x4*>(lds + kq * stride + (r_in + 32 * i) * 4) = regs[i];
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int M, int K, int N, int act, float lo, float hi) {
  using C = GemmCfg<WAVES_M, WAVES_N, TM, TN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + C::KQ * C::A_STRIDE;

  // XCD-aware tile order: blocks that share an XCD (bid % 8) walk neighbouring tiles,
  // so the x rows / weight rows they share stay in that XCD's L2.
  const int nbn = (N + C::BN - 1) / C::BN;
  const int nbm = (M + C::BM - 1) / C::BM;
  const int nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int bn = bid % nbn, bm = bid / nbn;
  const int m0 = bm * C::BM, n0 = bn * C::BN;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wr = wave / WAVES_N, wc = wave % WAVES_N;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[8], rb[8];
  // split-K: slice z = blockIdx.y of gridDim.y owns a contiguous range of K tiles and writes its partial sums to
  // Y + z*M*N (the consumer adds the slices in a fixed order); the bias goes into slice 0
  const int nk_all = (K + C::BK - 1) / C::BK;
  const int kt0 = (int)((long)blockIdx.y * nk_all / gridDim.y);
  const int nk = (int)((long)(blockIdx.y + 1) * nk_all / gridDim.y);
  Y += (size_t)blockIdx.y * M * N;
  if (blockIdx.y != 0) bias = nullptr;
  load_tile<C, VEC>(A, M, K, m0, kt0 * C::BK, ra, C::A_LD);
  load_tile<C, VEC>(W, N, K, n0, kt0 * C::BK, rb, C::B_LD);
  store_tile(As, C::A_STRIDE, ra, C::A_LD);
  store_tile(Bs, C::B_STRIDE, rb, C::B_LD);
  __syncthreads();

  for (int kt = kt0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      load_tile<C, VEC>(A, M, K, m0, (kt + 1) * C::BK, ra, C::A_LD);
      load_tile<C, VEC>(W, N, K, n0, (kt + 1) * C::BK, rb, C::B_LD);
    }
#pragma unroll
    for (int q2 = 0; q2 < C::KQ / 2; ++q2) {
      const int kq = 2 * q2 + half;
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(As + kq * C::A_STRIDE + ((wr * TM + i) * 32 + l31) * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Bs + kq * C::B_STRIDE + ((wc * TN + j) * 32 + l31) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile(As, C::A_STRIDE, ra, C::A_LD);
      store_tile(Bs, C::B_STRIDE, rb, C::B_LD);
      __syncthreads();
    }
  }

  // epilogue: lane holds column n = l31 of 16 rows
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wc * TN + j) * 32 + l31;
    const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wr * TM + i) * 32 + mfma32_row(r, lane);
        if (m < M && n < N) {
          float v = acc[i][j][r] + bv;
          if (act == MS_ACT_CLAMP) v = fminf(fmaxf(v, lo), hi);
          Y[(size_t)m * N + n] = v;
        }
      }
    }
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static int launch_cfg(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                      float lo, float hi, hipStream_t stream, int ksplit = 1) {
  using C = GemmCfg<WAVES_M, WAVES_N, TM, TN>;
  const int nwg = cdiv(M, C::BM) * cdiv(N, C::BN);
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  const bool vec = (K % 4 == 0) && (((uintptr_t)x | (uintptr_t)w) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((gemm_nt_f32_kernel<WAVES_M, WAVES_N, TM, TN, true>), dim3(nwg, ksplit), dim3(256), lds, stream, x, w,
                       bias, y, M, K, N, act, lo, hi);
  else
    hipLaunchKernelGGL((gemm_nt_f32_kernel<WAVES_M, WAVES_N, TM, TN, false>), dim3(nwg, ksplit), dim3(256), lds, stream, x, w,
                       bias, y, M, K, N, act, lo, hi);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

int linear_launch(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, float lo,
                  float hi, hipStream_t stream) {
  if (N <= 32) return launch_cfg<4, 1, 1, 1>(x, w, bias, y, M, K, N, act, lo, hi, stream);   // 128 x 32
  if (N <= 64) return launch_cfg<2, 2, 2, 1>(x, w, bias, y, M, K, N, act, lo, hi, stream);   // 128 x 64
  // few rows (decode steps, single clips): 128 x 128 tiles would occupy N/128 CUs only; narrower tiles spread the
  // weight stream over 2x / 4x as many workgroups
  const long wg128 = (long)cdiv(M, 128) * cdiv(N, 128);
  if (wg128 * 4 <= num_cus()) return launch_cfg<4, 1, 1, 1>(x, w, bias, y, M, K, N, act, lo, hi, stream);   // 128 x 32
  if (wg128 * 2 <= num_cus()) return launch_cfg<2, 2, 2, 1>(x, w, bias, y, M, K, N, act, lo, hi, stream);   // 128 x 64
  return launch_cfg<2, 2, 2, 2>(x, w, bias, y, M, K, N, act, lo, hi, stream);                // 128 x 128
}

// Few-row GEMM for decode steps (M <= 128): 128 x 32 tiles and `ksplit` K slices, so N/32 * ksplit workgroups stream
// the weights; y holds ksplit partial results [ksplit][M][N] that the caller adds in slice order (deterministic).
int linear_splitk_launch(const float* x, const float* w, const float* bias, float* y_parts, int M, int K, int N, int ksplit,
                         hipStream_t stream) {
  return launch_cfg<4, 1, 1, 1>(x, w, bias, y_parts, M, K, N, MS_ACT_NONE, 0.f, 0.f, stream, ksplit);
}

// y = act(sum over the K slices, in slice order, + nothing else: the bias went into slice 0)
__global__ void splitk_reduce_kernel(const float* __restrict__ parts, float* __restrict__ y, size_t mn, int ksplit, int act, float lo,
                                     float hi) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (size_t)gridDim.x * blockDim.x) {
    float v = parts[i];
    for (int z = 1; z < ksplit; ++z) v += parts[(size_t)z * mn + i];
    if (act == MS_ACT_CLAMP) v = fminf(fmaxf(v, lo), hi);
    y[i] = v;
  }
}

// K slices for a GEMM with few output COLUMNS (an output layer: 29 symbols = one 32-column tile per 128 rows, so a chunk's
// 1 024 rows are eight workgroups each walking all of K).  The slice count depends on (K, N) and the caller's flags only,
// NEVER on the number of rows: an utterance's logits must not depend on what it is batched with (utterance shards reproduce
// the whole batch bit for bit, streaming output does not depend on the chunking).  0 = the plain kernel.
// MS_LINEAR_FEW_ROWS (this used to be inferred from M <= 256, which made a row's rounding depend on its batch):
// the CALLER states that the layer serves a handful of rows (a single clip's hidden layers -- DeepSpeech1 on one 4 s clip:
// 201 x 1 024 x 1 024, 64 workgroups walking all of K) and accepts K-slice rounding for wide layers too.
int linear_splitk_slices(int K, int N, int flags) {
  if (K < 512) return 0;
  if (N > 64) return ((flags & MS_LINEAR_FEW_ROWS) && N <= 4096) ? std::min(8, K / 128) : 0;
  return std::min(8, K / 128);
}

}  // namespace ms

extern "C" size_t ms_linear_splitk_workspace_bytes(int M, int K, int N, int flags) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const int ks = ms::linear_splitk_slices(K, N, flags);
  return ks ? ms::align_up((size_t)ks * M * N * sizeof(float), 256) : 0;
}

extern "C" int ms_linear_splitk_forward(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                                        float act_lo, float act_hi, int flags, void* workspace, size_t workspace_bytes, void* stream_) {
  ms::ProfScope prof_span(MS_PROF_LINEAR, (hipStream_t)stream_);
  MS_REQUIRE(x && w && y, "null pointer");
  MS_REQUIRE(M > 0 && K > 0 && N > 0, "bad shape");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  MS_REQUIRE((flags & ~MS_LINEAR_FEW_ROWS) == 0, "unknown flag");
  hipStream_t stream = (hipStream_t)stream_;
  const int ks = ms::linear_splitk_slices(K, N, flags);
  // (no workspace = the caller asks for the plain kernel: MS_LINEAR_SPLITK=0 in the Python layer, A/B runs)
  if (ks == 0 || workspace == nullptr) return ms::linear_launch(x, w, bias, y, M, K, N, act, act_lo, act_hi, stream);
  MS_REQUIRE(workspace_bytes >= ms_linear_splitk_workspace_bytes(M, K, N, flags), "workspace too small");
  int rc = ms::linear_splitk_launch(x, w, bias, (float*)workspace, M, K, N, ks, stream);
  if (rc != MS_OK) return rc;
  const size_t mn = (size_t)M * N;
  hipLaunchKernelGGL(ms::splitk_reduce_kernel, dim3((unsigned)std::min<size_t>((mn + 255) / 256, 2048)), dim3(256), 0, stream,
                     (const float*)workspace, y, mn, ks, act, act_lo, act_hi);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_linear_forward(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                                 int act, float act_lo, float act_hi, void* stream) {
  ms::ProfScope prof_span(MS_PROF_LINEAR, (hipStream_t)stream);
  MS_REQUIRE(x && w && y, "null pointer");
  MS_REQUIRE(M > 0 && K > 0 && N > 0, "bad shape");
  MS_REQUIRE(act == MS_ACT_NONE || act == MS_ACT_CLAMP, "bad act");
  return ms::linear_launch(x, w, bias, y, M, K, N, act, act_lo, act_hi, (hipStream_t)stream);
}
