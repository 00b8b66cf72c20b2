// Device-resident RNN-T decoding (greedy and time-synchronous beam search).  NOT reference-derived: the reference
// snapshot has no transducer; the specification is this repository's own
// (myrtlespeech_amd/model/rnnt.py, oracle/rnnt_oracle.py).
//
// The whole decode of a batch is ONE host call that enqueues a fixed launch sequence per frame and never reads anything
// back: hypothesis lists, the prefix trie, the predictor-state pool and the emission lists all live in the caller's
// workspace.  Per frame and emission round:
//   joint_slots_kernel   log P(. | frame t, hypothesis) for every live hypothesis row (one workgroup per row); a row whose
//                        predictor output is fresh adds the pred_proj GEMM's K-slice partial sums itself and commits them
//   beam_round_kernel    per utterance: blank transitions merge into the next frame's set B (same prefix = same trie node,
//                        float32 scores joined by logaddexp), the beam_width best label extensions become the live set
//   greedy_round_kernel  per utterance: argmax; blank ends the frame, a label is emitted
//   predictor step       gather [embedding(label) | h_src] rows -> exact-f32 MFMA GEMM against [W_ih | W_hh] ->
//                        lstm_cell_kernel (writes the new state, builds the next layer's rows) -> ... -> pred_proj GEMM
//   beam_frame_end_kernel  B sorted (stable, descending) -> beam_width survivors; their states move to the frame-start region
// Rows of utterances that have ended (t >= len) or of empty beam slots are skipped inside the kernels, so the beam's launch
// sequence does not depend on the data.
//
// Greedy decode (round 3) does not pay for blanks any more.  The predictor's output only changes when a label is emitted, so
// between two emissions of an utterance every frame's joint is evaluated against the SAME predictor output: one iteration =
//   joint_slots_kernel<CHUNK>  log P(. | frame cur_t[i] + c, pred_i) for GREEDY_CHUNK consecutive frames of every utterance at once
//   greedy_scan_kernel         per utterance: walk those frames in order, blanks advance the frame, the FIRST label is emitted
//                              (at most max_symbols per frame, then the frame advances), requests ONE predictor step and stops
//   predictor step             only for the utterances that emitted
// i.e. about (labels of the longest transcript + blank runs longer than a chunk) iterations instead of T x max_symbols rounds,
// with the same arithmetic per (frame, prediction) pair and therefore the same transcripts.  The number of iterations depends
// on the data: the host enqueues iterations ahead and every few of them fetches a device counter of finished utterances
// (4 bytes, asynchronously, looked at two checks later so that the queue never drains).
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace ms {
int linear_launch(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, float lo,
                  float hi, hipStream_t stream);
int linear_splitk_launch(const float* x, const float* w, const float* bias, float* y_parts, int M, int K, int N, int ksplit,
                         hipStream_t stream);
}

constexpr int KSPLIT = 8;  // K slices of the gate GEMMs (partials added in slice order by lstm_cell_kernel)
#ifndef MS_RNNT_PF2
#define MS_RNNT_PF2 4      // k-steps of operands in flight in the two-plane beam GEMMs (8 measured: 163 VGPRs, one workgroup
                          // per CU instead of two, cell GEMM 12.9 -> 14.7 us)
#endif
constexpr int GREEDY_CHUNK = 32;  // greedy decode: frames evaluated per utterance and iteration

namespace {

struct DecLayout {
  // hypothesis lists
  size_t A_cnt, A_node, A_score, A_slot;
  size_t B_cnt, B_node, B_score, B_slot;
  size_t ext_label, ext_src, ext_dst;
  size_t live, out_cnt;
  // trie
  size_t node_cnt, node_parent, node_label, child;
  // predictor state pool + scratch
  size_t st_h, st_c, pp, pp_tmp, xrow, xrow2, gates, htop, logp, wcat, bcat;
  size_t total;
  int R, slots, maxn, bcap;
  size_t wcat_off[8], bcat_off[8];
  size_t v2_E, v2_Etmp, v2_G, v2_hpl, v2_whh, v2_wih, v2_wpred, v2_bias, v2_plog, v2_A2;
};

DecLayout dec_layout(int T, int N, int V, int D, int H, int L, int J, int w, int max_symbols, int greedy) {
  DecLayout W{};
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += ms::align_up(bytes, 256); return at; };
  const int R = N * w;
  const int regions = greedy ? 1 : (2 + (max_symbols > 1 ? max_symbols - 1 : 0));
  W.R = R;
  W.slots = regions * R;
  W.bcap = w * max_symbols;
  W.maxn = greedy ? 1 : 1 + T * (max_symbols > 1 ? max_symbols - 1 : 0) * w;
  const int V1 = V + 1;
  W.A_cnt = take((size_t)N * 4);
  W.A_node = take((size_t)R * 4);
  W.A_score = take((size_t)R * 4);
  W.A_slot = take((size_t)R * 4);
  W.B_cnt = take((size_t)N * 4);
  W.B_node = take((size_t)N * W.bcap * 4);
  W.B_score = take((size_t)N * W.bcap * 4);
  W.B_slot = take((size_t)N * W.bcap * 4);
  W.ext_label = take((size_t)R * 4);
  W.ext_src = take((size_t)R * 4);
  W.ext_dst = take((size_t)R * 4);
  W.live = take((size_t)N * 4);
  W.out_cnt = take((size_t)N * 4);
  W.node_cnt = take((size_t)N * 4);
  W.node_parent = take((size_t)N * W.maxn * 4);
  W.node_label = take((size_t)N * W.maxn * 4);
  W.child = take(greedy ? 4 : (size_t)N * W.maxn * V * 4);
  W.st_h = take((size_t)W.slots * L * H * 4);
  W.st_c = take((size_t)W.slots * L * H * 4);
  W.pp = take((size_t)W.slots * J * 4);
  W.pp_tmp = take((size_t)KSPLIT * R * J * 4);
  const int in_max = (D > H ? D : H) + H;
  W.xrow = take((size_t)((R + 31) / 32 * 32) * in_max * 6);   // float32 rows, or three bf16 planes of whole 32-row groups (6 B)
  W.xrow2 = take((size_t)((R + 31) / 32 * 32) * in_max * 6);  // the fused layer kernel reads one x buffer and writes the other
  W.gates = take((size_t)KSPLIT * R * 4 * H * 4);
  W.htop = take((size_t)R * H * 4);
  W.logp = take((size_t)R * (greedy ? GREEDY_CHUNK : 1) * V1 * 4);
  size_t wc = 0, bc = 0;
  for (int l = 0; l < L; ++l) {
    W.wcat_off[l] = wc;
    W.bcat_off[l] = bc;
    wc += (size_t)4 * H * ((l == 0 ? D : H) + H);
    bc += (size_t)4 * H;
  }
  W.wcat = take(wc * 6);      // float32 rows [4H][K] (4 B per weight) or three bf16 planes of them (6 B): sized for the planes
  W.bcat = take(bc * 4);
  // ---- the round-5 beam path (beam2_*): see the comment above pred_gemm3_kernel
  if (!greedy) {
    const size_t rpad = (size_t)((R + 63) / 64 * 64);
    W.v2_E = take((size_t)V1 * 4 * H * 4);
    W.v2_Etmp = take((size_t)V1 * 4 * H * 4);
    W.v2_G = take((size_t)W.slots * L * 4 * H * 4);
    W.v2_hpl = take((size_t)L * 3 * rpad * H * 2);
    W.v2_whh = take((size_t)L * 3 * 4 * H * H * 2);
    W.v2_wih = take((size_t)(L > 1 ? L - 1 : 1) * 3 * 4 * H * H * 2);
    W.v2_wpred = take((size_t)3 * J * H * 2);
    W.v2_bias = take((size_t)L * 4 * H * 4);
    W.v2_plog = take((size_t)((J + 31) / 32) * R * V1 * 4);
    W.v2_A2 = take((size_t)(N + 3 * R) * 4);       // the next frame's live set (A_cnt, A_node, A_score, A_slot)
  }
  W.total = o;
  return W;
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// float32 logaddexp of the specification: evaluated in float64, rounded once.
__device__ __forceinline__ float logaddexp32(float a, float b) {
  const double x = (double)a, y = (double)b;
  if (x == y) return (float)(x + 0.6931471805599453);
  const double m = x > y ? x : y, d = x > y ? y - x : x - y;
  return (float)(m + log1p(exp(d)));
}

// [W_ih | W_hh] rows and b_ih + b_hh of one layer (once per decode call).
__global__ void pack_cat_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                const float* __restrict__ b_hh, float* __restrict__ wcat, float* __restrict__ bcat, int H,
                                int In) {
  const int row = blockIdx.x;  // 4H rows
  const int K = In + H;
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    wcat[(size_t)row * K + k] = k < In ? w_ih[(size_t)row * In + k] : w_hh[(size_t)row * H + (k - In)];
  if (threadIdx.x == 0) bcat[row] = (b_ih ? b_ih[row] : 0.f) + (b_hh ? b_hh[row] : 0.f);
}


// ---- the predictor's gate GEMMs as an error-free bf16 split (round 4).  gates[R <= 128.., 4H] = x[R, K] . Wcat[4H, K]^T with
// R = utterances x beam width hypothesis rows (128 at configs[3]) ran as exact-f32 MFMA in 128 x 32 tiles and eight K slices:
// 17.6 us per launch, four launches per frame, 45 % of the decode (profiles/r04e_cfg4_kernel_stats.csv) -- float32 MFMA issues
// 1/16 of the bf16 rate.  Here every float32 operand is written as THREE bf16 values h + m + l (8 + 8 + 8 significand bits: an
// exact decomposition) and the product keeps the six terms down to 2^-24 relative (hh, hm, mh, mm, hl, lh), accumulated in float32:
// float32-grade products at 6/16 of the float32 MFMA time.  The weights are split once per decode call (pack_cat3_kernel), the
// rows by the kernels that produce them (pred_gather_kernel, lstm_cell_kernel: store_split3), both into FRAGMENT-MAJOR planes
// (frag_off): a wave's operand load is one contiguous 1 KB block.  No LDS, no barrier: a wave owns 32 rows x 32 gate columns x
// one K slice, four k-steps of operands in flight; partial sums per K slice as before (lstm_cell_kernel adds them in slice
// order).  Measured (configs[3], same box): a gate GEMM 22.5 -> 17.8 us (the exact-f32 kernel's 51 426 launches average 17.6 us
// over gate AND the smaller pred_proj GEMMs), beam-8 decode 122 -> 112 ms, greedy 70 -> 60 ms.  Not the ~6 us its MFMA time
// would allow: 6 bytes per weight now come from beyond L2 per launch (50 MB at K = 2048) -- two W planes instead of three or a
// deeper prefetch moved it by 4 % / 0 % (EXPERIMENTS.md).  MS_RNNT_GATES_F32=1 keeps the exact-f32 MFMA path.
// Fragment-major operand layouts of the split gate GEMM: element (row, k) of a [rows, K] matrix sits where the lane that
// needs it (row % 32, (k % 16) / 8) finds its 8 consecutive k next to the other 63 lanes' -- a wave's operand load is then
// one contiguous block (1 KB of bf16, 2 KB of float32) instead of 32 row segments of 32 / 64 bytes (row-major operands made
// the kernel address-coalescing-bound: 32 us per launch).
__host__ __device__ __forceinline__ size_t frag_off(int row, int k, int K) {
  return ((((size_t)(row >> 5) * (K >> 4) + (k >> 4)) * 2 + ((k >> 3) & 1)) * 32 + (row & 31)) * 8 + (k & 7);
}
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
// x = h + m + l (three bf16 values, exact) written to the three planes of a fragment-major operand
__device__ __forceinline__ void store_split3(unsigned short* planes, size_t plane_stride, size_t off, float x) {
  const unsigned h = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
  const float r1 = x - __uint_as_float(h << 16);
  const unsigned m = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r1);
  const unsigned l = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(r1 - __uint_as_float(m << 16)));
  planes[off] = (unsigned short)h;
  planes[plane_stride + off] = (unsigned short)m;
  planes[2 * plane_stride + off] = (unsigned short)l;
}
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));
typedef float f32x16r __attribute__((ext_vector_type(16)));
// Round 6 (the re-cut beam sequence only): `two` != 0 = TWO fp16 planes hi + lo (22 mantissa bits; the encoder's f16x3 split)
// in planes 0 and 1 instead of the exact three-bf16 decomposition -- a launch of these GEMMs is a per-CU operand pull (590 KB
// at ~50 GB/s: EXPERIMENTS.md round 5), two planes are two thirds of it and half of the MFMAs.  MS_RNNT_PLANES=3 keeps the
// exact form.
__device__ __forceinline__ void store_split(unsigned short* planes, size_t plane_stride, size_t off, float x, int two) {
  if (!two) { store_split3(planes, plane_stride, off, x); return; }
  unsigned hi, lo;
  ms::plane_split<true>(x, hi, lo);
  planes[off] = (unsigned short)hi;
  planes[plane_stride + off] = (unsigned short)lo;
}

__device__ __forceinline__ unsigned bf16_rne(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

// unit_major (the fused layer kernel): gate row g H + u is stored as row (u / 8) 32 + g 8 + u % 8, so that the 32 rows of a
// 32-row fragment group are the four gates of eight hidden units
__global__ void pack_cat3_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                 const float* __restrict__ b_hh, unsigned short* __restrict__ planes, float* __restrict__ bcat,
                                 int H, int In, int unit_major) {
  const int src_row = blockIdx.x;  // 4H rows
  const int K = In + H;
  const size_t plane = (size_t)4 * H * K;
  const int gate_ = src_row / H, unit_ = src_row % H;
  const int row = unit_major ? (unit_ >> 3) * 32 + gate_ * 8 + (unit_ & 7) : src_row;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float x = k < In ? w_ih[(size_t)src_row * In + k] : w_hh[(size_t)src_row * H + (k - In)];
    const unsigned h = bf16_rne(x);
    const float r1 = x - __uint_as_float(h << 16);
    const unsigned m = bf16_rne(r1);
    const unsigned l = bf16_rne(r1 - __uint_as_float(m << 16));
    const size_t o = frag_off(row, k, K);
    planes[o] = (unsigned short)h;
    planes[plane + o] = (unsigned short)m;
    planes[2 * plane + o] = (unsigned short)l;
  }
  if (threadIdx.x == 0) bcat[row] = (b_ih ? b_ih[src_row] : 0.f) + (b_hh ? b_hh[src_row] : 0.f);
}

// grid (N4 / 64, nsplit, cdiv(R, 128)), 512 threads; parts [nsplit][R][N4]; K % (16 * nsplit) == 0, N4 % 64 == 0.
// Eight waves: wave = (row group of 32, column tile of 32); with eight K slices that is four waves per SIMD, which is what
// hides the L2 latency of a k-step's operands (a first form with four waves of 32 x 64 and one k-step of prefetch ran at the
// latency: 31.8 us per launch against 17.6 for the exact-f32 kernel).
__global__ __launch_bounds__(512) void pred_gates_split_kernel(const unsigned short* __restrict__ x, size_t xplane,
                                                               const unsigned short* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ parts, int R,
                                                               int K, int N4, int nsplit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int rg = wave & 3, nt = wave >> 2;
  const int n0 = blockIdx.x * 64 + nt * 32, z = blockIdx.y;
  const int m = blockIdx.z * 128 + rg * 32 + l31;             // this lane's x row
  const int kslice = K / nsplit, k0 = z * kslice;
  const size_t wplane = (size_t)N4 * K;
  const unsigned short* xp = x + frag_off(m, k0 + half * 8, K);          // x and W: three bf16 planes each, fragment-major
  const unsigned short* w0 = wp + frag_off(n0 + l31, k0 + half * 8, K);
  const bool row_ok = m < R;

  f32x16r acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  constexpr int PF = 4;                                        // k-steps of operands in flight (the weights come from beyond L2)
  u32x4r xf[PF][3], wf[PF][3];
  auto load = [&](int slot, int ks) {                           // one k-step further: 2 halves x 32 rows x 8 elements
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      xf[slot][pl] = u32x4r{0u, 0u, 0u, 0u};
      if (row_ok) xf[slot][pl] = *reinterpret_cast<const u32x4r*>(xp + (size_t)pl * xplane + (size_t)ks * 512);
      wf[slot][pl] = *reinterpret_cast<const u32x4r*>(w0 + (size_t)pl * wplane + (size_t)ks * 512);
    }
  };
  auto step = [&](int slot) {
    const bf16x8r bh = __builtin_bit_cast(bf16x8r, wf[slot][0]), bm = __builtin_bit_cast(bf16x8r, wf[slot][1]);
    const bf16x8r bl = __builtin_bit_cast(bf16x8r, wf[slot][2]);
    const bf16x8r xh = __builtin_bit_cast(bf16x8r, xf[slot][0]), xm = __builtin_bit_cast(bf16x8r, xf[slot][1]);
    const bf16x8r xl = __builtin_bit_cast(bf16x8r, xf[slot][2]);
    // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc, 0, 0, 0);
  };
  const int nks = kslice / 16;
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < nks) load(j, j);
  for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (ks + j < nks) {
        step(j);
        if (ks + j + PF < nks) load(j, ks + j + PF);
      }
  }
  // D[row = (r & 3) + 8 (r >> 2) + 4 half of the wave's 32][col = l31]
  float* out = parts + (size_t)z * R * N4;
  const int n = n0 + l31;
  const float bv = (z == 0 && bias != nullptr) ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = blockIdx.z * 128 + rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (mr < R) out[(size_t)mr * N4 + n] = acc[r] + bv;
  }
}

// x0[r] = [embedding[label_r] | h[src_r][layer 0]]; rows without a request are zero.
__global__ void pred_gather_kernel(const float* __restrict__ embedding, const int32_t* __restrict__ ext_label,
                                   const int32_t* __restrict__ ext_src, const int32_t* __restrict__ ext_dst,
                                   const float* __restrict__ st_h, float* __restrict__ xrow, int D, int H, int L, int V1,
                                   int frag, size_t xplane) {
  const int r = blockIdx.x;
  const bool valid = ext_dst[r] >= 0;
  const int src = ext_src[r];
  const int lab = min(max(ext_label[r], 0), V1 - 1);
  const int K = D + H;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float v = 0.f;
    if (valid) v = k < D ? embedding[(size_t)lab * D + k] : (src >= 0 ? st_h[((size_t)src * L + 0) * H + (k - D)] : 0.f);
    if (frag) store_split3(reinterpret_cast<unsigned short*>(xrow), xplane, frag_off(r, k, K), v);
    else xrow[(size_t)r * K + k] = v;
  }
}

// gates [R, 4H] (i, f, g, o) of layer l -> new (h, c) of the destination slot; also the next layer's input row
// [h' | h[src][l+1]] or, after the last layer, the row of `htop`.
__global__ void lstm_cell_kernel(const float* __restrict__ gates, const int32_t* __restrict__ ext_src,
                                 const int32_t* __restrict__ ext_dst, float* __restrict__ st_h, float* __restrict__ st_c,
                                 float* __restrict__ xnext, float* __restrict__ htop, int H, int L, int l, int R, int nsplit,
                                 int frag, size_t xplane) {
  const int r = blockIdx.x;
  const int src = ext_src[r], dst = ext_dst[r];
  const float* g = gates + (size_t)r * 4 * H;
  const size_t part = (size_t)R * 4 * H;  // stride between the K-slice partial sums
  for (int u = threadIdx.x; u < H; u += blockDim.x) {
    const float c_old = (dst >= 0 && src >= 0) ? st_c[((size_t)src * L + l) * H + u] : 0.f;
    float pre[4];
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) {
      float pz[KSPLIT];
#pragma unroll
      for (int z = 0; z < KSPLIT; ++z) pz[z] = z < nsplit ? g[z * part + gate * H + u] : 0.f;   // independent loads ...
      float v = pz[0];
#pragma unroll
      for (int z = 1; z < KSPLIT; ++z)
        if (z < nsplit) v += pz[z];                                                            // ... added in slice order
      pre[gate] = v;
    }
    const float gi = 1.f / (1.f + expf(-pre[0])), gf = 1.f / (1.f + expf(-pre[1]));
    const float gg = tanhf(pre[2]), go = 1.f / (1.f + expf(-pre[3]));
    const float c_new = gf * c_old + gi * gg;
    const float h_new = go * tanhf(c_new);
    float h_next_src = 0.f;
    if (l + 1 < L && dst >= 0 && src >= 0) h_next_src = st_h[((size_t)src * L + l + 1) * H + u];
    if (dst >= 0) {
      st_h[((size_t)dst * L + l) * H + u] = h_new;
      st_c[((size_t)dst * L + l) * H + u] = c_new;
    }
    if (l + 1 < L) {
      if (frag) {
        store_split3(reinterpret_cast<unsigned short*>(xnext), xplane, frag_off(r, u, 2 * H), dst >= 0 ? h_new : 0.f);
        store_split3(reinterpret_cast<unsigned short*>(xnext), xplane, frag_off(r, H + u, 2 * H), h_next_src);
      } else {
        xnext[(size_t)r * 2 * H + u] = dst >= 0 ? h_new : 0.f;
        xnext[(size_t)r * 2 * H + H + u] = h_next_src;
      }
    } else {
      htop[(size_t)r * H + u] = dst >= 0 ? h_new : 0.f;
    }
  }
}

// ---- one predictor layer in ONE launch (round 4): gate GEMM + cell.  The split-K gate GEMM left 4 x R x 4H partial sums in
// memory for a second kernel to add (8.4 MB written and read back per layer-step at configs[3], 17.8 + 9.5 us and a kernel
// boundary, four times per frame: half of the beam decode).  Here a workgroup owns EIGHT hidden units -- their four gates are
// the 32 rows of one fragment group of the unit-major weight planes (pack_cat3_kernel) -- and 32 RG hypothesis rows; its eight
// waves split the rows into RG groups and K into 8 / RG slices, meet in LDS, and the workgroup applies the cell to its
// (row, unit) pairs itself: c_old gathered from the source slot, (h, c) written to the destination slot, the next layer's
// operand row [h' | h_src] (or the htop row) written as the GEMM wrote nothing.  Partial sums are added in slice order, slice 0
// carrying the bias, as lstm_cell_kernel did with the K-slice launches.  256 workgroups at R = 128 (one per CU, each streaming
// its 32 weight rows once), 128 at R <= 32.  The next layer's rows go to the OTHER x buffer: other workgroups still read this one.
template <int RG>
__global__ __launch_bounds__(512) void pred_layer_fused_kernel(const unsigned short* __restrict__ x, size_t xplane,
                                                               const unsigned short* __restrict__ wp,
                                                               const float* __restrict__ bias, const int32_t* __restrict__ ext_src,
                                                               const int32_t* __restrict__ ext_dst, float* __restrict__ st_h,
                                                               float* __restrict__ st_c, unsigned short* __restrict__ xnext,
                                                               float* __restrict__ htop, int R, int K, int H, int L, int l) {
  constexpr int KQ = 8 / RG;
  __shared__ float red[8][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int rg = wave % RG, kq = wave / RG;
  const int b = blockIdx.x, rb = blockIdx.y;
  const int m = rb * 32 * RG + rg * 32 + l31;                  // this lane's x row
  const int kslice = K / KQ, k0 = kq * kslice;
  const size_t wplane = (size_t)4 * H * K;
  const unsigned short* xp = x + frag_off(m, k0 + half * 8, K);
  const unsigned short* w0 = wp + frag_off(b * 32 + l31, k0 + half * 8, K);
  const bool row_ok = m < R;

  f32x16r acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PF = 4;
  u32x4r xf[PF][3], wf[PF][3];
  auto load = [&](int slot, int ks) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      xf[slot][pl] = u32x4r{0u, 0u, 0u, 0u};
      if (row_ok) xf[slot][pl] = *reinterpret_cast<const u32x4r*>(xp + (size_t)pl * xplane + (size_t)ks * 512);
      wf[slot][pl] = *reinterpret_cast<const u32x4r*>(w0 + (size_t)pl * wplane + (size_t)ks * 512);
    }
  };
  auto step = [&](int slot) {
    const bf16x8r bh = __builtin_bit_cast(bf16x8r, wf[slot][0]), bm = __builtin_bit_cast(bf16x8r, wf[slot][1]);
    const bf16x8r bl = __builtin_bit_cast(bf16x8r, wf[slot][2]);
    const bf16x8r xh = __builtin_bit_cast(bf16x8r, xf[slot][0]), xm = __builtin_bit_cast(bf16x8r, xf[slot][1]);
    const bf16x8r xl = __builtin_bit_cast(bf16x8r, xf[slot][2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc, 0, 0, 0);   // smallest terms first (pred_gates_split_kernel)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc, 0, 0, 0);
  };
  const int nks = kslice / 16;
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < nks) load(j, j);
  for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (ks + j < nks) {
        step(j);
        if (ks + j + PF < nks) load(j, ks + j + PF);
      }
  }
  // D[row = (r & 3) + 8 (r >> 2) + 4 half of the wave's 32][col = l31 = gate * 8 + unit % 8]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
  __syncthreads();
  const int t = threadIdx.x;
  if (t >= 256 * RG) return;
  const int rowl = t >> 3, u8 = t & 7, rgi = rowl >> 5, rr = rowl & 31;
  const int r = rb * 32 * RG + rowl, u = b * 8 + u8;
  if (r >= R) return;
  float pre[4];
#pragma unroll
  for (int gate = 0; gate < 4; ++gate) {
    float v = red[rgi][rr][gate * 8 + u8] + bias[b * 32 + gate * 8 + u8];      // slice 0 (wave index = kq RG + rg)
#pragma unroll
    for (int q = 1; q < KQ; ++q) v += red[q * RG + rgi][rr][gate * 8 + u8];
    pre[gate] = v;
  }
  const int src = ext_src[r], dst = ext_dst[r];
  const float c_old = (dst >= 0 && src >= 0) ? st_c[((size_t)src * L + l) * H + u] : 0.f;
  const float gi = 1.f / (1.f + expf(-pre[0])), gf = 1.f / (1.f + expf(-pre[1]));
  const float gg = tanhf(pre[2]), go = 1.f / (1.f + expf(-pre[3]));
  const float c_new = gf * c_old + gi * gg;
  const float h_new = go * tanhf(c_new);
  float h_next_src = 0.f;
  if (l + 1 < L && dst >= 0 && src >= 0) h_next_src = st_h[((size_t)src * L + l + 1) * H + u];
  if (dst >= 0) {
    st_h[((size_t)dst * L + l) * H + u] = h_new;
    st_c[((size_t)dst * L + l) * H + u] = c_new;
  }
  if (l + 1 < L) {
    store_split3(xnext, xplane, frag_off(r, u, 2 * H), dst >= 0 ? h_new : 0.f);
    store_split3(xnext, xplane, frag_off(r, H + u, 2 * H), h_next_src);
  } else {
    htop[(size_t)r * H + u] = dst >= 0 ? h_new : 0.f;
  }
}

// log_softmax(W_out . tanh(enc_p[t, i] + pp[slot]) + b_out) for hypothesis row r = i*w + j, if it is live.
// A row whose predictor state was requested by the previous round (ext_dst[r] >= 0) finds its projected predictor
// output as KSPLIT partial sums in pp_tmp[.][r]: they are added here in slice order and the sum is also stored in
// pp[slot] (the copy that survives the frame), which used to be a kernel of its own.
// CHUNK (greedy, round 3): `w` consecutive frames of utterance i against ONE predictor output: row r = i * w + c is frame
// A_cnt[i] + c (A_cnt = the utterance's current frame), its predictor row (slot, request, pp_tmp row) is i; row c == 0 commits
// the projected predictor output, the others add the same partial sums in the same order without storing them.
template <bool CHUNK>
__global__ __launch_bounds__(256) void joint_slots_kernel(const float* __restrict__ enc_p, const int32_t* __restrict__ lens,
                                                          float* __restrict__ pp, const float* __restrict__ pp_tmp,
                                                          const int32_t* __restrict__ ext_dst,
                                                          const int32_t* __restrict__ A_slot,
                                                          const int32_t* __restrict__ A_cnt, const float* __restrict__ w_out,
                                                          const float* __restrict__ b_out, float* __restrict__ logp, int t,
                                                          int N, int w, int J, int V1, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* z = smem;
  float* lg = smem + J;
  const int r = blockIdx.x, i = r / w, j = r - i * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // four independent loads (one memory round trip), then the test: these words were written by the previous kernel on
  // another XCD, every dependent global load here costs ~2 us
  const int pr = CHUNK ? i : r;                     // row of the predictor-side arrays
  const int len_i = lens[i], cnt_i = A_cnt[i], slot = A_slot[pr], fresh = ext_dst[pr];
  if (CHUNK) t = cnt_i + j;
  // Everything that does not depend on those words is requested before they are looked at: this wave's W_out rows
  // (symbols v = wave, wave + 4, ...) and the frame's encoder projection.
  constexpr int KMAX = 8, VMAX = 8;  // register tile: J <= 512, V1 <= 32
  const bool reg_tile = J <= 64 * KMAX && V1 <= 4 * VMAX;
  float wreg[KMAX][VMAX];
  if (reg_tile) {
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk)
#pragma unroll
      for (int vi = 0; vi < VMAX; ++vi) {
        const int k = lane + 64 * kk, v = wave + 4 * vi;
        wreg[kk][vi] = (k < J && v < V1) ? w_out[(size_t)v * J + k] : 0.f;
      }
  }
  float ereg[4] = {0.f, 0.f, 0.f, 0.f};
  const bool e_tile = J <= 1024;
  if (e_tile && !CHUNK) {
    const float* e = enc_p + ((size_t)t * N + i) * J;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (tid + 256 * m < J) ereg[m] = e[tid + 256 * m];
  }
  if (t >= len_i || (!CHUNK && j >= cnt_i)) return;
  float* p = pp + (size_t)slot * J;
  const size_t part = (size_t)R * J;
  auto pred_term = [&](int k) {
    if (fresh < 0) return p[k];
    float pv = pp_tmp[(size_t)pr * J + k];
#pragma unroll
    for (int zz = 1; zz < KSPLIT; ++zz) pv += pp_tmp[zz * part + (size_t)pr * J + k];
    if (!CHUNK || j == 0) p[k] = pv;
    return pv;
  };
  if (e_tile) {
    // (CHUNK: the frame is known only now, so the encoder row is loaded here)
    if (CHUNK) {
      const float* ec = enc_p + ((size_t)t * N + i) * J;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (tid + 256 * m < J) ereg[m] = ec[tid + 256 * m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = tid + 256 * m;
      if (k < J) z[k] = tanhf(ereg[m] + pred_term(k));
    }
  } else {
    const float* ec = enc_p + ((size_t)t * N + i) * J;
    for (int k = tid; k < J; k += 256) z[k] = tanhf(ec[k] + pred_term(k));
  }
  __syncthreads();
  if (reg_tile) {
    float acc[VMAX];
#pragma unroll
    for (int vi = 0; vi < VMAX; ++vi) acc[vi] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
      const int k = lane + 64 * kk;
      const float zk = k < J ? z[k] : 0.f;
#pragma unroll
      for (int vi = 0; vi < VMAX; ++vi) acc[vi] += wreg[kk][vi] * zk;
    }
#pragma unroll
    for (int vi = 0; vi < VMAX; ++vi) {
      const int v = wave + 4 * vi;
      if (v < V1) {
        const float tot = wsum(acc[vi]);
        if (lane == 0) lg[v] = tot + (b_out ? b_out[v] : 0.f);
      }
    }
  } else if (V1 <= 64) {
    // each wave owns symbols v = wave, wave + 4, ...; their dot products run side by side so the loads of the W_out
    // rows are independent and pipeline (a loop over v pays one L2 round trip per symbol)
    float acc[16];
#pragma unroll
    for (int vi = 0; vi < 16; ++vi) acc[vi] = 0.f;
    for (int k = lane; k < J; k += 64) {
      const float zk = z[k];
#pragma unroll
      for (int vi = 0; vi < 16; ++vi) {
        const int v = wave + 4 * vi;
        if (v < V1) acc[vi] += w_out[(size_t)v * J + k] * zk;
      }
    }
#pragma unroll
    for (int vi = 0; vi < 16; ++vi) {
      const int v = wave + 4 * vi;
      if (v < V1) {
        const float tot = wsum(acc[vi]);
        if (lane == 0) lg[v] = tot + (b_out ? b_out[v] : 0.f);
      }
    }
  } else {
    for (int v = wave; v < V1; v += 4) {
      const float* wr = w_out + (size_t)v * J;
      float acc = 0.f;
      for (int k = lane; k < J; k += 64) acc += wr[k] * z[k];
      acc = wsum(acc);
      if (lane == 0) lg[v] = acc + (b_out ? b_out[v] : 0.f);
    }
  }
  __syncthreads();
  if (wave == 0) {
    float m = -INFINITY;
    for (int v = lane; v < V1; v += 64) m = fmaxf(m, lg[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int v = lane; v < V1; v += 64) sum += expf(lg[v] - m);
    sum = wsum(sum);
    const float lz = logf(sum) + m;
    for (int v = lane; v < V1; v += 64) logp[(size_t)r * V1 + v] = lg[v] - lz;
  }
}

struct BeamP {
  const int32_t* lens;
  const float* logp;
  int32_t *A_cnt, *A_node, *A_slot;
  float* A_score;
  int32_t *B_cnt, *B_node, *B_slot;
  float* B_score;
  int32_t *ext_label, *ext_src, *ext_dst;
  int32_t *node_cnt, *node_parent, *node_label, *child;
  int N, w, V, bcap, maxn, R;
};

// One emission round of one utterance (one workgroup).  Everything the round needs is fetched by independent loads at
// the top (one memory round trip: the words were written by the previous kernel, usually on another XCD), the list
// surgery then runs in LDS, and the trie insertions of the picked extensions go out in parallel.
__global__ __launch_bounds__(256) void beam_round_kernel(BeamP p, int t, int region, int first, int last) {
  extern __shared__ __attribute__((aligned(16))) float cand[];  // [w * V1] log-probabilities, then candidate scores
  __shared__ int wi[4];
  __shared__ int pick_idx[32];
  __shared__ float pick_val[32];
  __shared__ int old_node[32], old_slot[32];
  __shared__ float old_score[32];
  __shared__ int b_node[128], b_slot[128];
  __shared__ float b_score[128];
  __shared__ int n_pick, next_node;
  __shared__ int match_at[32];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = p.w, V1 = p.V + 1, blank = p.V;
  const int len_i = p.lens[i], cnt = p.A_cnt[i], bc0 = first ? 0 : p.B_cnt[i], nodes0 = p.node_cnt[i];
  if (tid < w) {
    old_node[tid] = p.A_node[i * w + tid];
    old_slot[tid] = p.A_slot[i * w + tid];
    old_score[tid] = p.A_score[i * w + tid];
    p.ext_dst[i * w + tid] = -1;  // requests default to "none" (also for utterances that have ended)
  }
  if (tid < p.bcap) {
    b_node[tid] = p.B_node[(size_t)i * p.bcap + tid];
    b_score[tid] = p.B_score[(size_t)i * p.bcap + tid];
    b_slot[tid] = p.B_slot[(size_t)i * p.bcap + tid];
  }
  for (int c = tid; c < w * V1; c += 256) cand[c] = p.logp[(size_t)i * w * V1 + c];
  if (t >= len_i) return;
  if (tid == 0) { n_pick = 0; next_node = nodes0; }
  if (tid < 32) match_at[tid] = -1;
  __syncthreads();
  // blank transitions, in hypothesis order: same prefix (trie node) -> logaddexp, else append (first arrival keeps its
  // predictor state).  The live hypotheses' nodes are distinct (picks are distinct (parent, label) pairs, the frame's first
  // set comes out of B, whose nodes are distinct), so the hypotheses do not interact: every (hypothesis, B entry) pair is
  // compared at once and wave 0 merges / appends, an append's position = its rank among the unmatched (round 4: thread 0
  // walked cnt x bc dependent LDS reads, ~10 of the kernel's 14.7 us).  Should two live hypotheses ever share a node the
  // serial walk below does what it always did.
  for (int idx = tid; idx < cnt * bc0; idx += 256) {
    const int j = idx / bc0, b = idx - j * bc0;
    if (b_node[b] == old_node[j]) match_at[j] = b;
  }
  __syncthreads();
  if (wave == 0) {
    const int j = lane;
    bool dup = false;
    if (j < cnt)
      for (int k = 0; k < j; ++k) dup |= old_node[k] == old_node[j];
    if (__any(dup)) {
      if (lane == 0) {
        int bc = bc0;
        for (int jj = 0; jj < cnt; ++jj) {
          const float s = old_score[jj] + cand[jj * V1 + blank];
          int at = -1;
          for (int b = 0; b < bc; ++b)
            if (b_node[b] == old_node[jj]) { at = b; break; }
          if (at >= 0) {
            b_score[at] = logaddexp32(b_score[at], s);
          } else if (bc < p.bcap) {
            b_node[bc] = old_node[jj];
            b_score[bc] = s;
            b_slot[bc] = old_slot[jj];
            ++bc;
          }
        }
        p.B_cnt[i] = bc;
        wi[0] = bc;
      }
    } else {
      const bool have = j < cnt;
      const int at = have ? match_at[j] : 0;
      const bool unmatched = have && at < 0;
      const unsigned long long um = __ballot(unmatched);
      const int pos = bc0 + __popcll(um & ((1ull << lane) - 1ull));
      if (have) {
        const float s = old_score[j] + cand[j * V1 + blank];
        if (at >= 0) {
          b_score[at] = logaddexp32(b_score[at], s);
        } else if (pos < p.bcap) {
          b_node[pos] = old_node[j];
          b_score[pos] = s;
          b_slot[pos] = old_slot[j];
        }
      }
      if (lane == 0) {
        const int bc = min(bc0 + (int)__popcll(um), p.bcap);
        p.B_cnt[i] = bc;
        wi[0] = bc;
      }
    }
    if (lane == 0 && last) p.A_cnt[i] = 0;
  }
  __syncthreads();
  {
    const int bc = wi[0];
    if (tid < bc) {
      p.B_node[(size_t)i * p.bcap + tid] = b_node[tid];
      p.B_score[(size_t)i * p.bcap + tid] = b_score[tid];
      p.B_slot[(size_t)i * p.bcap + tid] = b_slot[tid];
    }
  }
  if (last) return;
  __syncthreads();
  // label extensions: the w best of cnt * V1 candidates (score desc, ties -> lowest flat index), blank and
  // non-finite scores excluded
  const int C = cnt * V1;
  for (int c = tid; c < C; c += 256) {
    const int j = c / V1, k = c - j * V1;
    const float s = old_score[j] + cand[c];
    cand[c] = (k == blank || !isfinite(s)) ? -INFINITY : s;
  }
  __syncthreads();
  // the w best, one after the other, by ONE wave (round 4: all four waves with two barriers and a serial merge per pick took
  // most of the kernel's 14.7 us): a lane scans its candidates, the wave reduces (value desc, ties -> lowest flat index), lane 0
  // records the pick and removes it -- a wave's LDS operations execute in order, so the next scan sees the removal
  if (wave == 0) {
    int npk = 0;
    for (int q = 0; q < w; ++q) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < C; c += 64) {
        const float v = cand[c];
        if (v > bv) { bv = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (bi == 0x7fffffff || !(bv > -INFINITY)) break;      // wave-uniform: nothing finite is left
      if (lane == 0) {
        pick_idx[npk] = bi;
        pick_val[npk] = bv;
        cand[bi] = -INFINITY;
      }
      ++npk;
    }
    if (lane == 0) n_pick = npk;
  }
  __syncthreads();
  const int np = n_pick;
  if (tid < np) {
    // distinct picks are distinct (parent, label) pairs, so their trie insertions are independent; node numbers are
    // identities only, an atomic counter hands them out
    const int q = tid;
    const int hi = pick_idx[q] / V1, k = pick_idx[q] - hi * V1;
    const int parent = old_node[hi];
    int32_t* ch = p.child + ((size_t)i * p.maxn + parent) * p.V + k;
    int node = *ch;
    if (node < 0) {
      node = atomicAdd(&next_node, 1);
      if (node < p.maxn) {
        p.node_parent[(size_t)i * p.maxn + node] = parent;
        p.node_label[(size_t)i * p.maxn + node] = k;
        *ch = node;
      } else {
        node = parent;  // cannot happen: maxn bounds every possible insertion
      }
    }
    const int slot = region * p.R + i * w + q;
    p.A_node[i * w + q] = node;
    p.A_score[i * w + q] = pick_val[q];
    p.A_slot[i * w + q] = slot;
    p.ext_label[i * w + q] = k;
    p.ext_src[i * w + q] = old_slot[hi];
    p.ext_dst[i * w + q] = slot;
  }
  __syncthreads();
  if (tid == 0) {
    p.A_cnt[i] = np;
    p.node_cnt[i] = min(next_node, p.maxn);
  }
}

// End of frame t: the beam_width best entries of B (stable, descending) become the live set of the next frame; their
// predictor states are copied into frame-start region (t + 1) & 1.  Grid (w, N): every workgroup repeats the (cheap, LDS)
// selection and moves ONE survivor's state with all its loads in flight at once.
__global__ __launch_bounds__(256) void beam_frame_end_kernel(BeamP p, float* st_h, float* st_c, float* pp, int t, int LH, int J) {
  __shared__ int b_node[128], b_slot[128];
  __shared__ float b_score[128];
  __shared__ int sel_src[32], sel_node[32];
  __shared__ float sel_score[32];
  __shared__ int n_sel;
  const int q = blockIdx.x, i = blockIdx.y, tid = threadIdx.x;
  const int w = p.w;
  const int len_i = p.lens[i], bc = p.B_cnt[i];
  if (tid < p.bcap) {
    b_node[tid] = p.B_node[(size_t)i * p.bcap + tid];
    b_score[tid] = p.B_score[(size_t)i * p.bcap + tid];
    b_slot[tid] = p.B_slot[(size_t)i * p.bcap + tid];
  }
  if (t >= len_i) return;
  __syncthreads();
  // the beam_width best of B, one after the other (first maximum = earliest arrival among equals), by ONE wave: a lane holds
  // entries lane and lane + 64, the wave reduces (an entry before none, score desc, index asc), the owner marks its entry taken
  // (round 4: thread 0 walked w x bc dependent LDS reads, ~9 us, in every one of the w x N workgroups)
  if (tid < 64) {
    const int lane = tid;
    const float s0 = lane < bc ? b_score[lane] : 0.f, s1 = lane + 64 < bc ? b_score[lane + 64] : 0.f;
    bool t0 = !(lane < bc), t1 = !(lane + 64 < bc);          // "taken" also stands for "does not exist"
    int ns = 0;
    for (int x = 0; x < w && x < bc; ++x) {
      bool have = false;
      float bv = 0.f;
      int bi = 0x7fffffff;
      if (!t0) { have = true; bv = s0; bi = lane; }
      if (!t1 && (!have || s1 > bv)) { have = true; bv = s1; bi = lane + 64; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const bool oh = __shfl_xor((int)have, o, 64) != 0;
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        // the serial walk keeps the FIRST maximum: a later entry replaces the best only if its score is greater
        const bool take = oh && (!have || ov > bv || (!(bv > ov) && oi < bi));
        if (take) { have = true; bv = ov; bi = oi; }
      }
      if (!have) break;                                       // wave-uniform
      if (bi == lane) t0 = true;
      if (bi == lane + 64) t1 = true;
      if (lane == 0) {
        sel_src[ns] = b_slot[bi];
        sel_node[ns] = b_node[bi];
        sel_score[ns] = b_score[bi];
      }
      ++ns;
    }
    if (lane == 0) n_sel = ns;
  }
  __syncthreads();
  const int ns = n_sel;
  if (q == 0 && tid == 0) p.A_cnt[i] = ns;
  if (q >= ns) return;
  const size_t src = (size_t)sel_src[q], dst = (size_t)((t + 1) & 1) * p.R + i * w + q;
  for (int base = 0; base < LH; base += 256 * 8) {
    float vh[8], vc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int k = base + tid + 256 * m;
      if (k < LH) { vh[m] = st_h[src * LH + k]; vc[m] = st_c[src * LH + k]; }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int k = base + tid + 256 * m;
      if (k < LH) { st_h[dst * LH + k] = vh[m]; st_c[dst * LH + k] = vc[m]; }
    }
  }
  for (int k = tid; k < J; k += 256) pp[dst * J + k] = pp[src * J + k];
  if (tid == 0) {
    p.A_node[i * w + q] = sel_node[q];
    p.A_score[i * w + q] = sel_score[q];
    p.A_slot[i * w + q] = (int)dst;
  }
}

// Best hypothesis (highest score, first among equals) -> label sequence by walking the trie to the root.
__global__ void beam_finish_kernel(BeamP p, int32_t* out_idx, int32_t* out_len, float* out_score, int out_stride) {
  const int i = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int cnt = p.A_cnt[i];
  int best = -1;
  for (int j = 0; j < cnt; ++j)
    if (best < 0 || p.A_score[i * p.w + j] > p.A_score[i * p.w + best]) best = j;
  int len = 0;
  if (best >= 0) {
    int node = p.A_node[i * p.w + best];
    for (int n = node; n > 0; n = p.node_parent[(size_t)i * p.maxn + n]) ++len;
    int at = len;
    for (int n = node; n > 0 && at > 0; n = p.node_parent[(size_t)i * p.maxn + n])
      out_idx[(size_t)i * out_stride + --at] = p.node_label[(size_t)i * p.maxn + n];
  }
  out_len[i] = len;
  if (out_score) out_score[i] = best >= 0 ? p.A_score[i * p.w + best] : -INFINITY;
}

// root hypothesis of every utterance + the request that creates its predictor state (blank on the zero state)
__global__ void beam_init_kernel(BeamP p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.N) return;
  for (int q = 0; q < p.w; ++q) p.ext_dst[i * p.w + q] = -1;
  p.ext_label[i * p.w] = p.V;
  p.ext_src[i * p.w] = -1;
  p.ext_dst[i * p.w] = i * p.w;  // region 0, row i*w
  p.A_cnt[i] = 1;
  p.A_node[i * p.w] = 0;
  p.A_score[i * p.w] = 0.f;
  p.A_slot[i * p.w] = i * p.w;
  p.B_cnt[i] = 0;
  p.node_cnt[i] = 1;
}

// ---- greedy -----------------------------------------------------------------------------------------------------

__global__ void greedy_init_kernel(int32_t* ext_label, int32_t* ext_src, int32_t* ext_dst, int32_t* slot, int32_t* out_cnt,
                                   int32_t* cur_t, int32_t* sym, int32_t* done_cnt, int N, int blank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *done_cnt = 0;
  if (i >= N) return;
  ext_label[i] = blank;
  ext_src[i] = -1;
  ext_dst[i] = i;
  slot[i] = i;
  out_cnt[i] = 0;
  cur_t[i] = 0;
  sym[i] = 0;
}

// One wave per utterance: walk the chunk's frames in order.  A frame's symbol is the argmax of its log-probabilities
// (first maximum); blank moves on to the next frame; a label is emitted, requests a predictor step in place (source slot =
// destination slot = i) and ends the walk, because every later joint of the chunk was computed against the old prediction.
// After max_symbols labels on one frame the frame advances.  An utterance whose frame reaches its length is finished.
__global__ __launch_bounds__(64) void greedy_scan_kernel(const float* __restrict__ logp, const int32_t* __restrict__ lens,
                                                         int32_t* __restrict__ cur_t, int32_t* __restrict__ sym,
                                                         int32_t* __restrict__ done_cnt, int32_t* __restrict__ out_idx,
                                                         int32_t* __restrict__ out_cnt, int32_t* __restrict__ ext_label,
                                                         int32_t* __restrict__ ext_src, int32_t* __restrict__ ext_dst, int V1,
                                                         int blank, int out_stride, int max_symbols) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int len_i = lens[i], t0 = cur_t[i];
  int nsym = sym[i];
  if (lane == 0) ext_dst[i] = -1;
  if (t0 >= len_i) return;                                   // finished in an earlier iteration
  int t = t0, emitted = -1;
  for (int c = 0; c < GREEDY_CHUNK && t < len_i; ++c) {       // row c of the chunk is frame t0 + c == t on every pass
    const float* row = logp + ((size_t)i * GREEDY_CHUNK + c) * V1;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V1; v += 64) {
      const float x = row[v];
      if (x > bv || (x == bv && v < bi) || bi == 0x7fffffff) { bv = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (bi == blank) {
      ++t;
      nsym = 0;
      continue;
    }
    emitted = bi;
    if (++nsym >= max_symbols) {                              // the frame's quota is used up: the next joint is frame t + 1's
      ++t;
      nsym = 0;
    }
    break;
  }
  if (lane == 0) {
    if (emitted >= 0) {
      const int n = out_cnt[i];
      if (n < out_stride) out_idx[(size_t)i * out_stride + n] = emitted;
      out_cnt[i] = n + 1;
      ext_label[i] = emitted;
      ext_src[i] = i;
      ext_dst[i] = i;
    }
    cur_t[i] = t;
    sym[i] = nsym;
    if (t >= len_i) atomicAdd(done_cnt, 1);
  }
}

// The greedy decode polls a device counter (utterances finished) through a small pinned ring; allocating pinned memory or
// freeing it synchronises the whole device (a decode on one stream of pipeline.TwoBatchesInFlight would stall the other), so
// ring and events are made once per (host thread, device) and kept.
struct GreedyPoll {
  static constexpr int RING = 4;
  int32_t* host = nullptr;
  hipEvent_t ev[RING] = {nullptr, nullptr, nullptr, nullptr};
};
GreedyPoll* greedy_poll() {
  static thread_local GreedyPoll polls[64];
  static thread_local unsigned char state[64];       // 0 = not made, 1 = ready, 2 = failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  dev &= 63;
  if (state[dev] == 0) {
    GreedyPoll& g = polls[dev];
    bool ok = hipHostMalloc((void**)&g.host, GreedyPoll::RING * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
    for (int k = 0; ok && k < GreedyPoll::RING; ++k) ok = hipEventCreateWithFlags(&g.ev[k], hipEventDisableTiming) == hipSuccess;
    state[dev] = ok ? 1 : 2;
  }
  return state[dev] == 1 ? &polls[dev] : nullptr;
}


// =====================================================================================================================
// The beam decode's predictor, round 5 (154 us per frame in 16 dependent launches).  The same
// arithmetic -- every float32 product as an exact three-way bf16 split, float32 accumulation -- cut differently:
//
//   gates_0 = W_ih0 . emb[label] + W_hh0 . h0[parent] + b0      gates_l = W_ihl . h'_{l-1} + W_hhl . h_l[parent] + b_l
//             `---- E[label] ---'  `--- G_0[parent] ---'                                    `--- G_l[parent] ---'
//
//   * E = W_ih0 . emb + b0 is a table of V + 1 rows, made once per decode;
//   * G_l[slot] = W_hhl . h_l[slot] depends on a STATE only, not on what is appended to it: it is computed once per state,
//     right after the state is made (pred_gemm3_kernel<.., false>, all layers in one launch), and kept beside it in the pool;
//   * so a predictor step's chain is: round -> cell 0 of the picks (elementwise: E[label] + G_0[parent], in the round kernel
//     itself) -> W_ihl . h'_{l-1} + G_l[parent] -> cell l (pred_gemm3_kernel<.., true>, K = H instead of In + H) -> joint;
//   * pred_proj and the joint network are ONE kernel (beam2_joint_kernel): a workgroup owns 64 hypothesis rows x 32 joint
//     units, contracts h_top against its W_pred rows, adds the encoder projection, applies tanh and leaves the 32-unit partial
//     sums of the output layer; the round kernel adds the J / 32 partials in slice order and normalises;
//   * the last round of a frame and the frame end are one kernel (beam2_frame_end_kernel).
// A frame at max_symbols = 3, two layers: 10 launches (joint, round | layer 1, G, joint, round | layer 1, G, joint, frame end).
// Shapes: H % 64 == 0, J % 32 == 0; anything else takes the round-4 sequence.  MS_RNNT_V2=0 (read per call) selects it too.

// rows [rows][K] float32 -> three fragment-major bf16 planes; unit_major: gate row g H + u -> row (u / 8) 32 + g 8 + u % 8
__global__ void pack_rows3_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, int rows, int K, int H_units, int two) {
  const int src_row = blockIdx.x;
  int row = src_row;
  if (H_units > 0) {
    const int gate_ = src_row / H_units, unit_ = src_row % H_units;
    row = (unit_ >> 3) * 32 + gate_ * 8 + (unit_ & 7);
  }
  const size_t plane = (size_t)rows * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    store_split(planes, plane, frag_off(row, k, K), w[(size_t)src_row * K + k], two);
}

// out[unit-major row] = a[row] + b[row] (+ tab[v][row] for a table of nv rows)
__global__ void unit_major_add_kernel(const float* __restrict__ tab, const float* __restrict__ a, const float* __restrict__ b,
                                      float* __restrict__ out, int H, int nv) {
  const int src = blockIdx.x * blockDim.x + threadIdx.x;
  if (src >= 4 * H) return;
  const int gate_ = src / H, unit_ = src % H;
  const int row = (unit_ >> 3) * 32 + gate_ * 8 + (unit_ & 7);
  const float bias = (a ? a[src] : 0.f) + (b ? b[src] : 0.f);
  if (tab == nullptr) { out[row] = bias; return; }
  for (int v = 0; v < nv; ++v) out[(size_t)v * 4 * H + row] = tab[(size_t)v * 4 * H + src] + bias;
}

struct Beam2P {
  const int32_t* lens;
  int32_t *A_cnt, *A_node, *A_slot;     // this frame's live set (the rounds replace it in place: one workgroup per utterance)
  float* A_score;
  int32_t *nA_cnt, *nA_node, *nA_slot;  // the next frame's, written by the frame end (its w workgroups per utterance all READ
  float* nA_score;                      // this frame's set, so they must not write it)
  int32_t *B_cnt, *B_node, *B_slot;
  float* B_score;
  int32_t *ext_label, *ext_src, *ext_dst;
  int32_t *node_cnt, *node_parent, *node_label, *child;
  float *st_h, *st_c, *pp, *G;
  const float *E, *bias, *plog, *b_out;
  unsigned short* hpl;          // [L][3][rpad * H] planes of the rows' new h of every layer
  size_t hplane;                // rpad * H
  int N, w, V, bcap, maxn, R, H, L, J, NS;
  int two;                      // operand planes of the step's GEMMs: 0 = three bf16 (exact), 1 = two fp16 (store_split)
};

__device__ __forceinline__ void lstm_cell(const float pre[4], float c_old, float& h_new, float& c_new) {
  const float gi = 1.f / (1.f + expf(-pre[0])), gf = 1.f / (1.f + expf(-pre[1]));
  const float gg = tanhf(pre[2]), go = 1.f / (1.f + expf(-pre[3]));
  c_new = gf * c_old + gi * gg;
  h_new = go * tanhf(c_new);
}

// cell 0 of every request row (label k appended to the state in slot src; -1 = the zero state): E[k] + G_0[src].  One
// workgroup per row; a thread's units' loads are all issued before the first is used (a unit at a time waited for an L2 round
// trip per unit: 40 us inside the round kernel).
__global__ __launch_bounds__(256) void beam2_cell0_kernel(Beam2P p) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int dst = p.ext_dst[r], src = p.ext_src[r], lab = p.ext_label[r];
  if (dst < 0) return;
  const int k = min(max(lab, 0), p.V);
  const int H = p.H, L = p.L;
  const float* e = p.E + (size_t)k * 4 * H;
  const float* g = src >= 0 ? p.G + ((size_t)src * L + 0) * 4 * H : nullptr;
  const float* cs = src >= 0 ? p.st_c + ((size_t)src * L + 0) * H : nullptr;
  constexpr int UB = 4;                                      // units per thread and batch
  for (int u0 = 0; u0 < H; u0 += 256 * UB) {
    float ev[UB][4], gv[UB][4], cv[UB];
#pragma unroll
    for (int m = 0; m < UB; ++m) {
      const int u = u0 + tid + 256 * m;
      const int o = (u >> 3) * 32 + (u & 7);
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) {
        ev[m][gate] = u < H ? e[o + gate * 8] : 0.f;
        gv[m][gate] = (u < H && g) ? g[o + gate * 8] : 0.f;
      }
      cv[m] = (u < H && cs) ? cs[u] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < UB; ++m) {
      const int u = u0 + tid + 256 * m;
      if (u >= H) continue;
      float pre[4];
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) pre[gate] = ev[m][gate] + gv[m][gate];
      float h_new, c_new;
      lstm_cell(pre, cv[m], h_new, c_new);
      p.st_h[((size_t)dst * L + 0) * H + u] = h_new;
      p.st_c[((size_t)dst * L + 0) * H + u] = c_new;
      store_split(p.hpl, p.hplane, frag_off(r, u, H), h_new, p.two);
    }
  }
}

// acc[64 rows x 32 weight rows] over K = H for one (weight-row group b, row block rb): eight waves = 2 row groups x 4 K
// slices, three-way split products (pred_layer_fused_kernel's loop), partial sums met in LDS in slice order.
// CELL: weight rows = W_ih of layer l (unit-major), x = the rows' new h of layer l - 1; pre = sum + bias + G_l[src] -> cell
//       -> state of the destination slot, the rows' new h of layer l as planes.
// !CELL: blockIdx.z = layer; weight rows = W_hh of that layer, x = the rows' new h of that layer; G_l[dst] = sum.
template <bool CELL, int RG, bool TWO>
__device__ __forceinline__ void pred_gemm3_body(const Beam2P& p, const unsigned short* __restrict__ wp_base, int l, int b, int rb,
                                                float (*red)[32][33]) {
  constexpr int KQ = 8 / RG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int rg = wave % RG, kq = wave / RG;
  const int H = p.H, L = p.L, K = H, R = p.R;
  const int lx = CELL ? l - 1 : l;                                   // layer whose new h is the x operand
  const size_t wplane = (size_t)4 * H * K;
  const unsigned short* wp = wp_base + (CELL ? (size_t)(l - 1) : (size_t)l) * 3 * wplane;
  const unsigned short* x = p.hpl + (size_t)lx * 3 * p.hplane;
  const int m = rb * 32 * RG + rg * 32 + l31;                        // this lane's x row (planes are padded to 64 rows)
  const int kslice = K / KQ, k0 = kq * kslice;
  const unsigned short* xp = x + frag_off(m, k0 + half * 8, K);
  const unsigned short* w0 = wp + frag_off(b * 32 + l31, k0 + half * 8, K);

  // the epilogue's row (its eight threads per row own a unit each): which request it serves is asked for HERE, before the
  // operands -- after the K loop the answer has long arrived, where asking then cost the epilogue a round trip of its own
  const int t = threadIdx.x;
  const int rowl = t >> 3, u8 = t & 7, rgi = (rowl >> 5) % RG, rr = rowl & 31;
  const int r = rb * 32 * RG + rowl, u = b * 8 + u8;
  const bool epi = t < 256 * RG && r < R;
  const int src = p.ext_src[min(r, R - 1)], dst = p.ext_dst[min(r, R - 1)];

  f32x16r acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PF = TWO ? MS_RNNT_PF2 : 4;                         // k-steps of operands in flight (two-plane form: registers for eight)
  u32x4r xf[PF][TWO ? 2 : 3], wf[PF][TWO ? 2 : 3];
  constexpr int two = TWO ? 1 : 0;                                   // two fp16 planes instead of three bf16 (a template parameter: as a
                                                                     // uniform run-time flag it cost the three-plane form 10 %)
  auto load = [&](int slot, int ks) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      if (pl >= (TWO ? 2 : 3)) continue;
      xf[slot][pl] = *reinterpret_cast<const u32x4r*>(xp + (size_t)pl * p.hplane + (size_t)ks * 512);
      wf[slot][pl] = *reinterpret_cast<const u32x4r*>(w0 + (size_t)pl * wplane + (size_t)ks * 512);
    }
  };
  auto step = [&](int slot) {
    if constexpr (TWO) {     // hi.hi + lo.hi + hi.lo in fp16 (smallest terms first)
      const f16x8r wh_ = __builtin_bit_cast(f16x8r, wf[slot][0]), wl_ = __builtin_bit_cast(f16x8r, wf[slot][1]);
      const f16x8r xh_ = __builtin_bit_cast(f16x8r, xf[slot][0]), xl_ = __builtin_bit_cast(f16x8r, xf[slot][1]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl_, wh_, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh_, wl_, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh_, wh_, acc, 0, 0, 0);
    } else {
    const bf16x8r bh = __builtin_bit_cast(bf16x8r, wf[slot][0]), bm = __builtin_bit_cast(bf16x8r, wf[slot][1]);
    const bf16x8r bl = __builtin_bit_cast(bf16x8r, wf[slot][2]);
    const bf16x8r xh = __builtin_bit_cast(bf16x8r, xf[slot][0]), xm = __builtin_bit_cast(bf16x8r, xf[slot][1]);
    const bf16x8r xl = __builtin_bit_cast(bf16x8r, xf[slot][2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc, 0, 0, 0);   // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc, 0, 0, 0);
    }
  };
  const int nks = kslice / 16;
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < nks) load(j, j);
  for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (ks + j < nks) {
        step(j);
        if (ks + j + PF < nks) load(j, ks + j + PF);
      }
  }
  // the cell's other inputs (bias, G of the source state, its c) are on their way while the partial sums meet in LDS
  const bool serve = epi && dst >= 0;                                 // a request in this row
  float bias_v[4] = {0.f, 0.f, 0.f, 0.f}, g_v[4] = {0.f, 0.f, 0.f, 0.f}, c_old = 0.f;
  if (CELL && serve) {
    const float* bias = p.bias + (size_t)l * 4 * H + b * 32 + u8;
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) bias_v[gate] = bias[gate * 8];
    if (src >= 0) {
      const float* g = p.G + ((size_t)src * L + l) * 4 * H + b * 32 + u8;
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) g_v[gate] = g[gate * 8];
      c_old = p.st_c[((size_t)src * L + l) * H + u];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
  __syncthreads();
  if (!serve) return;
  float sum[4];
#pragma unroll
  for (int gate = 0; gate < 4; ++gate) {
    float v = red[rgi][rr][gate * 8 + u8];                            // slice 0 (wave index = kq RG + rg)
#pragma unroll
    for (int q = 1; q < KQ; ++q) v += red[q * RG + rgi][rr][gate * 8 + u8];
    sum[gate] = v;
  }
  if (!CELL) {
    float* g = p.G + ((size_t)dst * L + l) * 4 * H + b * 32 + u8;
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) g[gate * 8] = sum[gate];
    return;
  }
  float pre[4];
#pragma unroll
  for (int gate = 0; gate < 4; ++gate) pre[gate] = (sum[gate] + bias_v[gate]) + g_v[gate];
  float h_new, c_new;
  lstm_cell(pre, c_old, h_new, c_new);
  p.st_h[((size_t)dst * L + l) * H + u] = h_new;
  p.st_c[((size_t)dst * L + l) * H + u] = c_new;
  store_split(p.hpl + (size_t)l * 3 * p.hplane, p.hplane, frag_off(r, u, H), h_new, p.two);
}

// Round 6: layer l's cell GEMM (W_ih(l) . h'_{l-1}) and G of layer l - 1 (W_hh(l-1) . h'_{l-1}) contract the SAME rows -- the
// new h of layer l - 1 -- so they are one launch, blockIdx.z = 0 / 1: 2 x H / 8 workgroups fill the chip where the cell GEMM
// alone used half of it, and the joint's launch carries only the top layer's G (27.7 -> ~19 us).  Same bodies, same sums.
template <int RG, bool TWO>
__global__ __launch_bounds__(512) void pred_gemm3_cell_g_kernel(Beam2P p, const unsigned short* __restrict__ wih, const unsigned short* __restrict__ whh,
                                                                 int l_cell) {
  __shared__ float red[8][32][33];
  if (blockIdx.z == 0) pred_gemm3_body<true, RG, TWO>(p, wih, l_cell, blockIdx.x, blockIdx.y, red);
  else pred_gemm3_body<false, RG, TWO>(p, whh, l_cell - 1, blockIdx.x, blockIdx.y, red);
}

template <bool CELL, int RG, bool TWO>
__global__ __launch_bounds__(512) void pred_gemm3_kernel(Beam2P p, const unsigned short* __restrict__ wp_base, int l_cell) {
  __shared__ float red[8][32][33];
  pred_gemm3_body<CELL, RG, TWO>(p, wp_base, CELL ? l_cell : (int)blockIdx.z, blockIdx.x, blockIdx.y, red);
}

// pred_proj + joint: grid (J / 32, cdiv(R, 32 RG)), 512 threads (RG row groups of 32 x 8 / RG K slices).  pp[slot][j] = W_pred[j] . h_top of a row whose state is new
// (`gemm` != 0 and ext_dst[r] >= 0: committed to the pool), else the pool's; z = tanh(enc_p[t, i, j] + pp); the output
// layer's partial sums over this workgroup's 32 joint units go to plog[slice][r][v].
template <int RG, bool TWO>
__device__ __forceinline__ void beam2_joint_body(const Beam2P& p, const unsigned short* __restrict__ wpred, const float* __restrict__ enc_p,
                                                 const float* __restrict__ w_out, float* __restrict__ plog, int t, int gemm, int js, int rb,
                                                 float (*red)[32][33], float (*zs)[33], float* wo) {
  constexpr int KQ = 8 / RG, ROWS = 32 * RG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int rg = wave % RG, kq = wave / RG;
  const int H = p.H, J = p.J, V1 = p.V + 1, R = p.R, w = p.w, K = H;
  // What the z phase needs from memory -- is the row live, its slot, is its state new, the encoder's term -- is asked for HERE,
  // before the operands: asked after the GEMM it was a chain of three round trips (length and count -> slot -> pp) behind the
  // barrier.  A thread's ZI (row, unit) pairs are those of its z loop below.
  constexpr int ZI = ROWS * 32 / 512;
  bool z_live[ZI];
  int z_slot[ZI], z_dst[ZI];
  float z_enc[ZI];
#pragma unroll
  for (int k = 0; k < ZI; ++k) {
    const int idx = threadIdx.x + 512 * k, rowl = idx >> 5, c = idx & 31;
    const int r = rb * ROWS + rowl, rc = min(r, R - 1);
    const int i = rc / w, j = rc - i * w;
    const int len_i = p.lens[i], cnt_i = p.A_cnt[i];
    z_slot[k] = p.A_slot[rc];
    z_dst[k] = p.ext_dst[rc];
    z_enc[k] = enc_p[((size_t)t * p.N + i) * J + js * 32 + c];
    z_live[k] = r < R && t < len_i && j < cnt_i;
  }
  for (int idx = threadIdx.x; idx < V1 * 32; idx += 512) {
    const int v = idx >> 5, c = idx & 31;
    wo[idx] = w_out[(size_t)v * J + js * 32 + c];
  }
  if (gemm) {
    const size_t wplane = (size_t)J * K;
    const unsigned short* x = p.hpl + (size_t)(p.L - 1) * 3 * p.hplane;
    const int m = rb * ROWS + rg * 32 + l31;
    const int kslice = K / KQ, k0 = kq * kslice;
    const unsigned short* xp = x + frag_off(m, k0 + half * 8, K);
    const unsigned short* w0 = wpred + frag_off(js * 32 + l31, k0 + half * 8, K);
    f32x16r acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int PF = TWO ? MS_RNNT_PF2 : 4;
    u32x4r xf[PF][TWO ? 2 : 3], wf[PF][TWO ? 2 : 3];
    auto load = [&](int slot, int ks) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        if (pl >= (TWO ? 2 : 3)) continue;
        xf[slot][pl] = *reinterpret_cast<const u32x4r*>(xp + (size_t)pl * p.hplane + (size_t)ks * 512);
        wf[slot][pl] = *reinterpret_cast<const u32x4r*>(w0 + (size_t)pl * wplane + (size_t)ks * 512);
      }
    };
    auto step = [&](int slot) {
      if constexpr (TWO) {     // hi.hi + lo.hi + hi.lo in fp16 (smallest terms first)
        const f16x8r wh_ = __builtin_bit_cast(f16x8r, wf[slot][0]), wl_ = __builtin_bit_cast(f16x8r, wf[slot][1]);
        const f16x8r xh_ = __builtin_bit_cast(f16x8r, xf[slot][0]), xl_ = __builtin_bit_cast(f16x8r, xf[slot][1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl_, wh_, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh_, wl_, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh_, wh_, acc, 0, 0, 0);
      } else {
      const bf16x8r bh = __builtin_bit_cast(bf16x8r, wf[slot][0]), bm = __builtin_bit_cast(bf16x8r, wf[slot][1]);
      const bf16x8r bl = __builtin_bit_cast(bf16x8r, wf[slot][2]);
      const bf16x8r xh = __builtin_bit_cast(bf16x8r, xf[slot][0]), xm = __builtin_bit_cast(bf16x8r, xf[slot][1]);
      const bf16x8r xl = __builtin_bit_cast(bf16x8r, xf[slot][2]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc, 0, 0, 0);
      }
    };
    const int nks = kslice / 16;
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (j < nks) load(j, j);
    for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j)
        if (ks + j < nks) {
          step(j);
          if (ks + j + PF < nks) load(j, ks + j + PF);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
  }
  __syncthreads();
  // z of this workgroup's 64 rows x 32 joint units
#pragma unroll
  for (int k = 0; k < ZI; ++k) {
    const int idx = threadIdx.x + 512 * k;
    const int rowl = idx >> 5, c = idx & 31, rgi = rowl >> 5, rr = rowl & 31;
    float z = 0.f;
    if (z_live[k]) {
      float* ppv = p.pp + (size_t)z_slot[k] * J + js * 32 + c;
      float pv;
      if (gemm && z_dst[k] >= 0) {
        pv = red[rgi][rr][c];
#pragma unroll
        for (int q = 1; q < KQ; ++q) pv += red[q * RG + rgi][rr][c];
        *ppv = pv;
      } else {
        pv = *ppv;
      }
      z = tanhf(z_enc[k] + pv);
    }
    zs[rowl][c] = z;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < ROWS * V1; idx += 512) {
    const int rowl = idx / V1, v = idx - rowl * V1;
    const int r = rb * ROWS + rowl;
    if (r >= R) continue;
    float a = 0.f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) a = fmaf(wo[v * 32 + c], zs[rowl][c], a);
    plog[((size_t)js * R + r) * V1 + v] = a;
  }
}

template <int RG, bool TWO>
__global__ __launch_bounds__(512) void beam2_joint_kernel(Beam2P p, const unsigned short* __restrict__ wpred, const float* __restrict__ enc_p,
                                                          const float* __restrict__ w_out, float* __restrict__ plog, int t, int gemm) {
  __shared__ float red[8][32][33];
  __shared__ float zs[32 * RG][33];
  extern __shared__ __attribute__((aligned(16))) float wo[];           // [V1][32] this slice's columns of W_out
  beam2_joint_body<RG, TWO>(p, wpred, enc_p, w_out, plog, t, gemm, blockIdx.x, blockIdx.y, red, zs, wo);
}

// The joint of a predictor step and G of the step's new states in ONE launch: G is not on the step's chain (a state's G is first
// read a round later), the joint is and uses a quarter of the CUs -- so the G workgroups (block ids behind the joint's) fill
// the rest of the chip beside it instead of a launch of their own (19 us per step).  A second stream was measured for this: the
// event hand-overs between two streams cost 7 .. 25 us of idle each on this chip (profiles/r05k_*), more than G itself.
template <int RGJ, int RGG, bool TWO>
__global__ __launch_bounds__(512) void beam2_joint_g_kernel(Beam2P p, const unsigned short* __restrict__ wpred, const float* __restrict__ enc_p,
                                                            const float* __restrict__ w_out, float* __restrict__ plog, int t,
                                                            const unsigned short* __restrict__ whh, int jx, int jy, int gx, int gy,
                                                            int g_layer0) {
  __shared__ float red[8][32][33];
  __shared__ float zs[32 * RGJ][33];
  extern __shared__ __attribute__((aligned(16))) float wo[];
  int id = blockIdx.x;
  if (id < jx * jy) {
    beam2_joint_body<RGJ, TWO>(p, wpred, enc_p, w_out, plog, t, 1, id % jx, id / jx, red, zs, wo);
    return;
  }
  id -= jx * jy;
  const int b = id % gx, rb = (id / gx) % gy, l = g_layer0 + id / (gx * gy);      // (the lower layers' G rode with the cell GEMMs)
  pred_gemm3_body<false, RGG, TWO>(p, whh, l, b, rb, red);
}

// The logit of candidate c (row c / V1 of utterance i's w rows, label c % V1): b_out[v] + the J / 32 slices' partial sums, loaded
// in batches of 16 independent loads (a load at a time waited for an L2 round trip per slice: 7 us of the round kernel) and
// added in slice order.  Needs nothing but the launch arguments, so a kernel calls it for c = tid BEFORE it waits for its first
// loaded word (the live count, the length): the logits' round trip then runs beside that one instead of after it.  Rows past
// the live count hold stale sums; they are loaded and never used.
__device__ __forceinline__ float beam2_logit(const Beam2P& p, int i, int c) {
  const int V1 = p.V + 1;
  const int j = c / V1, v = c - j * V1;
  const float* src = p.plog + ((size_t)i * p.w + j) * V1 + v;
  const size_t stride = (size_t)p.R * V1;
  float a = 0.f;
  for (int s0 = 0; s0 < p.NS; s0 += 16) {
    float part[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) part[m] = s0 + m < p.NS ? src[(size_t)(s0 + m) * stride] : 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m)
      if (s0 + m < p.NS) a = (s0 + m == 0) ? part[m] : a + part[m];
  }
  return a + (p.b_out ? p.b_out[v] : 0.f);
}

// cand[j * V1 + v] = log_softmax_v(logit of (j, v)) for the rows j < rows of utterance i; `cand` in LDS, all 256 threads, ends
// with a barrier.  `first` = beam2_logit(p, i, tid), loaded by the caller at its top (any value when tid >= w * V1).
__device__ __forceinline__ void beam2_logp_rows(const Beam2P& p, float* cand, int i, int rows, int tid, float first) {
  const int V1 = p.V + 1, lane = tid & 63, wave = tid >> 6;
  if (tid < rows * V1) cand[tid] = first;
  for (int c = tid + 256; c < rows * V1; c += 256) cand[c] = beam2_logit(p, i, c);
  __syncthreads();
  for (int j = wave; j < rows; j += 4) {
    float m = -INFINITY;
    for (int v = lane; v < V1; v += 64) m = fmaxf(m, cand[j * V1 + v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int v = lane; v < V1; v += 64) sum += expf(cand[j * V1 + v] - m);
    sum = wsum(sum);
    const float lz = logf(sum) + m;
    for (int v = lane; v < V1; v += 64) cand[j * V1 + v] -= lz;
  }
  __syncthreads();
}

// blank transitions of one round into the LDS copy of B (beam_round_kernel's merge); returns the new count through wi[0].
// All 256 threads; b_* hold B (bc0 entries), old_* the live hypotheses (cnt), cand their log-probabilities.
__device__ __forceinline__ void beam2_blank_merge(int cnt, int bc0, int bcap, int V1, int blank, const float* cand, const int* old_node,
                                                  const int* old_slot, const float* old_score, int* b_node, int* b_slot,
                                                  float* b_score, int* match_at, int* wi, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  if (tid < 32) match_at[tid] = -1;
  __syncthreads();
  for (int idx = tid; idx < cnt * bc0; idx += 256) {
    const int j = idx / bc0, b = idx - j * bc0;
    if (b_node[b] == old_node[j]) match_at[j] = b;
  }
  __syncthreads();
  if (wave == 0) {
    const int j = lane;
    bool dup = false;
    if (j < cnt)
      for (int k = 0; k < j; ++k) dup |= old_node[k] == old_node[j];
    if (__any(dup)) {
      if (lane == 0) {
        int bc = bc0;
        for (int jj = 0; jj < cnt; ++jj) {
          const float s = old_score[jj] + cand[jj * V1 + blank];
          int at = -1;
          for (int b = 0; b < bc; ++b)
            if (b_node[b] == old_node[jj]) { at = b; break; }
          if (at >= 0) {
            b_score[at] = logaddexp32(b_score[at], s);
          } else if (bc < bcap) {
            b_node[bc] = old_node[jj];
            b_score[bc] = s;
            b_slot[bc] = old_slot[jj];
            ++bc;
          }
        }
        wi[0] = bc;
      }
    } else {
      const bool have = j < cnt;
      const int at = have ? match_at[j] : 0;
      const bool unmatched = have && at < 0;
      const unsigned long long um = __ballot(unmatched);
      const int pos = bc0 + __popcll(um & ((1ull << lane) - 1ull));
      if (have) {
        const float s = old_score[j] + cand[j * V1 + blank];
        if (at >= 0) {
          b_score[at] = logaddexp32(b_score[at], s);
        } else if (pos < bcap) {
          b_node[pos] = old_node[j];
          b_score[pos] = s;
          b_slot[pos] = old_slot[j];
        }
      }
      if (lane == 0) wi[0] = min(bc0 + (int)__popcll(um), bcap);
    }
  }
  __syncthreads();
}

// One emission round that is not the frame's last: beam_round_kernel with the log-probabilities made from the joint kernel's
// partial sums.
__global__ __launch_bounds__(256) void beam2_round_kernel(Beam2P p, int t, int region, int first) {
  extern __shared__ __attribute__((aligned(16))) float cand[];  // [w * V1] log-probabilities, then candidate scores
  __shared__ int wi[4];
  __shared__ int pick_idx[32];
  __shared__ float pick_val[32];
  __shared__ int old_node[32], old_slot[32];
  __shared__ float old_score[32];
  __shared__ int b_node[128], b_slot[128];
  __shared__ float b_score[128];
  __shared__ int n_pick, next_node;
  __shared__ int match_at[32];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = p.w, V1 = p.V + 1, blank = p.V;
  // every load of the kernel's top is issued before the first loaded word is used: the sets' entries into registers, then the
  // logits, and only then the LDS copies (which wait)
  const int len_i = p.lens[i], cnt = p.A_cnt[i], bc0 = first ? 0 : p.B_cnt[i], nodes0 = p.node_cnt[i];
  const int ia = i * w + min(tid, w - 1);
  const size_t ib = (size_t)i * p.bcap + min(tid, p.bcap - 1);
  const int a_node = p.A_node[ia], a_slot = p.A_slot[ia], bb_node = p.B_node[ib], bb_slot = p.B_slot[ib];
  const float a_score = p.A_score[ia], bb_score = p.B_score[ib];
  const float logit0 = beam2_logit(p, i, min(tid, w * V1 - 1));
  if (tid < w) {
    old_node[tid] = a_node;
    old_slot[tid] = a_slot;
    old_score[tid] = a_score;
    p.ext_dst[i * w + tid] = -1;  // requests default to "none" (also for utterances that have ended)
  }
  if (tid < p.bcap) {
    b_node[tid] = bb_node;
    b_score[tid] = bb_score;
    b_slot[tid] = bb_slot;
  }
  if (t >= len_i) return;
  if (tid == 0) { n_pick = 0; next_node = nodes0; }
  beam2_logp_rows(p, cand, i, cnt, tid, logit0);
  beam2_blank_merge(cnt, bc0, p.bcap, V1, blank, cand, old_node, old_slot, old_score, b_node, b_slot, b_score, match_at, wi, tid);
  {
    const int bc = wi[0];
    if (tid == 0) p.B_cnt[i] = bc;
    if (tid < bc) {
      p.B_node[(size_t)i * p.bcap + tid] = b_node[tid];
      p.B_score[(size_t)i * p.bcap + tid] = b_score[tid];
      p.B_slot[(size_t)i * p.bcap + tid] = b_slot[tid];
    }
  }
  __syncthreads();
  // label extensions: the w best of cnt * V1 candidates (score desc, ties -> lowest flat index), blank and
  // non-finite scores excluded
  const int C = cnt * V1;
  for (int c = tid; c < C; c += 256) {
    const int j = c / V1, k = c - j * V1;
    const float s = old_score[j] + cand[c];
    cand[c] = (k == blank || !isfinite(s)) ? -INFINITY : s;
  }
  __syncthreads();
  if (wave == 0) {
    int npk = 0;
    for (int q = 0; q < w; ++q) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < C; c += 64) {
        const float v = cand[c];
        if (v > bv) { bv = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (bi == 0x7fffffff || !(bv > -INFINITY)) break;      // wave-uniform: nothing finite is left
      if (lane == 0) {
        pick_idx[npk] = bi;
        pick_val[npk] = bv;
        cand[bi] = -INFINITY;
      }
      ++npk;
    }
    if (lane == 0) n_pick = npk;
  }
  __syncthreads();
  const int np = n_pick;
  if (tid < np) {
    const int q = tid;
    const int hi = pick_idx[q] / V1, k = pick_idx[q] - hi * V1;
    const int parent = old_node[hi];
    int32_t* ch = p.child + ((size_t)i * p.maxn + parent) * p.V + k;
    int node = *ch;
    if (node < 0) {
      node = atomicAdd(&next_node, 1);
      if (node < p.maxn) {
        p.node_parent[(size_t)i * p.maxn + node] = parent;
        p.node_label[(size_t)i * p.maxn + node] = k;
        *ch = node;
      } else {
        node = parent;  // cannot happen: maxn bounds every possible insertion
      }
    }
    const int slot = region * p.R + i * w + q;
    p.A_node[i * w + q] = node;
    p.A_score[i * w + q] = pick_val[q];
    p.A_slot[i * w + q] = slot;
    p.ext_label[i * w + q] = k;
    p.ext_src[i * w + q] = old_slot[hi];
    p.ext_dst[i * w + q] = slot;
  }
  __syncthreads();
  if (tid == 0) {
    p.A_cnt[i] = np;
    p.node_cnt[i] = min(next_node, p.maxn);
  }
}

// The frame's last round (blank transitions only) and the frame end in one kernel, grid (w, N): every workgroup repeats the
// (cheap, LDS) merge and selection and moves ONE survivor's state -- h, c, G of every layer and the projected predictor
// output -- into frame-start region (t + 1) & 1.
__global__ __launch_bounds__(256) void beam2_frame_end_kernel(Beam2P p, int t, int first) {
  extern __shared__ __attribute__((aligned(16))) float cand[];
  __shared__ int wi[4];
  __shared__ int old_node[32], old_slot[32];
  __shared__ float old_score[32];
  __shared__ int b_node[128], b_slot[128];
  __shared__ float b_score[128];
  __shared__ int match_at[32];
  __shared__ int sel_src[32], sel_node[32];
  __shared__ float sel_score[32];
  __shared__ int n_sel;
  const int q = blockIdx.x, i = blockIdx.y, tid = threadIdx.x;
  const int w = p.w, V1 = p.V + 1, blank = p.V;
  // (the loads of the top all issued before the first use, as in the round kernel)
  const int len_i = p.lens[i], cnt = p.A_cnt[i], bc0 = first ? 0 : p.B_cnt[i];
  const int ia = i * w + min(tid, w - 1);
  const size_t ib = (size_t)i * p.bcap + min(tid, p.bcap - 1);
  const int a_node = p.A_node[ia], a_slot = p.A_slot[ia], bb_node = p.B_node[ib], bb_slot = p.B_slot[ib];
  const float a_score = p.A_score[ia], bb_score = p.B_score[ib];
  const float logit0 = beam2_logit(p, i, min(tid, w * V1 - 1));
  if (tid < w) {
    old_node[tid] = a_node;
    old_slot[tid] = a_slot;
    old_score[tid] = a_score;
  }
  if (tid < p.bcap) {
    b_node[tid] = bb_node;
    b_score[tid] = bb_score;
    b_slot[tid] = bb_slot;
  }
  if (t >= len_i) {      // the utterance has ended: its final set moves along unchanged, so that it ends in the last frame's buffer
    if (q == 0) {
      if (tid == 0) p.nA_cnt[i] = cnt;
      if (tid < w) {     // (a thread copies the entry it has just read itself)
        p.nA_node[i * w + tid] = old_node[tid]; p.nA_slot[i * w + tid] = old_slot[tid]; p.nA_score[i * w + tid] = old_score[tid];
      }
    }
    return;
  }
  beam2_logp_rows(p, cand, i, cnt, tid, logit0);
  beam2_blank_merge(cnt, bc0, p.bcap, V1, blank, cand, old_node, old_slot, old_score, b_node, b_slot, b_score, match_at, wi, tid);
  const int bc = wi[0];
  // the beam_width best of B, one after the other (first maximum = earliest arrival among equals), by ONE wave
  if (tid < 64) {
    const int lane = tid;
    const float s0 = lane < bc ? b_score[lane] : 0.f, s1 = lane + 64 < bc ? b_score[lane + 64] : 0.f;
    bool t0 = !(lane < bc), t1 = !(lane + 64 < bc);          // "taken" also stands for "does not exist"
    int ns = 0;
    for (int x = 0; x < w && x < bc; ++x) {
      bool have = false;
      float bv = 0.f;
      int bi = 0x7fffffff;
      if (!t0) { have = true; bv = s0; bi = lane; }
      if (!t1 && (!have || s1 > bv)) { have = true; bv = s1; bi = lane + 64; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const bool oh = __shfl_xor((int)have, o, 64) != 0;
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool take = oh && (!have || ov > bv || (!(bv > ov) && oi < bi));
        if (take) { have = true; bv = ov; bi = oi; }
      }
      if (!have) break;                                       // wave-uniform
      if (bi == lane) t0 = true;
      if (bi == lane + 64) t1 = true;
      if (lane == 0) {
        sel_src[ns] = b_slot[bi];
        sel_node[ns] = b_node[bi];
        sel_score[ns] = b_score[bi];
      }
      ++ns;
    }
    if (lane == 0) n_sel = ns;
  }
  __syncthreads();
  const int ns = n_sel;
  if (q == 0 && tid == 0) p.nA_cnt[i] = ns;
  if (q >= ns) return;
  const size_t src = (size_t)sel_src[q], dst = (size_t)((t + 1) & 1) * p.R + i * w + q;
  const int LH = p.L * p.H, J = p.J;
  if (src != dst) {
    // h, c (L H each), G (4 L H), pp (J): ~50 KB at two layers of 1024.  Up to that size every load of the move is issued
    // before the first store (13 float4 per thread); larger states go array by array.
    const u32x4r* sh4 = reinterpret_cast<const u32x4r*>(p.st_h + src * LH);
    const u32x4r* sc4 = reinterpret_cast<const u32x4r*>(p.st_c + src * LH);
    const u32x4r* sg4 = reinterpret_cast<const u32x4r*>(p.G + src * 4 * LH);
    const u32x4r* sp4 = reinterpret_cast<const u32x4r*>(p.pp + src * J);
    u32x4r* dh4 = reinterpret_cast<u32x4r*>(p.st_h + dst * LH);
    u32x4r* dc4 = reinterpret_cast<u32x4r*>(p.st_c + dst * LH);
    u32x4r* dg4 = reinterpret_cast<u32x4r*>(p.G + dst * 4 * LH);
    u32x4r* dp4 = reinterpret_cast<u32x4r*>(p.pp + dst * J);
    const int nh = LH / 4, ng = LH, np4 = J / 4;
    if (LH % 4 == 0 && J % 4 == 0 && nh <= 512 && np4 <= 256) {
      // (the loads are unconditional at clamped indices: behind `if (index < n)` hipcc kept the thirteen values in scratch
      // memory and waited for every load before the next one -- the move was a chain of thirteen round trips)
      u32x4r vh[2], vc[2], vg[8], vp;
#pragma unroll
      for (int m = 0; m < 2; ++m) { vh[m] = sh4[min(tid + 256 * m, nh - 1)]; vc[m] = sc4[min(tid + 256 * m, nh - 1)]; }
#pragma unroll
      for (int m = 0; m < 8; ++m) vg[m] = sg4[min(tid + 256 * m, ng - 1)];
      vp = sp4[min(tid, np4 - 1)];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        if (tid + 256 * m < nh) { dh4[tid + 256 * m] = vh[m]; dc4[tid + 256 * m] = vc[m]; }
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (tid + 256 * m < ng) dg4[tid + 256 * m] = vg[m];
      if (tid < np4) dp4[tid] = vp;
    } else {
      for (int k = tid; k < LH; k += 256) { p.st_h[dst * LH + k] = p.st_h[src * LH + k]; p.st_c[dst * LH + k] = p.st_c[src * LH + k]; }
      for (int k = tid; k < 4 * LH; k += 256) p.G[dst * 4 * LH + k] = p.G[src * 4 * LH + k];
      for (int k = tid; k < J; k += 256) p.pp[dst * J + k] = p.pp[src * J + k];
    }
  }
  if (tid == 0) {
    p.nA_node[i * w + q] = sel_node[q];
    p.nA_score[i * w + q] = sel_score[q];
    p.nA_slot[i * w + q] = (int)dst;
  }
}

struct Net {
  const float* embedding;
  const float* w_pred;
  const float* w_out;
  const float* b_out;
  int V, D, H, L, J;
  int fused;     // one launch per predictor layer (pred_layer_fused_kernel): unit-major weight planes
};

// One prediction-network step for the R request rows described by ext_label / ext_src / ext_dst.
// The pred_proj GEMM leaves its K-slice partial sums in pp_tmp; joint_slots_kernel adds and commits them.
// the split form needs K slices of whole 16-deep k-steps and whole 64-column tiles; all layers of a call take the same path
// K slices of the split form: MS_RNNT_GATES_KSPLIT (2, 4 or 8); default 4 -- 64 column tiles x 4 slices = 256 workgroups of 8 waves,
// and half the partial sums for the cell kernel to add (measured 111 .. 114 ms per decode against 114 .. 119 with 8)
int gates_split_slices() {
  static const int v = [] {
    const char* e = getenv("MS_RNNT_GATES_KSPLIT");
    const int k = e ? atoi(e) : 4;
    return (k == 2 || k == 4 || k == 8) ? k : 4;
  }();
  return v;
}
bool gates_split_ok(const Net& n, int l) {
  static const bool off = getenv("MS_RNNT_GATES_F32") && getenv("MS_RNNT_GATES_F32")[0] == '1';
  if (off) return false;
  for (int i = 0; i < n.L; ++i) {
    const int K = (i == 0 ? n.D : n.H) + n.H;
    if (K % (16 * KSPLIT) != 0 || (4 * n.H) % 64 != 0) return false;
  }
  (void)l;
  return true;
}

int predictor_step(const Net& n, const DecLayout& W, char* ws, hipStream_t s) {
  const int R = W.R, H = n.H, L = n.L;
  int32_t* ext_label = (int32_t*)(ws + W.ext_label);
  int32_t* ext_src = (int32_t*)(ws + W.ext_src);
  int32_t* ext_dst = (int32_t*)(ws + W.ext_dst);
  float* st_h = (float*)(ws + W.st_h);
  float* st_c = (float*)(ws + W.st_c);
  float* xrow = (float*)(ws + W.xrow);
  float* gates = (float*)(ws + W.gates);
  float* htop = (float*)(ws + W.htop);
  const int frag = gates_split_ok(n, 0) ? 1 : 0;
  const size_t xplane = (size_t)((R + 31) / 32 * 32) * ((n.D > H ? n.D : H) + H);   // elements between the three x planes
  hipLaunchKernelGGL(pred_gather_kernel, dim3(R), dim3(256), 0, s, n.embedding, ext_label, ext_src, ext_dst, st_h, xrow, n.D,
                     H, L, n.V + 1, frag, xplane);
  MS_LAUNCH_CHECK();
  for (int l = 0; l < L; ++l) {
    const int K = (l == 0 ? n.D : H) + H;
    int nsplit = KSPLIT;
    if (n.fused) {
      const unsigned short* planes = (const unsigned short*)(ws + W.wcat) + 3 * W.wcat_off[l];
      const unsigned short* xin = reinterpret_cast<const unsigned short*>(ws + ((l & 1) ? W.xrow2 : W.xrow));
      unsigned short* xout = reinterpret_cast<unsigned short*>(ws + ((l & 1) ? W.xrow : W.xrow2));
      const float* bias = (const float*)(ws + W.bcat) + W.bcat_off[l];
      if (R > 32)
        hipLaunchKernelGGL(pred_layer_fused_kernel<2>, dim3(H / 8, (R + 63) / 64), dim3(512), 0, s, xin, xplane, planes, bias, ext_src,
                           ext_dst, st_h, st_c, xout, htop, R, K, H, L, l);
      else
        hipLaunchKernelGGL(pred_layer_fused_kernel<1>, dim3(H / 8, 1), dim3(512), 0, s, xin, xplane, planes, bias, ext_src, ext_dst,
                           st_h, st_c, xout, htop, R, K, H, L, l);
      MS_LAUNCH_CHECK();
      continue;
    }
    if (gates_split_ok(n, l)) {
      // three bf16 planes of [W_ih | W_hh] (pack_cat3_kernel), plane l at 3 x wcat_off[l] elements of 2 bytes
      nsplit = gates_split_slices();
      const unsigned short* planes = (const unsigned short*)(ws + W.wcat) + 3 * W.wcat_off[l];
      hipLaunchKernelGGL(pred_gates_split_kernel, dim3(4 * H / 64, nsplit, (R + 127) / 128), dim3(512), 0, s, reinterpret_cast<const unsigned short*>(xrow), xplane, planes,
                         (const float*)(ws + W.bcat) + W.bcat_off[l], gates, R, K, 4 * H, nsplit);
      MS_LAUNCH_CHECK();
    } else {
      int rc = ms::linear_splitk_launch(xrow, (const float*)(ws + W.wcat) + W.wcat_off[l],
                                        (const float*)(ws + W.bcat) + W.bcat_off[l], gates, R, K, 4 * H, KSPLIT, s);
      if (rc != MS_OK) return rc;
    }
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(R), dim3(256), 0, s, gates, ext_src, ext_dst, st_h, st_c, xrow, htop, H, L, l,
                       R, nsplit, frag, xplane);
    MS_LAUNCH_CHECK();
  }
  return ms::linear_splitk_launch(htop, n.w_pred, nullptr, (float*)(ws + W.pp_tmp), R, H, n.J, KSPLIT, s);
}

// The round-5 beam decode (see pred_gemm3_kernel): the same hypothesis lists, trie and pool as the round-4 sequence below it.
int beam2_decode(const BeamP& bp, const DecLayout& W, char* ws, const float* embedding, const float* const* w_ih,
                 const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, const float* w_pred,
                 const float* w_out, const float* b_out, const float* enc_p, int32_t* out_idx, int32_t* out_len, float* out_score,
                 int T, int N, int V, int D, int H, int L, int J, int w, int max_symbols, hipStream_t s) {
  const int R = W.R, V1 = V + 1;
  const size_t rpad = (size_t)((R + 63) / 64 * 64);
  Beam2P q{};
  {
    const char* e = getenv("MS_RNNT_PLANES");      // 3 = the exact three-bf16 operands of round 5 (A/B runs); default: two fp16 planes
    q.two = (e && e[0] == '3') ? 0 : 1;
  }
  q.lens = bp.lens;
  int32_t* a2 = (int32_t*)(ws + W.v2_A2);
  int32_t* Acnt[2] = {bp.A_cnt, a2};
  int32_t* Anode[2] = {bp.A_node, a2 + N};
  float* Ascore[2] = {bp.A_score, (float*)(a2 + N + R)};
  int32_t* Aslot[2] = {bp.A_slot, a2 + N + 2 * R};
  q.B_cnt = bp.B_cnt; q.B_node = bp.B_node; q.B_slot = bp.B_slot; q.B_score = bp.B_score;
  q.ext_label = bp.ext_label; q.ext_src = bp.ext_src; q.ext_dst = bp.ext_dst;
  q.node_cnt = bp.node_cnt; q.node_parent = bp.node_parent; q.node_label = bp.node_label; q.child = bp.child;
  q.st_h = (float*)(ws + W.st_h); q.st_c = (float*)(ws + W.st_c); q.pp = (float*)(ws + W.pp); q.G = (float*)(ws + W.v2_G);
  q.E = (const float*)(ws + W.v2_E); q.bias = (const float*)(ws + W.v2_bias); q.plog = (const float*)(ws + W.v2_plog); q.b_out = b_out;
  q.hpl = (unsigned short*)(ws + W.v2_hpl); q.hplane = rpad * H;
  q.N = N; q.w = w; q.V = V; q.bcap = bp.bcap; q.maxn = bp.maxn; q.R = R; q.H = H; q.L = L; q.J = J; q.NS = J / 32;
  auto set_frame = [&](int t) {
    const int c = t & 1, n = c ^ 1;
    q.A_cnt = Acnt[c]; q.A_node = Anode[c]; q.A_score = Ascore[c]; q.A_slot = Aslot[c];
    q.nA_cnt = Acnt[n]; q.nA_node = Anode[n]; q.nA_score = Ascore[n]; q.nA_slot = Aslot[n];
  };
  set_frame(0);
  unsigned short* whh = (unsigned short*)(ws + W.v2_whh);
  unsigned short* wih = (unsigned short*)(ws + W.v2_wih);
  unsigned short* wpred = (unsigned short*)(ws + W.v2_wpred);
  float* plog = (float*)(ws + W.v2_plog);
  // ---- once per call: the operand planes, the embedding table, the biases
  const size_t wplane3 = (size_t)3 * 4 * H * H;
  for (int l = 0; l < L; ++l) {
    hipLaunchKernelGGL(pack_rows3_kernel, dim3(4 * H), dim3(256), 0, s, w_hh[l], whh + (size_t)l * wplane3, 4 * H, H, H, q.two);
    if (l > 0) {
      hipLaunchKernelGGL(pack_rows3_kernel, dim3(4 * H), dim3(256), 0, s, w_ih[l], wih + (size_t)(l - 1) * wplane3, 4 * H, H, H, q.two);
      hipLaunchKernelGGL(unit_major_add_kernel, dim3(ms::cdiv(4 * H, 256)), dim3(256), 0, s, (const float*)nullptr, b_ih[l], b_hh[l],
                         (float*)(ws + W.v2_bias) + (size_t)l * 4 * H, H, 0);
    }
  }
  hipLaunchKernelGGL(pack_rows3_kernel, dim3(J), dim3(256), 0, s, w_pred, wpred, J, H, 0, q.two);
  MS_LAUNCH_CHECK();
  int rc = ms::linear_launch(embedding, w_ih[0], nullptr, (float*)(ws + W.v2_Etmp), V1, D, 4 * H, MS_ACT_NONE, 0.f, 0.f, s);
  if (rc != MS_OK) return rc;
  hipLaunchKernelGGL(unit_major_add_kernel, dim3(ms::cdiv(4 * H, 256)), dim3(256), 0, s, (const float*)(ws + W.v2_Etmp), b_ih[0],
                     b_hh[0], (float*)(ws + W.v2_E), H, V1);
  MS_LAUNCH_CHECK();
  // rows per workgroup: 32 (eight K slices) for the joint -- 64 workgroups instead of 32 at configs[3] --, 64 (four K slices) for
  // the gate GEMMs (measured: 11.7 against 14.1 us per layer; MS_RNNT_RG = 1 / 2 forces one form for both)
  static const int rg_env = getenv("MS_RNNT_RG") ? atoi(getenv("MS_RNNT_RG")) : 0;
  const int rgg = rg_env == 1 && H % 128 == 0 ? 1 : 2;
  const int rgj = rg_env == 2 || H % 128 != 0 ? 2 : 1;
  const dim3 ggrid(H / 8, (R + 32 * rgg - 1) / (32 * rgg));
  const dim3 jgrid(J / 32, (R + 32 * rgj - 1) / (32 * rgj));
  const size_t cand_lds = (size_t)w * V1 * 4, wo_lds = (size_t)V1 * 32 * 4;
  // (MS_RNNT_FUSE_G=0, read per call: the lower layers' G in the joint's launch as in round 5 -- A/B runs)
  const char* fg = getenv("MS_RNNT_FUSE_G");
  const bool fuse_g = !(fg && fg[0] == '0') && L > 1;
  auto predictor_step2 = [&]() {       // cell 0 of the request rows, then cells 1 .. L - 1; G rides with the step's joint launch
    hipLaunchKernelGGL(beam2_cell0_kernel, dim3(R), dim3(256), 0, s, q);
    for (int l = 1; l < L; ++l) {
      if (fuse_g) {
        const dim3 cg(ggrid.x, ggrid.y, 2);
#define MS_CG(RG_) \
        if (q.two) hipLaunchKernelGGL((pred_gemm3_cell_g_kernel<RG_, true>), cg, dim3(512), 0, s, q, (const unsigned short*)wih, (const unsigned short*)whh, l); \
        else hipLaunchKernelGGL((pred_gemm3_cell_g_kernel<RG_, false>), cg, dim3(512), 0, s, q, (const unsigned short*)wih, (const unsigned short*)whh, l)
        if (rgg == 1) { MS_CG(1); } else { MS_CG(2); }
#undef MS_CG
      } else if (rgg == 1) {
        if (q.two) hipLaunchKernelGGL((pred_gemm3_kernel<true, 1, true>), ggrid, dim3(512), 0, s, q, (const unsigned short*)wih, l);
        else hipLaunchKernelGGL((pred_gemm3_kernel<true, 1, false>), ggrid, dim3(512), 0, s, q, (const unsigned short*)wih, l);
      } else {
        if (q.two) hipLaunchKernelGGL((pred_gemm3_kernel<true, 2, true>), ggrid, dim3(512), 0, s, q, (const unsigned short*)wih, l);
        else hipLaunchKernelGGL((pred_gemm3_kernel<true, 2, false>), ggrid, dim3(512), 0, s, q, (const unsigned short*)wih, l);
      }
    }
  };
  auto joint = [&](int t, bool after_step) {
    const int g_layer0 = fuse_g ? L - 1 : 0;
    const int nj = jgrid.x * jgrid.y, ng = ggrid.x * ggrid.y * (L - g_layer0);
    if (after_step) {
      // the step's new states: their projected predictor output (joint) and their G (beside it)
#define MS_JG(RJ, RGm) do { if (q.two) MS_JG2(RJ, RGm, true); else MS_JG2(RJ, RGm, false); } while (0)
#define MS_JG2(RJ, RGm, TW) hipLaunchKernelGGL((beam2_joint_g_kernel<RJ, RGm, TW>), dim3(nj + ng), dim3(512), wo_lds, s, q, (const unsigned short*)wpred, \
                                          enc_p, w_out, plog, t, (const unsigned short*)whh, (int)jgrid.x, (int)jgrid.y, (int)ggrid.x, (int)ggrid.y, g_layer0)
      if (rgj == 1 && rgg == 1) MS_JG(1, 1);
      else if (rgj == 1) MS_JG(1, 2);
      else if (rgg == 1) MS_JG(2, 1);
      else MS_JG(2, 2);
#undef MS_JG
#undef MS_JG2
    } else if (rgj == 1) {
      if (q.two) hipLaunchKernelGGL((beam2_joint_kernel<1, true>), jgrid, dim3(512), wo_lds, s, q, (const unsigned short*)wpred, enc_p, w_out, plog, t, 0);
      else hipLaunchKernelGGL((beam2_joint_kernel<1, false>), jgrid, dim3(512), wo_lds, s, q, (const unsigned short*)wpred, enc_p, w_out, plog, t, 0);
    } else {
      if (q.two) hipLaunchKernelGGL((beam2_joint_kernel<2, true>), jgrid, dim3(512), wo_lds, s, q, (const unsigned short*)wpred, enc_p, w_out, plog, t, 0);
      else hipLaunchKernelGGL((beam2_joint_kernel<2, false>), jgrid, dim3(512), wo_lds, s, q, (const unsigned short*)wpred, enc_p, w_out, plog, t, 0);
    }
  };
  // ---- the root hypotheses' predictor state (blank on the zero state)
  predictor_step2();
  MS_LAUNCH_CHECK();
  for (int t = 0; t < T; ++t) {
    set_frame(t);
    for (int v = 0; v < max_symbols; ++v) {
      const int last = v == max_symbols - 1;
      // the rows' states are new after a predictor step (and at the very start); a frame's first round works on the survivors
      joint(t, v > 0 || t == 0);
      if (last) {
        hipLaunchKernelGGL(beam2_frame_end_kernel, dim3(w, N), dim3(256), cand_lds, s, q, t, v == 0 ? 1 : 0);
      } else {
        hipLaunchKernelGGL(beam2_round_kernel, dim3(N), dim3(256), cand_lds, s, q, t, 2 + v, v == 0 ? 1 : 0);
        predictor_step2();
      }
      MS_LAUNCH_CHECK();
    }
  }
  set_frame(T);           // the last frame end wrote buffer T & 1
  BeamP fin = bp;
  fin.A_cnt = q.A_cnt; fin.A_node = q.A_node; fin.A_score = q.A_score; fin.A_slot = q.A_slot;
  hipLaunchKernelGGL(beam_finish_kernel, dim3(N), dim3(64), 0, s, fin, out_idx, out_len, out_score,
                     T * (max_symbols > 1 ? max_symbols - 1 : 0) + 1);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

}  // namespace

extern "C" size_t ms_rnnt_decode_workspace_bytes(int T, int N, int V, int D, int H, int L, int J, int beam_width,
                                                 int max_symbols, int greedy) {
  if (T <= 0 || N <= 0 || V <= 0 || D <= 0 || H <= 0 || L <= 0 || L > 8 || J <= 0 || max_symbols <= 0) return 0;
  if (!greedy && (beam_width <= 0 || beam_width > 32)) return 0;
  return dec_layout(T, N, V, D, H, L, J, greedy ? 1 : beam_width, max_symbols, greedy).total;
}

extern "C" int ms_rnnt_decode(const float* enc_p, const int32_t* lens, const float* embedding, const float* const* w_ih,
                              const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                              const float* w_pred, const float* w_out, const float* b_out, int32_t* out_idx,
                              int32_t* out_len, float* out_score, int T, int N, int V, int D, int H, int L, int J,
                              int beam_width, int max_symbols, int greedy, void* workspace, size_t workspace_bytes,
                              void* stream) {
  MS_REQUIRE(enc_p && lens && embedding && w_ih && w_hh && b_ih && b_hh && w_pred && w_out && out_idx && out_len && workspace,
             "null pointer");
  MS_REQUIRE(T > 0 && N > 0 && V > 0 && D > 0 && H > 0 && L > 0 && L <= 8 && J > 0 && max_symbols > 0, "bad shape");
  MS_REQUIRE(greedy || (beam_width > 0 && beam_width <= 32), "beam_width must be in [1, 32]");
  const int w = greedy ? 1 : beam_width, V1 = V + 1;
  MS_REQUIRE((size_t)(J + V1) * 4 <= 64 * 1024, "joint width too large");
  MS_REQUIRE((size_t)w * V1 * 4 <= 60 * 1024, "beam_width * (V + 1) candidates exceed the LDS budget");
  MS_REQUIRE(w * max_symbols <= 128, "beam_width * max_symbols must not exceed 128");
  const DecLayout W = dec_layout(T, N, V, D, H, L, J, w, max_symbols, greedy);
  MS_REQUIRE(workspace_bytes >= W.total, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int R = W.R;
  Net net{embedding, w_pred, w_out, b_out, V, D, H, L, J, 0};
  {
    // one launch per predictor layer where the shapes allow it (K slices of whole 16-deep k-steps for eight waves, eight
    // units per workgroup); MS_RNNT_FUSED=0 (read per call: the tests' A/B switch) keeps gate GEMM + cell kernel
    const char* e = getenv("MS_RNNT_FUSED");
    bool ok = !(e && e[0] == '0') && gates_split_ok(net, 0) && H % 8 == 0;
    for (int l = 0; l < L && ok; ++l) ok = (((l == 0 ? D : H) + H) % 128) == 0;
    net.fused = ok ? 1 : 0;
  }

  for (int l = 0; l < L; ++l) {
    MS_REQUIRE(w_ih[l] && w_hh[l], "null layer weights");
    if (gates_split_ok(net, l))
      hipLaunchKernelGGL(pack_cat3_kernel, dim3(4 * H), dim3(256), 0, s, w_ih[l], w_hh[l], b_ih[l], b_hh[l],
                         (unsigned short*)(ws + W.wcat) + 3 * W.wcat_off[l], (float*)(ws + W.bcat) + W.bcat_off[l], H, l == 0 ? D : H,
                         net.fused);
    else
      hipLaunchKernelGGL(pack_cat_kernel, dim3(4 * H), dim3(256), 0, s, w_ih[l], w_hh[l], b_ih[l], b_hh[l],
                         (float*)(ws + W.wcat) + W.wcat_off[l], (float*)(ws + W.bcat) + W.bcat_off[l], H, l == 0 ? D : H);
    MS_LAUNCH_CHECK();
  }
  const size_t joint_lds = (size_t)(J + V1) * 4;
  int32_t* A_cnt = (int32_t*)(ws + W.A_cnt);
  int32_t* A_slot = (int32_t*)(ws + W.A_slot);
  float* logp = (float*)(ws + W.logp);
  float* pp = (float*)(ws + W.pp);

  if (greedy) {
    int32_t* cur_t = (int32_t*)(ws + W.live);          // current frame of every utterance
    int32_t* sym = (int32_t*)(ws + W.A_cnt);           // labels emitted on that frame so far
    int32_t* done_cnt = (int32_t*)(ws + W.B_cnt);      // utterances that have reached their length
    int32_t* out_cnt = (int32_t*)(ws + W.out_cnt);
    const int out_stride = T * max_symbols;
    hipLaunchKernelGGL(greedy_init_kernel, dim3(ms::cdiv(N, 64)), dim3(64), 0, s, (int32_t*)(ws + W.ext_label),
                       (int32_t*)(ws + W.ext_src), (int32_t*)(ws + W.ext_dst), A_slot, out_cnt, cur_t, sym, done_cnt, N, V);
    MS_LAUNCH_CHECK();
    int rc = predictor_step(net, W, ws, s);
    if (rc != MS_OK) return rc;
    const float* pp_tmp = (const float*)(ws + W.pp_tmp);
    const int32_t* ext_dst = (const int32_t*)(ws + W.ext_dst);
    // every iteration advances every unfinished utterance by one label or by a whole chunk of blanks, so this bound is
    // never reached unless the device counter cannot be read
    const long max_iters = (long)T * max_symbols + ms::cdiv(T, GREEDY_CHUNK) + 1;
    constexpr int CHECK_EVERY = 2;
    GreedyPoll* poll = greedy_poll();
    if (poll == nullptr) {
      ms::set_error("ms_rnnt_decode: pinned counter ring / events for the greedy decode's end-of-work poll could not be created");
      return MS_ERR_HIP;
    }
    int32_t* host_done = poll->host;
    hipEvent_t* ev = poll->ev;
    constexpr int RING = GreedyPoll::RING;
    for (int k = 0; k < RING; ++k) host_done[k] = 0;      // no copy of an earlier call is in flight: every call ends on its last event
    bool polled = true;
    int last_slot = -1;
    long checks = 0;
    rc = MS_OK;
    for (long it = 0; it < max_iters && rc == MS_OK; ++it) {
      hipLaunchKernelGGL(joint_slots_kernel<true>, dim3(N * GREEDY_CHUNK), dim3(256), joint_lds, s, enc_p, lens, pp, pp_tmp, ext_dst,
                         A_slot, cur_t, w_out, b_out, logp, 0, N, GREEDY_CHUNK, J, V1, R);
      hipLaunchKernelGGL(greedy_scan_kernel, dim3(N), dim3(64), 0, s, logp, lens, cur_t, sym, done_cnt, out_idx, out_cnt,
                         (int32_t*)(ws + W.ext_label), (int32_t*)(ws + W.ext_src), (int32_t*)(ws + W.ext_dst), V1, V, out_stride,
                         max_symbols);
      if (hipGetLastError() != hipSuccess) { ms::set_error("ms_rnnt_decode: launch failed"); rc = MS_ERR_HIP; break; }
      rc = predictor_step(net, W, ws, s);
      if (rc != MS_OK || !polled || (it + 1) % CHECK_EVERY != 0) continue;
      // fetch the counter behind this iteration; look at the one fetched two checks ago (already complete, or nearly):
      // the queue stays at least CHECK_EVERY iterations deep and at most 2 * CHECK_EVERY iterations run past the end
      const int slot = (int)(checks % RING);
      if (hipMemcpyAsync(&host_done[slot], done_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipEventRecord(ev[slot], s) != hipSuccess) { ms::set_error("ms_rnnt_decode: counter read-back failed"); rc = MS_ERR_HIP; break; }
      last_slot = slot;
      ++checks;
      if (checks > 2) {
        const int old = (int)((checks - 3) % RING);
        if (hipEventSynchronize(ev[old]) == hipSuccess && host_done[old] >= N) break;
      }
    }
    if (rc == MS_OK && hipMemcpyAsync(out_len, out_cnt, (size_t)N * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = MS_ERR_HIP;
    // the pinned words are targets of copies in flight: wait for the last one (the host has run at most 2 * CHECK_EVERY
    // iterations ahead, so this is short); the ring and the events stay with the thread for its next call
    if (last_slot >= 0) (void)hipEventSynchronize(ev[last_slot]);
    (void)polled;
    return rc;
  }

  BeamP p;
  p.lens = lens;
  p.logp = logp;
  p.A_cnt = A_cnt;
  p.A_node = (int32_t*)(ws + W.A_node);
  p.A_slot = A_slot;
  p.A_score = (float*)(ws + W.A_score);
  p.B_cnt = (int32_t*)(ws + W.B_cnt);
  p.B_node = (int32_t*)(ws + W.B_node);
  p.B_slot = (int32_t*)(ws + W.B_slot);
  p.B_score = (float*)(ws + W.B_score);
  p.ext_label = (int32_t*)(ws + W.ext_label);
  p.ext_src = (int32_t*)(ws + W.ext_src);
  p.ext_dst = (int32_t*)(ws + W.ext_dst);
  p.node_cnt = (int32_t*)(ws + W.node_cnt);
  p.node_parent = (int32_t*)(ws + W.node_parent);
  p.node_label = (int32_t*)(ws + W.node_label);
  p.child = (int32_t*)(ws + W.child);
  p.N = N;
  p.w = w;
  p.V = V;
  p.bcap = W.bcap;
  p.maxn = W.maxn;
  p.R = R;
  MS_HIP(hipMemsetAsync(p.child, 0xFF, (size_t)N * W.maxn * V * 4, s));
  hipLaunchKernelGGL(beam_init_kernel, dim3(ms::cdiv(N, 64)), dim3(64), 0, s, p);
  MS_LAUNCH_CHECK();
  {
    const char* e = getenv("MS_RNNT_V2");
    // LDS of the re-cut sequence's two largest kernels, each against the 64 KB a launch gets without an attribute: the joint
    // (static red[8][32][33] + zs[64][33] = 42 240 B, dynamic W_out slice V1 x 32 floats) and the round kernel (candidates
    // w x V1 floats beside ~3 KB of static arrays).  Larger vocabularies take the round-4 sequence below.
    constexpr size_t JOINT_STATIC_LDS = (8 * 32 * 33 + 64 * 33) * sizeof(float);
    if (!(e && e[0] == '0') && H % 64 == 0 && J % 32 == 0 && JOINT_STATIC_LDS + (size_t)V1 * 32 * 4 <= 64 * 1024 &&
        (size_t)w * V1 * 4 <= 48 * 1024)
      return beam2_decode(p, W, ws, embedding, w_ih, w_hh, b_ih, b_hh, w_pred, w_out, b_out, enc_p, out_idx, out_len, out_score, T,
                          N, V, D, H, L, J, w, max_symbols, s);
  }
  int rc = predictor_step(net, W, ws, s);
  if (rc != MS_OK) return rc;
  const float* pp_tmp = (const float*)(ws + W.pp_tmp);
  const size_t cand_lds = (size_t)w * V1 * 4;
  for (int t = 0; t < T; ++t) {
    for (int v = 0; v < max_symbols; ++v) {
      const int last = v == max_symbols - 1;
      hipLaunchKernelGGL(joint_slots_kernel<false>, dim3(R), dim3(256), joint_lds, s, enc_p, lens, pp, pp_tmp, p.ext_dst, A_slot, A_cnt,
                         w_out, b_out, logp, t, N, w, J, V1, R);
      MS_LAUNCH_CHECK();
      hipLaunchKernelGGL(beam_round_kernel, dim3(N), dim3(256), cand_lds, s, p, t, 2 + v, v == 0, last);
      MS_LAUNCH_CHECK();
      if (!last) {
        rc = predictor_step(net, W, ws, s);
        if (rc != MS_OK) return rc;
      }
    }
    hipLaunchKernelGGL(beam_frame_end_kernel, dim3(w, N), dim3(256), 0, s, p, (float*)(ws + W.st_h), (float*)(ws + W.st_c), pp,
                       t, L * H, J);
    MS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(beam_finish_kernel, dim3(N), dim3(64), 0, s, p, out_idx, out_len, out_score,
                     T * (max_symbols > 1 ? max_symbols - 1 : 0) + 1);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
