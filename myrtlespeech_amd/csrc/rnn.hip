// Recurrent layers of the acoustic encoder (model/rnn.py:170-183 -> torch.nn.LSTM/GRU/RNN,
// model/hard_lstm.py:346-379, 416-456, 513-561).
//
// A layer is two launches' worth of work:
//   (i)  input projection  xproj[t,n,:] = x[t,n,:] . W_ih^T (+ biases) for every frame at
//        once -- one big f32-MFMA GEMM (gemm.hip);
//   (ii) the recurrence over time.
//
// Fast path (LSTM / hard LSTM, H % 32 == 0, H <= 1024): ONE persistent launch per layer,
// both directions concurrently.  Direction d is served by J = H/8 workgroups (one per CU);
// workgroup j keeps the 32 recurrent-weight rows of hidden units [8j, 8j+8) (i,f,g,o) in LDS
// for the whole sequence (128 B x H: 128 KiB at H = 1024), so W_hh never leaves the chip
// after the first read.  Per time step every workgroup needs the whole h_{t-1}: it is
// exchanged through an L2/MALL-resident buffer hx[parity][k/4][n][4] with write-through
// (sc1) stores, one epoch flag per producer, relaxed sc1 polls and sc1 loads (the
// placement-independent hand-off of cdna_hip_programming.md G16 / MI355X_MICROARCH.md
// "Valid forms": every payload store sc1 + drained before the flag, every payload load
// sc1, the polling wave is the loading wave).  The 32(batch) x 32(gate rows) x H product
// is split over the 4 waves along K (each wave polls only the producers of its K-quarter),
// reduced through LDS, and the cell update keeps c in registers.  pack_padded_sequence
// semantics are a per-(t,n) predicate: inactive frames keep (h,c) frozen and output 0, so
// the reverse direction starts at each sequence's own last frame.
//
// Generic path (GRU, tanh RNN, odd sizes): one launch per time step, one wave per
// (hidden unit, direction), K split across lanes.  Correct for every shape; not tuned.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace ms {
int split_planes_launch(const float* x, unsigned short* hi, unsigned short* lo, size_t elems, int prec, hipStream_t stream);
bool gemm_rows_from_device_ok(int M, int K, int N);
int gemm_bf16x3_launch_rows(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                            const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                            float hi, int prec, hipStream_t stream, const int* m_eff);
int gemm_bf16x3_launch(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                       const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                       float hi, int prec, hipStream_t stream);
int gemm_bf16x3_launch_ld(const unsigned short* ah, const unsigned short* al, const unsigned short* wh,
                          const unsigned short* wl, const float* bias, float* y, int M, int K, int N, int act, float lo,
                          float hi, int prec, hipStream_t stream, const int* m_eff, int lda, int ldw, int kmode);
bool gemm_k_halves_ok(int M, int N);
int linear_launch(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, float lo,
                  float hi, hipStream_t stream);
}

namespace {

using ms::f32x16;
using ms::f32x4;

constexpr int RED_STRIDE = 40;                    // floats per reduction row (conflict-free, see cell read)
constexpr int RED_FLOATS = 4 * 32 * RED_STRIDE;   // 4 waves x 32 batch rows
constexpr size_t STATUS_BYTES = 256;
constexpr int HX_REGIONS = 8;                      // exchange regions in a workspace (flags bits 8..11 select one)
constexpr unsigned long long SPIN_LIMIT_TICKS = 200000000ull;  // 2 s of the 100 MHz wall clock

inline int gates_of(int cell) { return (cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM) ? 4 : (cell == MS_CELL_GRU ? 3 : 1); }

bool force_generic() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_RNN_FORCE_GENERIC");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

// Every workgroup of a persistent launch spins on its peers, so the whole grid has to be resident at once.  The
// decision is taken here, once per device and identically at pack time and at launch time: the occupancy API's answer for
// the persistent kernels (defined below them) times the CU count must cover the grid, otherwise the layer is packed for
// and run by the per-step kernels, which need no co-residency.
int persistent_blocks_per_cu(bool gru);

// LSTM widths beyond 1024 (round 4): the two-stream kernel with a wave's K-quarter of 32 gate rows x H in 16 H / 128 VGPRs per
// lane (256 at H = 2048) -- bf16x3 operands only, H / 8 workgroups per direction, the directions of a bidirectional layer in
// two launches when they do not fit the CUs together.  They had one launch per step: 15 .. 21 ms per layer at [501, 32, H].
bool wide_lstm_h(int H) { return H == 1280 || H == 1536 || H == 2048; }
bool use_fast(int cell, int H, int ndir) {
  if (force_generic()) return false;
  if (!(cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM)) return false;
  const int cus = ms::num_cus();
  if (H > 1024) {
    // (round 6: HardLSTM too -- the reference's ONNX-exportable cell, hard_lstm.py; a DeepSpeech1 of the paper's width, 2 048, used it)
    if (!wide_lstm_h(H) || !(ms::precision_mode() == ms::PREC_BF16X3 || ms::precision_mode() == ms::PREC_F16X3)) return false;
    return cus > 0 && (H / 8) <= cus * std::min(1, persistent_blocks_per_cu(false));
  }
  if (H % 32 != 0) return false;
  return cus > 0 && ndir * (H / 8) <= cus * std::min(1, persistent_blocks_per_cu(false));
}

// Operand precision of the persistent recurrence: "f32" = exact float32 MFMA,
// "bf16x3" (default) = every f32 operand split into bf16 hi + lo, products hi*hi + lo*hi +
// hi*lo accumulated in f32 (relative error ~2^-17 per product instead of 2^-24).
bool want_split() { return ms::precision_mode() != ms::PREC_F32; }
bool use_split(int cell, int H, int ndir) { return use_fast(cell, H, ndir) && want_split() && H % 64 == 0; }
bool two_stream_shape(int H) { return H == 256 || H == 512 || H == 768 || H == 1024 || wide_lstm_h(H); }
// MS_PRECISION=f32 on the two-stream shapes: float32-MFMA two-stream kernel with register-resident W_hh; h crosses
// workgroups with its mantissa LSB used as the epoch tag (MS_LSTM_F32_ONE_STREAM=1 keeps the one-stream LDS-weights
// kernel, whose exchange is bit-exact)
bool use_f32x2(int cell, int H, int ndir) {
  static const bool off = getenv("MS_LSTM_F32_ONE_STREAM") && getenv("MS_LSTM_F32_ONE_STREAM")[0] == '1';
  return !off && use_fast(cell, H, ndir) && !want_split() && two_stream_shape(H);
}
// MS_PRECISION=fp16: single-pass fp16 operands, only on the two-stream kernel's shapes (elsewhere bf16x3)
bool use_f16(int cell, int H, int ndir) {
  return ms::precision_mode() == ms::PREC_F16 && use_split(cell, H, ndir) && two_stream_shape(H);
}
// Plane format of a layer's split operands: one fp16 plane where MS_PRECISION=fp16 has a kernel for the shape; otherwise the
// two-plane format of the mode -- fp16 pairs in the default f16x3 mode, bf16 pairs in bf16x3 mode (and, as before round 6, for
// the layers MS_PRECISION=fp16 has no one-plane kernel for).
int two_plane_mode() { return ms::precision_mode() == ms::PREC_F16X3 ? ms::PREC_F16X3 : ms::PREC_BF16X3; }
int layer_prec(int cell, int H, int ndir) { return use_f16(cell, H, ndir) ? ms::PREC_F16 : two_plane_mode(); }
// the input projection runs as the bf16x3 GEMM whenever the operands are split (every cell, also the streamed-weights
// path) and In allows 16-byte granules
bool use_split_gemm(int cell, int H, int ndir, int In) {
  if (In % 32 != 0 || !want_split() || force_generic()) return false;
  return use_split(cell, H, ndir) || !use_fast(cell, H, ndir);
}

// GRU whose recurrent weights fit the register files when cut into 10-unit workgroups (one per CU): H in {1280, 2560}
// -- the reference's shipped DS2 config is 3 x GRU-2560, unidirectional.  MS_GRU_PERSISTENT=0 falls back to the per-step
// streamed-weights kernel.
// Hidden sizes of the persistent GRU and their tiling: U units per workgroup (3 U gate rows <= 32), KS = H / 128 k-steps per
// wave.  2560 and 1280 (the reference's shipped width and half of it) since round 1; the other multiples of 128 (round 4)
// because a GRU of any other width fell to one launch per step: GRU-1024 bidirectional 10.9 ms against 3.1 for the LSTM.
constexpr int gru_units(int H) {
  return (H == 2560 || H == 1280) ? 10 : (H == 2048 || H == 1536 || H == 1024 || H == 768 || H == 512) ? 8 : 0;
}
bool use_gru_persistent(int cell, int H, int ndir) {
  static const bool off = getenv("MS_GRU_PERSISTENT") && getenv("MS_GRU_PERSISTENT")[0] == '0';
  if (off || force_generic() || !want_split() || cell != MS_CELL_GRU) return false;
  const int U = gru_units(H);
  if (U == 0) return false;
  const int cus = ms::num_cus();
  // (round 6) ONE direction's workgroups must fit the CUs; when both do not, the directions run as two launches (GruP::d_base),
  // as the two-stream LSTM's do -- a bidirectional GRU-1536 / -2048 had fallen to one launch per step (14.0 / 17.7 ms per layer)
  (void)ndir;
  return cus > 0 && (H / U) <= cus * std::min(1, persistent_blocks_per_cu(true));
}

// MS_LSTM_RING=<slots> (power of two, 2..128; default 2): exchange slots per (stream, plane) of the two-stream LSTM kernel
// and of the persistent GRU.  Same-box A/B runs show no difference between 2 and 8..128 slots beyond the +-2 % run-to-run
// noise, so the default stays at two; the switch remains for experiments.
int lstm_ring_shift() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_LSTM_RING");
    int slots = e ? atoi(e) : 2;
    v = 1;
    while ((1 << v) < slots && v < 7) ++v;
  }
  return v;
}
struct PackLayout {
  size_t wih, bias_x, whh, bhh, total;  // byte offsets
};
PackLayout pack_layout(int cell, int In, int H, int ndir) {
  const size_t GH = (size_t)gates_of(cell) * H;
  PackLayout L;
  size_t o = 0;
  L.wih = o; o += ms::align_up(ndir * GH * In * sizeof(float), 256);
  L.bias_x = o; o += ms::align_up(ndir * GH * sizeof(float), 256);
  size_t whh_bytes = ndir * GH * H * sizeof(float);
  if (use_gru_persistent(cell, H, ndir)) whh_bytes = (size_t)ndir * (H / gru_units(H)) * 128 * H;  // 32 packed rows per workgroup
  L.whh = o; o += ms::align_up(whh_bytes, 256);
  L.bhh = o; o += ms::align_up(ndir * GH * sizeof(float), 256);
  L.total = o;
  return L;
}

struct WsLayout {
  size_t status, flags, xproj, xproj_bytes, hx, hx_bytes, state_h, state_c, dbg, row_off, xsplit, total;
  int xproj_slots;
  // projection buffer of a layer: slot = layer parity where there are two (the overlapped schedule fills the next layer's
  // while this layer's is being read)
  size_t xslot(int parity) const { return xproj + (size_t)(parity % xproj_slots) * xproj_bytes; }
};
// Stacks whose layers may run on ms_rnn_stack_forward's overlapped schedule: bidirectional wide-workgroup LSTM layers.
bool overlap_shape(int cell, int H, int ndir) {
  return (cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM) && H == 1024 && ndir == 2 && ms::precision_mode() != ms::PREC_F32;
}
WsLayout ws_layout(int cell, int T, int N, int H, int ndir, int In) {
  const size_t GH = (size_t)gates_of(cell) * H;
  const int npad = ms::cdiv(N, 32) * 32;
  WsLayout L;
  size_t o = 0;
  o += STATUS_BYTES;              // [0]: sticky time-out word (set by the kernels, read and cleared by ms_rnn_status)
  L.status = o; o += STATUS_BYTES;  // per-call word, zeroed by every layer call
  L.flags = o; o += ms::align_up((size_t)ndir * std::max(H / 8, 1) * sizeof(unsigned), 256);
  L.xproj_slots = overlap_shape(cell, H, ndir) ? 2 : 1;
  L.xproj_bytes = ms::align_up((size_t)T * N * ndir * GH * sizeof(float), 256);
  L.xproj = o; o += L.xproj_bytes * L.xproj_slots;
  // two slots per (stream, plane) by default; the two-stream kernel may use a ring of 2^lstm_ring_shift() slots
  // (the float32 two-stream kernel: ndir * 2 streams * 2 slots * 16 rows * H floats = the same ndir * 256 * H bytes)
  L.hx = o;
  L.hx_bytes = ms::align_up(((size_t)ndir * 2 * H * std::min(npad, 64) * sizeof(float)) << (lstm_ring_shift() - 1), 256);
  o += L.hx_bytes * HX_REGIONS;     // one exchange region per layer of a stack (ms_rnn_hx_preinit); a single call uses region 0
  L.state_h = o; o += ms::align_up((size_t)2 * ndir * N * H * sizeof(float), 256);
  L.state_c = o; o += ms::align_up((size_t)ndir * N * H * sizeof(float), 256);
  L.dbg = o; o += ms::align_up((size_t)ndir * std::max(H / 8, 1) * 16 * sizeof(unsigned long long), 256);
  L.row_off = o; o += ms::align_up((size_t)(T + 1) * sizeof(int), 256);   // packed rows: first row of every frame, and the total
  L.xsplit = o; o += ms::align_up((size_t)T * N * In * 4, 256);  // bf16 hi + lo planes of the layer input
  L.total = o;
  return L;
}

// ------------------------------------------------------------------------------------------------ packing

// dst row (d, j, g, u) <- src row g*H + 8j + u of direction d   (fast LSTM column order)
__global__ void pack_rows_fast_kernel(const float* __restrict__ w, float* __restrict__ dst, int H, int In) {
  const int row = blockIdx.x;  // 0 .. 4H-1 in packed order: j*32 + g*8 + u
  const int j = row / 32, g = (row % 32) / 8, u = row % 8;
  const float* src = w + (size_t)(g * H + 8 * j + u) * In;
  float* d = dst + (size_t)row * In;
  for (int k = threadIdx.x; k < In; k += blockDim.x) d[k] = src[k];
}

// same row order, written as bf16 hi / lo planes (plane stride = rows_total * In elements)
template <int P>
__global__ void pack_rows_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi,
                                       unsigned short* __restrict__ lo, int H, int In) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  const int row = blockIdx.x;
  const int j = row / 32, g = (row % 32) / 8, u = row % 8;
  const float* src = w + (size_t)(g * H + 8 * j + u) * In;
  for (int k = threadIdx.x; k < In; k += blockDim.x) {
    const float x = src[k];
    if (F16) {
      hi[(size_t)row * In + k] = __builtin_bit_cast(unsigned short, (_Float16)x);
    } else {
      unsigned hb, lb;
      ms::plane_split<HM>(x, hb, lb);
      hi[(size_t)row * In + k] = (unsigned short)hb;
      lo[(size_t)row * In + k] = (unsigned short)lb;
    }
  }
}

__global__ void pack_bias_fast_kernel(const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                      float* __restrict__ dst, int H) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= 4 * H) return;
  const int j = row / 32, g = (row % 32) / 8, u = row % 8;
  const int src = g * H + 8 * j + u;
  dst[row] = (b_ih ? b_ih[src] : 0.f) + (b_hh ? b_hh[src] : 0.f);
}

// whh_p[j][kq][r = g*8+u][e] = w_hh[g*H + 8j + u][4kq + e]
__global__ void pack_whh_fast_kernel(const float* __restrict__ w, float* __restrict__ dst, int H) {
  const size_t total = (size_t)4 * H * H;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 3;
    const int r = (i >> 2) & 31;
    const int kq = (i >> 7) % (H / 4);
    const int j = (i >> 7) / (H / 4);
    const int g = r >> 3, u = r & 7;
    dst[i] = w[(size_t)(g * H + 8 * j + u) * H + 4 * kq + e];
  }
}

// whh_s[j][plane][kg][r = g*8+u][e] (bf16) = hi / lo part of w_hh[g*H + 8j + u][8kg + e]
template <int P>
__global__ void pack_whh_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int H) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  const size_t total = (size_t)4 * H * H;  // elements per plane
  const int KG = H / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 7;
    const int r = (i >> 3) & 31;
    const int kg = (i >> 8) % KG;
    const int j = (i >> 8) / KG;
    const int g = r >> 3, u = r & 7;
    const float x = w[(size_t)(g * H + 8 * j + u) * H + 8 * kg + e];
    const size_t base = (size_t)j * 2 * KG * 256 + ((size_t)kg * 32 + r) * 8 + e;
    if (F16) {
      dst[base] = __builtin_bit_cast(unsigned short, (_Float16)x);
      dst[base + (size_t)KG * 256] = 0;
    } else {
      unsigned hb, lb;
      ms::plane_split<HM>(x, hb, lb);
      dst[base] = (unsigned short)hb;
      dst[base + (size_t)KG * 256] = (unsigned short)lb;
    }
  }
}

__global__ void copy_or_zero_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src ? src[i] : 0.f;
}

int blocks_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }

// ------------------------------------------------------------------------------------------------ generic step

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
// v_exp_f32 / v_rcp_f32 forms for the persistent kernels (1 ulp each; the generic path keeps
// the libm versions): the cell update sits on the per-step latency chain.
__device__ __forceinline__ float fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
}
__device__ __forceinline__ float fast_tanh(float v) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * v));
}
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float clamp11(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct StepP {
  const float* xproj;   // [steps*N][ndir*G*H], natural column order, b_ih folded in
  const float* whh;     // [ndir][G*H][H]
  const float* bhh;     // [ndir][G*H]
  const int32_t* lens;  // may be null
  const float* h_prev;  // [ndir][N][H]
  float* h_next;
  float* c_state;       // [ndir][N][H] (LSTM cells)
  float* out;           // [T][N][ndir*H]
  float* hn;
  float* cn;
  int s, steps, N, H, ndir;
};

template <int CELL>
__global__ __launch_bounds__(64) void rnn_step_generic_kernel(StepP p) {
  constexpr int G = (CELL == MS_CELL_LSTM || CELL == MS_CELL_HARD_LSTM) ? 4 : (CELL == MS_CELL_GRU ? 3 : 1);
  const int u = blockIdx.x, d = blockIdx.y, lane = threadIdx.x;
  const int t = d ? (p.steps - 1 - p.s) : p.s;
  const int H = p.H, N = p.N;
  const size_t GH = (size_t)G * H;
  const float* wbase = p.whh + (size_t)d * GH * H;
  for (int n = 0; n < N; ++n) {
    const float* hp = p.h_prev + ((size_t)d * N + n) * H;
    float part[G];
#pragma unroll
    for (int g = 0; g < G; ++g) part[g] = 0.f;
    for (int k = lane; k < H; k += 64) {
      const float hv = hp[k];
#pragma unroll
      for (int g = 0; g < G; ++g) part[g] += wbase[((size_t)g * H + u) * H + k] * hv;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) part[g] = wave_sum(part[g]);
    if (lane == 0) {
      const bool active = p.lens ? (t < p.lens[n]) : true;
      const size_t sidx = ((size_t)d * N + n) * H + u;
      const float* xp = p.xproj + ((size_t)t * N + n) * (p.ndir * GH) + d * GH;
      const float* bh = p.bhh + d * GH;
      const float hold = hp[u];
      float hnew, cnew = 0.f;
      if (CELL == MS_CELL_LSTM || CELL == MS_CELL_HARD_LSTM) {
        const float gi = xp[u] + (part[0] + bh[u]);
        const float gf = xp[H + u] + (part[1] + bh[H + u]);
        const float gg = xp[2 * H + u] + (part[2] + bh[2 * H + u]);
        const float go = xp[3 * H + u] + (part[3] + bh[3 * H + u]);
        const float cold = p.c_state[sidx];
        if (CELL == MS_CELL_LSTM) {
          cnew = sigmoidf_(gf) * cold + sigmoidf_(gi) * tanhf(gg);
          hnew = sigmoidf_(go) * tanhf(cnew);
        } else {
          cnew = clamp01(0.2f * gf + 0.5f) * cold + clamp01(0.2f * gi + 0.5f) * clamp11(gg);
          hnew = clamp01(0.2f * go + 0.5f) * clamp11(cnew);
        }
        const float cs = active ? cnew : cold;
        p.c_state[sidx] = cs;
        if (p.s == p.steps - 1) p.cn[sidx] = cs;
      } else if (CELL == MS_CELL_GRU) {
        const float r = sigmoidf_(xp[u] + (part[0] + bh[u]));
        const float z = sigmoidf_(xp[H + u] + (part[1] + bh[H + u]));
        const float nn = tanhf(xp[2 * H + u] + r * (part[2] + bh[2 * H + u]));
        hnew = (1.0f - z) * nn + z * hold;
      } else {
        hnew = tanhf(xp[u] + (part[0] + bh[u]));
      }
      const float hs = active ? hnew : hold;
      p.h_next[sidx] = hs;
      if (p.s == p.steps - 1) p.hn[sidx] = hs;
      p.out[((size_t)t * N + n) * (p.ndir * H) + d * H + u] = active ? hnew : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ streamed-weights step
//
// Cells / sizes whose W_hh cannot stay on chip (the reference's shipped DS2 config: 3 x GRU-2560, 78.6 MB of f32
// recurrent weights per layer): one launch per time step, W_hh streamed from L2 / Infinity Cache through exact-f32
// MFMA.  A workgroup owns 16 hidden units (all their gate rows) x up to 64 batch rows; its 4 waves split K = H in
// quarters (16x16x4 tiles: A = h rows, B = W_hh rows, both as float4 per lane = four k-steps per load), wave 0 adds the
// quarters (LDS) and applies the cell.  H % 64 == 0.
template <int CELL, int NT>
__global__ __launch_bounds__(256) void rnn_step_mfma_kernel(StepP p) {
  constexpr int G = (CELL == MS_CELL_LSTM || CELL == MS_CELL_HARD_LSTM) ? 4 : (CELL == MS_CELL_GRU ? 3 : 1);
  __shared__ f32x4 red[3][G][NT][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = lane & 15, kq = lane >> 4;
  const int u0 = blockIdx.x * 16, d = blockIdx.y, n0 = blockIdx.z * (16 * NT);
  const int H = p.H, N = p.N;
  const int t = d ? (p.steps - 1 - p.s) : p.s;
  const size_t GH = (size_t)G * H;
  const float* wbase = p.whh + (size_t)d * GH * H;
  const float* hbase = p.h_prev + (size_t)d * N * H;
  const int kbeg = wave * (H / 4), kend = kbeg + H / 4;

  f32x4 acc[G][NT];
  const float* arow[NT];
  const float* brow[G];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) arow[nt] = hbase + (size_t)min(n0 + nt * 16 + x, N - 1) * H + 4 * kq;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    brow[g] = wbase + ((size_t)g * H + u0 + x) * H + 4 * kq;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[g][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // A stage = 32 k-values = one whole 128-byte line of every W_hh / h row the wave touches: its two 16-deep halves are
  // requested back to back (two float4 per lane), so each line is fetched from L2 / Infinity Cache once instead of once
  // per half.  Two stages rotate: the loads of stage i+2 are issued as soon as stage i has been consumed.
  constexpr int D = 2;
  f32x4 a[D][2][NT], b[D][2][G];
  auto load_stage = [&](int st, int kb) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (kb + 16 * hf < kend) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[st][hf][nt] = *reinterpret_cast<const f32x4*>(arow[nt] + kb + 16 * hf);
#pragma unroll
        for (int g = 0; g < G; ++g) b[st][hf][g] = *reinterpret_cast<const f32x4*>(brow[g] + kb + 16 * hf);
      }
    }
  };
#pragma unroll
  for (int st = 0; st < D; ++st)
    if (kbeg + 32 * st < kend) load_stage(st, kbeg + 32 * st);
  for (int kb0 = kbeg; kb0 < kend; kb0 += 32 * D) {
#pragma unroll
    for (int st = 0; st < D; ++st) {
      const int kb = kb0 + 32 * st;
      if (kb < kend) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          if (kb + 16 * hf < kend) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int g = 0; g < G; ++g)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][hf][nt][e], b[st][hf][g][e], acc[g][nt], 0, 0, 0);
          }
        }
        if (kb + 32 * D < kend) load_stage(st, kb + 32 * D);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) red[wave - 1][g][nt][lane] = acc[g][nt];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int w2 = 0; w2 < 3; ++w2) acc[g][nt] += red[w2][g][nt][lane];

  // lane holds (batch row n0 + nt*16 + 4*kq + r, unit u0 + x) for r = 0..3
  const int u = u0 + x;
  const float* bh = p.bhh + d * GH;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + nt * 16 + 4 * kq + r;
      if (n >= N) continue;
      const bool active = p.lens ? (t < p.lens[n]) : true;
      const size_t sidx = ((size_t)d * N + n) * H + u;
      const float* xp = p.xproj + ((size_t)t * N + n) * (p.ndir * GH) + d * GH;
      const float hold = hbase[(size_t)n * H + u];
      float hnew, cnew = 0.f;
      if (CELL == MS_CELL_LSTM || CELL == MS_CELL_HARD_LSTM) {
        const float gi = xp[u] + (acc[0][nt][r] + bh[u]);
        const float gf = xp[H + u] + (acc[1 % G][nt][r] + bh[H + u]);
        const float gg = xp[2 * H + u] + (acc[2 % G][nt][r] + bh[2 * H + u]);
        const float go = xp[3 * H + u] + (acc[3 % G][nt][r] + bh[3 * H + u]);
        const float cold = p.c_state[sidx];
        if (CELL == MS_CELL_LSTM) {
          cnew = sigmoidf_(gf) * cold + sigmoidf_(gi) * tanhf(gg);
          hnew = sigmoidf_(go) * tanhf(cnew);
        } else {
          cnew = clamp01(0.2f * gf + 0.5f) * cold + clamp01(0.2f * gi + 0.5f) * clamp11(gg);
          hnew = clamp01(0.2f * go + 0.5f) * clamp11(cnew);
        }
        const float cs = active ? cnew : cold;
        p.c_state[sidx] = cs;
        if (p.s == p.steps - 1) p.cn[sidx] = cs;
      } else if (CELL == MS_CELL_GRU) {
        const float rg = sigmoidf_(xp[u] + (acc[0][nt][r] + bh[u]));
        const float z = sigmoidf_(xp[H + u] + (acc[1 % G][nt][r] + bh[H + u]));
        const float nn = tanhf(xp[2 * H + u] + rg * (acc[2 % G][nt][r] + bh[2 * H + u]));
        hnew = (1.0f - z) * nn + z * hold;
      } else {
        hnew = tanhf(xp[u] + (acc[0][nt][r] + bh[u]));
      }
      const float hs = active ? hnew : hold;
      p.h_next[sidx] = hs;
      if (p.s == p.steps - 1) p.hn[sidx] = hs;
      p.out[((size_t)t * N + n) * (p.ndir * H) + d * H + u] = active ? hnew : 0.f;
    }
}

template <int CELL>
static void launch_step_mfma(const StepP& p, hipStream_t stream) {
  const int tiles = ms::cdiv(p.N, 16);
  if (tiles == 1)
    hipLaunchKernelGGL((rnn_step_mfma_kernel<CELL, 1>), dim3(p.H / 16, p.ndir, 1), dim3(256), 0, stream, p);
  else if (tiles == 2)
    hipLaunchKernelGGL((rnn_step_mfma_kernel<CELL, 2>), dim3(p.H / 16, p.ndir, 1), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((rnn_step_mfma_kernel<CELL, 4>), dim3(p.H / 16, p.ndir, ms::cdiv(p.N, 64)), dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ persistent LSTM

struct LstmP {
  const float* xproj;   // [steps*N][ndir*4H], column = d*4H + j*32 + g*8 + u, both biases folded in
  const float* whh;     // [ndir][J][H/4][32][4]
  const int32_t* lens;  // may be null
  const float* h0;
  const float* c0;      // [ndir][N][H] or null
  float* out;           // [T][N][ndir*H]
  float* hn;
  float* cn;
  float* hx;            // [ndir][2][H/4][NPAD][4]
  unsigned* flags;      // [ndir][J]
  unsigned* status;     // [0]: nonzero = a wait timed out
  unsigned long long* dbg;  // [ndir*J][8] stamp sums (diagnostic build only; the wide kernel: [workgroup][16])
  int steps, N, n_base, N_total, H, ndir, J, NPAD;
  int d_base;      // two-stream kernel: direction of workgroup 0 (a bidirectional layer whose directions run as two launches)
  int grp_wgs;     // two-stream kernel: workgroups per batch group of <= 32 rows when one launch holds two groups side by side (0: one group)
  int poll_sleep;  // s_sleep(1) repetitions between polls of the exchange buffer
  int xcd_map;     // wide kernel: (group, direction) -> XCD pair (two groups, two directions, J / 2 = 64)
  int ring_shift;  // two-stream kernel: log2 of the number of exchange slots per (stream, plane) (1 = two slots)
  // two-stream kernel: when set, the layer output goes out as the NEXT layer's GEMM operand planes [T*N][ndir*H] (bf16
  // hi / lo, or one fp16 plane) instead of float32 `out` -- same bytes, and the separate plane-split pass disappears
  unsigned short* out_hi;
  unsigned short* out_lo;
  // wide kernel, packed rows: row_off[t] = first row of frame t in `xproj` and in the planes, which then hold only the
  // rows (t, n) with t < lens[n] (lens sorted in decreasing order), frame after frame; null = every frame has N_total rows
  const int32_t* row_off;
  // wide kernel, time segments: this launch runs steps [s_begin, s_end) of the layer's `steps` (the step clock: forward t = s,
  // backward t = steps - 1 - s).  A launch with s_begin > 0 continues the previous one: h0 / c0 are then the state that one
  // left in hn / cn, and the exchange buffer still holds the h it published last -- in the slot and with the tag the first
  // step here expects, because the epoch clock is the sequence time, not the launch's step count.
  int s_begin, s_end;
};

__device__ __forceinline__ f32x4 load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, /*aux: sc1*/ 16);
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ void store_sc1_f32(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A wait gave up: raise the per-call word (peers stop waiting too) and the sticky word one STATUS block below it, which
// no launch clears -- ms_rnn_status reads that one, so a caller may check once after several layer calls.
__device__ __forceinline__ void flag_timeout(unsigned* status) {
  __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(status - STATUS_BYTES / sizeof(unsigned), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-level wait until every producer flag this wave depends on has reached `epoch`.
// Returns false when it gave up (peer not resident / dead): the caller stops waiting for
// the rest of the launch and the host reports MS_ERR_TIMEOUT.
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int count, unsigned epoch, unsigned* status, int lane) {
  const unsigned long long t0 = wall_clock64();
  unsigned spins = 0;
  for (;;) {
    unsigned v = epoch;
    if (lane < count) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all((int)(v - epoch) >= 0)) return true;
    if ((++spins & 63u) == 0) {
      const unsigned dead = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dead != 0 || wall_clock64() - t0 > SPIN_LIMIT_TICKS) {
        if (lane == 0) flag_timeout(status);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int NB, bool HARD, bool PIPE, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void lstm_persistent_kernel(LstmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                        // [H/4][32][4]
  float* red = smem + (size_t)p.H * 32;    // [4][32][RED_STRIDE]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int d = blockIdx.x / p.J, j = blockIdx.x % p.J;
  const int H = p.H, N = p.N, KQ = H / 4;
  const int nl = tid >> 3, u = tid & 7;  // cell owned by this thread: batch row nl (+32b), unit 8j+u
  const int unit = 8 * j + u;

  // --- resident recurrent weights
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.whh + ((size_t)d * p.J + j) * H * 32);
    f32x4* dst = reinterpret_cast<f32x4*>(Ws);
    for (int i = tid; i < H * 8; i += 256) dst[i] = src[i];
  }

  float c[NB], h[NB];
  int len_n[NB];
  float* hx_d = p.hx + (size_t)d * 2 * KQ * p.NPAD * 4;
  const int hx_slot = ((unit >> 2) * p.NPAD) * 4 + (unit & 3);  // + n*4, + parity*KQ*NPAD*4
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = b * 32 + nl;
    const bool valid = n < N;
    const size_t sidx = ((size_t)d * p.N_total + p.n_base + n) * H + unit;
    h[b] = (valid && p.h0) ? p.h0[sidx] : 0.f;
    c[b] = (valid && p.c0) ? p.c0[sidx] : 0.f;
    len_n[b] = valid ? (p.lens ? p.lens[p.n_base + n] : p.steps) : 0;
    store_sc1_f32(hx_d + hx_slot + n * 4, h[b]);  // h_{-1} into parity 0
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* my_flag = p.flags + d * p.J + j;
  if (tid == 0) __hip_atomic_store(my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  const __amdgpu_buffer_rsrc_t hx_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(hx_d, 0, 2 * KQ * p.NPAD * 16, 0x00020000);
  const unsigned* wave_flags = p.flags + d * p.J + wave * (p.J / 4);
  const int kq_base = wave * (KQ / 4);
  const int xcols = p.ndir * 4 * H;
  bool alive = true;
  unsigned long long st_sum[4] = {0, 0, 0, 0}, st_prev = 0;
  if (STAMP) st_prev = wall_clock64();

  for (int s = 0; s < p.steps; ++s) {
    const int t = d ? (p.steps - 1 - s) : s;
    const int par = s & 1;

    // gate pre-activations of the input projection for this frame (in flight during the wait)
    float xg[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = b * 32 + nl;
      const float* xp = p.xproj + ((size_t)t * p.N_total + p.n_base + n) * xcols + d * 4 * H + j * 32 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[b][g] = (n < N) ? xp[g * 8] : 0.f;
    }

    if (alive) alive = wait_flags(wave_flags, p.J / 4, (unsigned)(s + 1), p.status, lane);
    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[0] += now - st_prev; st_prev = now; }

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int hx_par = par * KQ * p.NPAD * 16;  // bytes
    const int iters = KQ / 8;                   // k-quad pairs per wave
    if (PIPE) {
      // H % 256 == 0: chunks of 8 k-quad pairs, the next chunk's sc1 loads in flight under
      // the current chunk's MFMAs (static register indexing throughout)
      f32x4 a0[8][NB], a1[8][NB];
      auto issue = [&](int it0, f32x4(&a)[8][NB]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int kq = kq_base + 2 * (it0 + i) + half;
#pragma unroll
          for (int b = 0; b < NB; ++b) a[i][b] = load_sc1_b128(hx_rsrc, hx_par + (kq * p.NPAD + b * 32 + l31) * 16);
        }
      };
      auto compute = [&](int it0, f32x4(&a)[8][NB]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int kq = kq_base + 2 * (it0 + i) + half;
          const f32x4 bw = *reinterpret_cast<const f32x4*>(Ws + (kq * 32 + l31) * 4);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][b][e], bw[e], acc[b], 0, 0, 0);
        }
      };
      issue(0, a0);
      for (int it0 = 0; it0 < iters; it0 += 16) {
        const bool more1 = it0 + 8 < iters;
        if (more1) issue(it0 + 8, a1);
        compute(it0, a0);
        if (it0 + 16 < iters) issue(it0 + 16, a0);
        if (more1) compute(it0 + 8, a1);
      }
    } else {
      for (int it = 0; it < iters; ++it) {
        const int kq = kq_base + 2 * it + half;
        const f32x4 bw = *reinterpret_cast<const f32x4*>(Ws + (kq * 32 + l31) * 4);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f32x4 a = load_sc1_b128(hx_rsrc, hx_par + (kq * p.NPAD + b * 32 + l31) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bw[e], acc[b], 0, 0, 0);
        }
      }
    }

    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[1] += now - st_prev; st_prev = now; }
    float hout[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b > 0) __syncthreads();  // previous tile's reads of `red` are done
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 32 + ms::mfma32_row(r, lane)) * RED_STRIDE + l31] = acc[b][r];
      __syncthreads();
      float gsum[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = xg[b][g];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) v += red[(w2 * 32 + nl) * RED_STRIDE + g * 8 + u];
        gsum[g] = v;
      }
      float cnew, hnew;
      if (HARD) {
        cnew = clamp01(0.2f * gsum[1] + 0.5f) * c[b] + clamp01(0.2f * gsum[0] + 0.5f) * clamp11(gsum[2]);
        hnew = clamp01(0.2f * gsum[3] + 0.5f) * clamp11(cnew);
      } else {
        cnew = sigmoidf_(gsum[1]) * c[b] + sigmoidf_(gsum[0]) * tanhf(gsum[2]);
        hnew = sigmoidf_(gsum[3]) * tanhf(cnew);
      }
      const bool active = t < len_n[b];
      c[b] = active ? cnew : c[b];
      h[b] = active ? hnew : h[b];
      hout[b] = active ? hnew : 0.f;
      store_sc1_f32(hx_d + (par ^ 1) * KQ * p.NPAD * 4 + hx_slot + (b * 32 + nl) * 4, h[b]);
    }
    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[2] += now - st_prev; st_prev = now; }
    // publish h_s: every storing wave drains, then one lane raises this producer's epoch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(s + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[3] += now - st_prev; st_prev = now; }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = b * 32 + nl;
      if (n < N) p.out[((size_t)t * p.N_total + p.n_base + n) * (p.ndir * H) + d * H + unit] = hout[b];
    }
  }
  if (STAMP && (tid & 63) == 0) {
    for (int k = 0; k < 4; ++k) atomicAdd(&p.dbg[(size_t)blockIdx.x * 8 + k], st_sum[k]);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = b * 32 + nl;
    if (n < N) {
      const size_t sidx = ((size_t)d * p.N_total + p.n_base + n) * H + unit;
      p.hn[sidx] = h[b];
      p.cn[sidx] = c[b];
    }
  }
}


// Epoch clock of the tagged exchange.  The 1-bit tag replaces the LSB of the published bf16 hi and lo words, so the
// value a consumer reconstructs depends (at the 2^-17 level) on the tag.  The clock is therefore the SEQUENCE time t of
// the step, not the launch's step index: forward u = s = t, backward u = steps-1-s = t.  Slot u & 1 is read expecting
// tag (u >> 1) & 1, h of the next step is written to the other slot with that step's tag.  A sequence's outputs are
// then bit-identical whatever the longest sequence of its batch is (utterance shards reproduce the whole batch).
struct EpochClock {
  int par;        // slot read at this step
  unsigned em;    // expected tag replicated to both 16-bit halves of a dword
  int wpar;       // slot the next step's h is written to
  unsigned wtag;  // tag of the h written for the next step
};
// rs = log2 of the number of slots (1: the two-slot scheme; larger: a ring, so that a slot's address is not touched
// again for 2^rs steps and a cached copy of it is long evicted before it could alias)
__device__ __forceinline__ EpochClock epoch_clock(int d, int s, int steps, int rs = 1) {
  const int u = d ? (steps - 1 - s) : s;
  const int un = d ? (u - 1) : (u + 1);
  const int mask = (1 << rs) - 1;
  EpochClock c;
  c.par = u & mask;
  c.em = ((u >> rs) & 1) ? 0x00010001u : 0u;
  c.wpar = un & mask;                      // two's complement: -1 & mask = last slot (written by the final step, never read)
  c.wtag = (unsigned)((un >> rs) & 1);
  return c;
}
// slot and tag of the initial state (the h read by the first step)
__device__ __forceinline__ int epoch_par0(int d, int steps, int rs = 1) { return d ? ((steps - 1) & ((1 << rs) - 1)) : 0; }
__device__ __forceinline__ unsigned epoch_tag0(int d, int steps, int rs = 1) {
  return d ? (unsigned)(((steps - 1) >> rs) & 1) : 0u;
}

__device__ __forceinline__ unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_val(unsigned bits) { return __uint_as_float(bits << 16); }

// ---- packed rows (round 3): a batch of ragged lengths, sorted in decreasing order, has sum(lens) rows that exist out of
// steps * N.  torch's packed sequences (rnn.py:174-181) drop the others; so does the wide-workgroup path when the caller
// allows it (MS_RNN_PACKED_ROWS): the operand planes and the projection hold frame 0's rows, then frame 1's, ...
// row_off[t] = rows of the frames before t = sum over n of min(lens[n], t); row_off[steps] = the number of rows.
__global__ __launch_bounds__(256) void row_offsets_kernel(const int32_t* __restrict__ lens, int N, int steps,
                                                          int32_t* __restrict__ row_off) {
  for (int t = threadIdx.x; t <= steps; t += blockDim.x) {
    int acc = 0;
    for (int n = 0; n < N; ++n) acc += min(max(lens[n], 0), t);
    row_off[t] = acc;
  }
}

// x f32 [steps][N][In] -> hi / lo planes of the rows that exist, packed (split_planes_kernel's arithmetic)
template <bool HM>
__global__ __launch_bounds__(256) void split_planes_packed_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi,
                                                                  unsigned short* __restrict__ lo,
                                                                  const int32_t* __restrict__ lens,
                                                                  const int32_t* __restrict__ row_off, int N, int In) {
  const int t = blockIdx.x / N, n = blockIdx.x % N;
  if (t >= lens[n]) return;
  const float4* src = reinterpret_cast<const float4*>(x + (size_t)blockIdx.x * In);
  const size_t dst = ((size_t)row_off[t] + n) * In;
  for (int k = threadIdx.x; k < In / 4; k += blockDim.x) {
    const float4 v = src[k];
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ms::plane_split<HM>(f[e], h[e], l[e]);
    *reinterpret_cast<uint2*>(hi + dst + 4 * k) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2*>(lo + dst + 4 * k) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
  }
}

// Exchange-buffer initialisation: every 16-bit word of slot p of direction d gets the tag that is NOT the first one
// expected there (all-ones or all-zeros words), so a poll that runs ahead of the producers never validates.
// Layout of a direction: ... [parity][slab_bytes] ..., i.e. parity = (byte offset / slab_bytes) & 1.
// The same launch clears the per-call status word and the epoch flags (`zero_words` words at `zero_base`), which used to
// be a memset of its own before every layer call and between batch groups.
// `ndir` regions of words_per_dir words follow each other; region r belongs to direction r % dir_mod (the wide path lays
// out [group][direction], so one launch initialises every batch group: ndir = groups * dir_mod).
__global__ void hx_init_kernel(unsigned* __restrict__ hx, size_t words_per_dir, size_t slab_words, int ndir, int steps,
                               int rs, unsigned* __restrict__ zero_base, int zero_words, int dir_mod) {
  const size_t total = words_per_dir * ndir;
  const int mask = (1 << rs) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)zero_words; i += (size_t)gridDim.x * blockDim.x)
    zero_base[i] = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / words_per_dir) % dir_mod;
    const int par = (int)(((i % words_per_dir) / slab_words) & mask);
    int first_u;  // first clock value at which slot `par` is read (forward: counting up from 0; backward: down from steps-1)
    if (d == 0) first_u = par;
    else first_u = (steps - 1) - (((steps - 1) - par) & mask);
    const unsigned expected = (unsigned)((first_u >> rs) & 1);   // arithmetic shift: a slot never read gives any value
    hx[i] = expected ? 0u : 0xFFFFFFFFu;
  }
}

// ------------------------------------------------------------------------------------------------ persistent LSTM, split-bf16
//
// Same decomposition as lstm_persistent_kernel, two changes:
//  * operands are bf16 (hi, lo) pairs and the product is three v_mfma_f32_32x32x16_bf16
//    (hi*hi + lo*hi + hi*lo, f32 accumulate): 48 MFMA x 32 cycles per wave and step instead
//    of 128 x 64;
//  * the hand-off carries its own validity: every 16-bit element of the exchanged h has its
//    least significant mantissa bit replaced by a 1-bit epoch tag.  The lo part is computed
//    against the TAGGED hi, so hi's lost bit is recovered by lo; lo's own tag bit is not: the
//    copy of h that other workgroups multiply is h * (1 + e) with |e| <= ~2^-13 in the worst
//    case (|lo| <= 1.5 * 2^-7 |h|, and its tagged 8-bit significand is off by up to 1.5 * 2^-7
//    of that), ~2^-15 typically, against 2^-17 for an untagged split; the own state, the
//    outputs and h_n / c_n are not affected.  End to end: DESIGN.md 2.  A consumer simply loads
//    (sc1) and re-loads until every element of its chunk shows the expected tag -- no flag,
//    no producer-side drain, no second hop (MI355X_MICROARCH.md "R2: the data IS the flag",
//    with per-element tags so even a torn 16-byte transfer is detected).  The exchange
//    buffers are memset to 0xFF (tag 1 = invalid for the first epoch) before every launch.
// hx layout per direction: [plane hi|lo][parity][k/8][N pad 32][8 bf16].

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


// sc1 + the intrinsic's "volatile" marker (aux bit 31): a re-load inside a polling loop
// must never be merged with the previous one.
__device__ __forceinline__ u32x4 load_sc1_u128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, (int)(0x80000000u | 16u));
}

// Packs this thread's h (unit u = lane & 7 of batch row lane >> 3) with its 7 neighbours
// (DPP row shifts: lane i reads lane i+k of its 16-lane row) and lets lane u == 0 store the
// 8 hi parts and the 8 lo parts (two 16-byte sc1 stores).
template <int K>
__device__ __forceinline__ unsigned row_shl(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + K, 0xF, 0xF, true);
}
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

// P = PREC_F16: the hi plane carries tagged fp16 values and the lo plane is not used.  PREC_F16X3: both planes fp16 (the
// tagged hi's lost bit is recovered by lo, lo's own costs 2^-11 of lo = ~2^-22 of h).  h is a cell output: |h| <= 1.
template <int P = ms::PREC_BF16X3>
__device__ __forceinline__ void publish_split(float hval, unsigned tag, __amdgpu_buffer_rsrc_t rsrc, int off_hi, int off_lo,
                                              int lane, bool row_exists = true) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  unsigned hi, lo;
  if (F16) {
    hi = ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)hval) & 0xFFFEu) | tag;
    lo = 0u;
  } else {
    hi = (ms::plane_bits_bounded<HM>(hval) & 0xFFFEu) | tag;
    lo = (ms::plane_bits_bounded<HM>(hval - ms::plane_val<HM>(hi)) & 0xFFFEu) | tag;
  }
  const unsigned v = hi | (lo << 16);
  unsigned g[8];
  g[0] = v;
  g[1] = row_shl<1>(v); g[2] = row_shl<2>(v); g[3] = row_shl<3>(v); g[4] = row_shl<4>(v);
  g[5] = row_shl<5>(v); g[6] = row_shl<6>(v); g[7] = row_shl<7>(v);
  if ((lane & 7) == 0 && row_exists) {   // (a padding row of the wide kernel is neither published nor pulled)
    u32x4 oh, ol;
    oh[0] = (g[0] & 0xFFFFu) | (g[1] << 16); oh[1] = (g[2] & 0xFFFFu) | (g[3] << 16);
    oh[2] = (g[4] & 0xFFFFu) | (g[5] << 16); oh[3] = (g[6] & 0xFFFFu) | (g[7] << 16);
    ol[0] = (g[0] >> 16) | (g[1] & 0xFFFF0000u); ol[1] = (g[2] >> 16) | (g[3] & 0xFFFF0000u);
    ol[2] = (g[4] >> 16) | (g[5] & 0xFFFF0000u); ol[3] = (g[6] >> 16) | (g[7] & 0xFFFF0000u);
    __builtin_amdgcn_raw_buffer_store_b128(oh, rsrc, off_hi, 0, /*aux: sc1*/ 16);
    if (!F16) __builtin_amdgcn_raw_buffer_store_b128(ol, rsrc, off_lo, 0, /*aux: sc1*/ 16);
  }
}

// NCH > 0: the wave's K-quarter is NCH chunks of 4 k-steps (K = 64 each), all held in registers;
// NCH == 0: any H % 64 == 0, one k-step at a time.
template <int NB, int NCH, bool HARD, bool STAMP = false, bool HM = false>
__global__ __launch_bounds__(256, 1) void lstm_persistent_split_kernel(LstmP p) {
  constexpr int P = HM ? ms::PREC_F16X3 : ms::PREC_BF16X3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = p.H, N = p.N, KG = H / 8;
  const char* Wbytes = reinterpret_cast<const char*>(smem);  // [hi|lo][KG][32][8 bf16]
  float* red = smem + (size_t)H * 32;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int d = blockIdx.x / p.J, j = blockIdx.x % p.J;
  const int nl = tid >> 3, u = tid & 7;
  const int unit = 8 * j + u;

  {
    const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.whh) + ((size_t)d * p.J + j) * 128 * H);
    u32x4* dst = reinterpret_cast<u32x4*>(smem);
    for (int i = tid; i < H * 8; i += 256) dst[i] = src[i];
  }

  const int plane_bytes = 2 * KG * p.NPAD * 16;
  char* hx_d = reinterpret_cast<char*>(p.hx) + (size_t)d * 2 * plane_bytes;
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx_d, 0, 2 * plane_bytes, 0x00020000);

  float c[NB], h[NB];
  int len_n[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = b * 32 + nl;
    const bool valid = n < N;
    const size_t sidx = ((size_t)d * p.N_total + p.n_base + n) * H + unit;
    h[b] = (valid && p.h0) ? p.h0[sidx] : 0.f;
    c[b] = (valid && p.c0) ? p.c0[sidx] : 0.f;
    len_n[b] = valid ? (p.lens ? p.lens[p.n_base + n] : p.steps) : 0;
    const int off = epoch_par0(d, p.steps) * KG * p.NPAD * 16 + (j * p.NPAD + n) * 16;  // slot read by the first step
    publish_split<P>(h[b], epoch_tag0(d, p.steps), hx_rsrc, off, plane_bytes + off, lane);
  }
  __syncthreads();  // weights are in LDS

  const int kg_base = wave * (KG / 4);
  const int xcols = p.ndir * 4 * H;
  bool alive = true;
  unsigned long long st_sum[4] = {0, 0, 0, 0}, st_prev = 0;
  if (STAMP) st_prev = wall_clock64();

  // NCH > 0: this wave's share of W_hh (its K-quarter x 32 gate rows, hi and lo) lives in
  // 128 VGPRs per lane for the whole sequence -- the MFMA B operand never touches LDS.
  u32x4 wreg_h[NCH > 0 ? 4 * NCH : 1], wreg_l[NCH > 0 ? 4 * NCH : 1];
  if (NCH > 0) {
#pragma unroll
    for (int i = 0; i < 4 * NCH; ++i) {
      const int kg = kg_base + 2 * i + half;
      wreg_h[i] = *reinterpret_cast<const u32x4*>(Wbytes + (kg * 32 + l31) * 16);
      wreg_l[i] = *reinterpret_cast<const u32x4*>(Wbytes + KG * 512 + (kg * 32 + l31) * 16);
    }
  }

  for (int s = 0; s < p.steps; ++s) {
    const int t = d ? (p.steps - 1 - s) : s;
    const EpochClock ec = epoch_clock(d, s, p.steps);
    const int par = ec.par;
    const unsigned em = ec.em;  // expected tag, replicated to both halves

    float xg[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = b * 32 + nl;
      const float* xp = p.xproj + ((size_t)t * p.N_total + p.n_base + n) * xcols + d * 4 * H + j * 32 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[b][g] = (n < N) ? xp[g * 8] : 0.f;
    }

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int par_off = par * KG * p.NPAD * 16;
    const unsigned long long t_wait0 = wall_clock64();
    unsigned spins = 0;

    // one k-step = 16 values of k = two 8-groups (lane half picks one); operands of k-group kg:
    //   A (h):  hi at par_off + (kg*NPAD + n)*16, lo one plane further
    //   B (W):  LDS (kg*32 + row)*16, lo KG*512 bytes further
    auto a_off = [&](int kg, int b) { return par_off + (kg * p.NPAD + b * 32 + l31) * 16; };
    auto mfma3 = [&](const u32x4& ah, const u32x4& al, int kg, f32x16& accb) {
      const u32x4 bh = *reinterpret_cast<const u32x4*>(Wbytes + (kg * 32 + l31) * 16);
      const u32x4 bl = *reinterpret_cast<const u32x4*>(Wbytes + KG * 512 + (kg * 32 + l31) * 16);
      accb = ms::mfma_32x32x16<HM>(ah, bh, accb);
      accb = ms::mfma_32x32x16<HM>(al, bh, accb);
      accb = ms::mfma_32x32x16<HM>(ah, bl, accb);
    };
    auto give_up = [&]() -> bool {  // called while waiting; true = stop waiting for the rest of the launch
      if ((++spins & 63u) != 0) { __builtin_amdgcn_s_sleep(1); return false; }
      const unsigned dead = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dead != 0 || wall_clock64() - t_wait0 > SPIN_LIMIT_TICKS) {
        if (lane == 0) flag_timeout(p.status);
        return true;
      }
      return false;
    };

    if (NCH > 0) {
      auto mfma3r = [&](const u32x4& xh, const u32x4& xl, const u32x4& bh, const u32x4& bl, f32x16& accb) {
        accb = ms::mfma_32x32x16<HM>(xh, bh, accb);
        accb = ms::mfma_32x32x16<HM>(xl, bh, accb);
        accb = ms::mfma_32x32x16<HM>(xh, bl, accb);
      };
      u32x4 ah[NCH > 0 ? NCH : 1][4][NB], al[NCH > 0 ? NCH : 1][4][NB];
      auto issue = [&](int cidx, u32x4(&xh)[4][NB], u32x4(&xl)[4][NB]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kg = kg_base + 2 * (4 * cidx + ks) + half;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            xh[ks][b] = load_sc1_u128(hx_rsrc, a_off(kg, b));
            xl[ks][b] = load_sc1_u128(hx_rsrc, plane_bytes + a_off(kg, b));
          }
        }
      };
      auto fresh = [&](u32x4(&xh)[4][NB], u32x4(&xl)[4][NB]) -> bool {
        unsigned bad = 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= (xh[ks][b][e] ^ em) | (xl[ks][b][e] ^ em);
        return !__any((bad & 0x00010001u) != 0);
      };
      // poll on the first chunk only, then fetch the rest in one burst
      issue(0, ah[0], al[0]);
      while (alive && !fresh(ah[0], al[0])) {
        if (give_up()) { alive = false; break; }
        issue(0, ah[0], al[0]);
      }
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[0] += now - st_prev; st_prev = now; }
#pragma unroll
      for (int cidx = 1; cidx < NCH; ++cidx) issue(cidx, ah[cidx], al[cidx]);
#pragma unroll
      for (int cidx = 0; cidx < NCH; ++cidx) {
        if (cidx > 0) {
          while (alive && !fresh(ah[cidx], al[cidx])) {
            if (give_up()) { alive = false; break; }
            issue(cidx, ah[cidx], al[cidx]);
          }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int b = 0; b < NB; ++b) mfma3r(ah[cidx][ks][b], al[cidx][ks][b], wreg_h[4 * cidx + ks], wreg_l[4 * cidx + ks], acc[b]);
        }
      }
    } else {
      const int ksteps = KG / 8;  // per wave
      for (int ks = 0; ks < ksteps; ++ks) {
        const int kg = kg_base + 2 * ks + half;
        u32x4 xh[NB], xl[NB];
        for (;;) {
          unsigned bad = 0;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            xh[b] = load_sc1_u128(hx_rsrc, a_off(kg, b));
            xl[b] = load_sc1_u128(hx_rsrc, plane_bytes + a_off(kg, b));
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= (xh[b][e] ^ em) | (xl[b][e] ^ em);
          }
          if (!alive || !__any((bad & 0x00010001u) != 0)) break;
          if (give_up()) { alive = false; break; }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) mfma3(xh[b], xl[b], kg, acc[b]);
      }
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[0] += now - st_prev; st_prev = now; }
    }
    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[1] += now - st_prev; st_prev = now; }

    const unsigned wtag = ec.wtag;
    float hout[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      __syncthreads();  // previous reads of `red` (earlier tile or earlier step) are done
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 32 + ms::mfma32_row(r, lane)) * RED_STRIDE + l31] = acc[b][r];
      __syncthreads();
      float gsum[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = xg[b][g];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) v += red[(w2 * 32 + nl) * RED_STRIDE + g * 8 + u];
        gsum[g] = v;
      }
      float cnew, hnew;
      if (HARD) {
        cnew = clamp01(0.2f * gsum[1] + 0.5f) * c[b] + clamp01(0.2f * gsum[0] + 0.5f) * clamp11(gsum[2]);
        hnew = clamp01(0.2f * gsum[3] + 0.5f) * clamp11(cnew);
      } else {
        cnew = fast_sigmoid(gsum[1]) * c[b] + fast_sigmoid(gsum[0]) * fast_tanh(gsum[2]);
        hnew = fast_sigmoid(gsum[3]) * fast_tanh(cnew);
      }
      const bool active = t < len_n[b];
      c[b] = active ? cnew : c[b];
      h[b] = active ? hnew : h[b];
      hout[b] = active ? hnew : 0.f;
      const int off = (par ^ 1) * KG * p.NPAD * 16 + (j * p.NPAD + b * 32 + nl) * 16;
      publish_split<P>(h[b], wtag, hx_rsrc, off, plane_bytes + off, lane);
    }
    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[2] += now - st_prev; st_prev = now; }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = b * 32 + nl;
      if (n < N) p.out[((size_t)t * p.N_total + p.n_base + n) * (p.ndir * H) + d * H + unit] = hout[b];
    }
    if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[3] += now - st_prev; st_prev = now; }
  }
  if (STAMP && (tid & 63) == 0) {
    for (int k = 0; k < 4; ++k) atomicAdd(&p.dbg[(size_t)blockIdx.x * 8 + k], st_sum[k]);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = b * 32 + nl;
    if (n < N) {
      const size_t sidx = ((size_t)d * p.N_total + p.n_base + n) * H + unit;
      p.hn[sidx] = h[b];
      p.cn[sidx] = c[b];
    }
  }
}


// ------------------------------------------------------------------------------------------------ persistent LSTM, split-bf16, two streams
//
// The recurrence of one batch is a latency chain: publish h_t -> visible on the other CUs
// (~1 us) -> load -> MFMA -> cell -> publish h_{t+1}.  Batch rows are independent, so the 32 rows
// are cut into two streams of 16 that alternate on the same workgroup: while stream A's h_t is
// in flight through the fabric, the workgroup computes stream B's step, and vice versa.  The
// MFMA shape drops to 16x16x32 (16 batch rows x 16 gate rows, two column tiles), which costs
// exactly half the cycles of the 32x32x16 tile, so no matrix throughput is lost.
// H in {256, 512, 768, 1024}, N <= 32 per launch.  hx layout per direction: [stream][plane hi|lo][parity][k/8][16][8 bf16].
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int KS, bool HARD, bool STAMP = false, int P = ms::PREC_BF16X3>
__global__ __launch_bounds__(256, 1) void lstm_persistent_split2_kernel(LstmP p) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  constexpr int H = 128 * KS, KG = H / 8;  // KS k-steps (K = 32) per wave; H in {256, 512, 768, 1024, 1280, 1536, 2048}
  constexpr int RED2 = 4 * 16 * RED_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;  // [stream][4 waves][16 rows][RED_STRIDE]
  // (round 6) H = 2048: a wave's share of W_hh is 256 registers per lane, with the operand chunks beside it the kernel took
  // all 512 and still spilled 26 words that every stream-step re-loaded (8.3 us per step where H = 1536 takes 4.6).  The lo
  // plane of the share now lives in LDS (128 KB: [wave][k-step][tile][lane] x 16 bytes, a lane reads back what it wrote --
  // consecutive lanes, consecutive 16-byte words: conflict-free), the hi plane stays in registers.
  constexpr bool WL_LDS = KS == 16 && !F16;
  u32x4* wlds = reinterpret_cast<u32x4*>(smem + RED_FLOATS);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  // (round 6) batch groups side by side: where two groups' workgroups fit the CUs together (H <= 512 bidirectional, H <= 1024
  // unidirectional) a launch of 33 .. 64 rows runs them at the same time instead of one after the other -- the wide
  // kernel's group index, here for the 8-unit kernel.  Group g: rows n_base + 32 g .., exchange region g (of ndir directions).
  const int grp = p.grp_wgs ? (int)blockIdx.x / p.grp_wgs : 0;
  const int bx = p.grp_wgs ? (int)blockIdx.x % p.grp_wgs : (int)blockIdx.x;
  const int d = p.d_base + bx / p.J, j = bx % p.J;
  const int nl = (tid >> 3) & 15, u = tid & 7;  // cell threads are waves 0 and 1
  const int unit = 8 * j + u;
  const int n_base = p.n_base + 32 * grp;
  const int N = min(32, p.N - 32 * grp);

  // this wave's share of W_hh (its K-quarter x 32 gate rows, hi and lo planes) stays in 128 VGPRs
  // per lane for the whole sequence: the MFMA B operand never touches LDS or HBM again
  u32x4 wh0[KS], wh1[KS], wl0[KS], wl1[KS];
  {
    const char* wsrc = reinterpret_cast<const char*>(p.whh) + ((size_t)d * p.J + j) * 128 * H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kg = wave * (KG / 4) + 4 * ks + q;
      const char* wp = wsrc + (kg * 32 + c16) * 16;
      wh0[ks] = *reinterpret_cast<const u32x4*>(wp);
      wh1[ks] = *reinterpret_cast<const u32x4*>(wp + 256);
      if (!F16) {
        if (WL_LDS) {
          wlds[((wave * KS + ks) * 2 + 0) * 64 + lane] = *reinterpret_cast<const u32x4*>(wp + KG * 512);
          wlds[((wave * KS + ks) * 2 + 1) * 64 + lane] = *reinterpret_cast<const u32x4*>(wp + KG * 512 + 256);
        } else {
          wl0[ks] = *reinterpret_cast<const u32x4*>(wp + KG * 512);
          wl1[ks] = *reinterpret_cast<const u32x4*>(wp + KG * 512 + 256);
        }
      }
    }
  }

  const int rs = p.ring_shift;
  const int PLANE = (KG * 256) << rs;       // bytes: [slot][kg][16][8 bf16]
  const int STREAM = 2 * PLANE;
  char* hx_d = reinterpret_cast<char*>(p.hx) + ((size_t)grp * p.ndir + d) * 2 * STREAM;
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx_d, 0, 2 * STREAM, 0x00020000);

  float c[2] = {0.f, 0.f}, h[2] = {0.f, 0.f};
  int len_n[2] = {0, 0};
  if (wave < 2) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      const bool valid = n < N;
      const size_t sidx = ((size_t)d * p.N_total + n_base + n) * H + unit;
      h[sg] = (valid && p.h0) ? p.h0[sidx] : 0.f;
      c[sg] = (valid && p.c0) ? p.c0[sidx] : 0.f;
      len_n[sg] = valid ? (p.lens ? p.lens[n_base + n] : p.steps) : 0;
      const int off = sg * STREAM + epoch_par0(d, p.steps, rs) * KG * 256 + j * 256 + nl * 16;  // slot read by the first step
      publish_split<P>(h[sg], epoch_tag0(d, p.steps, rs), hx_rsrc, off, PLANE + off, lane);
    }
  }
  __syncthreads();

  const int kg_base = wave * (KG / 4);
  const int xcols = p.ndir * 4 * H;
  bool alive = true;
  unsigned long long st_sum[4] = {0, 0, 0, 0}, st_prev = 0;
  if (STAMP) st_prev = wall_clock64();

  for (int s = 0; s < p.steps; ++s) {
    const int t = d ? (p.steps - 1 - s) : s;
    const EpochClock ec = epoch_clock(d, s, p.steps, rs);
    const int par = ec.par;
    const unsigned em = ec.em;
    const unsigned wtag = ec.wtag;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      float xg[4] = {0.f, 0.f, 0.f, 0.f};
      const int n = sg * 16 + nl;
      if (wave < 2 && n < N) {
        const float* xp = p.xproj + ((size_t)t * p.N_total + n_base + n) * xcols + d * 4 * H + j * 32 + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = xp[g * 8];
      }

      // ---- h_{t-1} of this stream: all 8 k-steps of the wave's K-quarter at once
      const int base = sg * STREAM + par * KG * 256 + c16 * 16;
      u32x4 ah[KS], al[KS];
      const unsigned long long t_wait0 = wall_clock64();
      unsigned spins = 0;
      // The first request is issued in straight-line code, the loop only re-requests.  With the loads at the loop header
      // the compiler must assume a previous iteration's load into the same registers is still in flight and emits
      // `s_waitcnt vmcnt(1)` there -- which on the first pass waits for the x-projection loads above (HBM latency) before
      // the h loads are even issued.
      auto request = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kg = kg_base + 4 * ks + q;
          ah[ks] = load_sc1_u128(hx_rsrc, base + kg * 256);
          if (!F16) al[ks] = load_sc1_u128(hx_rsrc, PLANE + base + kg * 256);
        }
      };
      {
        // first request: agent scope (sc1) is what cross-XCD visibility needs and measured 1-2 % faster than the
        // sc0 sc1 + volatile form, which the re-requests keep (a stale answer fails the tag check like any other)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kg = kg_base + 4 * ks + q;
          ah[ks] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, base + kg * 256, 0, /*aux: sc1*/ 16);
          if (!F16) al[ks] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, PLANE + base + kg * 256, 0, /*aux: sc1*/ 16);
        }
      }
      for (;;) {
        unsigned bad = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) bad |= (ah[ks][e] ^ em) | (F16 ? 0u : (al[ks][e] ^ em));
        if (!alive || !__any((bad & 0x00010001u) != 0)) break;
        if ((++spins & 63u) == 0) {
          const unsigned dead = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (dead != 0 || wall_clock64() - t_wait0 > SPIN_LIMIT_TICKS) {
            if (lane == 0) flag_timeout(p.status);
            alive = false;
            break;
          }
        }
        for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
        request();
      }
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[0] += now - st_prev; st_prev = now; }

      f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (F16) {
          const f16x8v xh = __builtin_bit_cast(f16x8v, ah[ks]);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, __builtin_bit_cast(f16x8v, wh0[ks]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, __builtin_bit_cast(f16x8v, wh1[ks]), acc1, 0, 0, 0);
        } else {
          const u32x4 bh0 = wh0[ks], bh1 = wh1[ks];
          const u32x4 bl0 = WL_LDS ? wlds[((wave * KS + ks) * 2 + 0) * 64 + lane] : wl0[ks];
          const u32x4 bl1 = WL_LDS ? wlds[((wave * KS + ks) * 2 + 1) * 64 + lane] : wl1[ks];
          const u32x4 xh = ah[ks], xl = al[ks];
          acc0 = ms::mfma_16x16x32<HM>(xh, bh0, acc0);
          acc1 = ms::mfma_16x16x32<HM>(xh, bh1, acc1);
          acc0 = ms::mfma_16x16x32<HM>(xl, bh0, acc0);
          acc1 = ms::mfma_16x16x32<HM>(xl, bh1, acc1);
          acc0 = ms::mfma_16x16x32<HM>(xh, bl0, acc0);
          acc1 = ms::mfma_16x16x32<HM>(xh, bl1, acc1);
        }
      }
      if (STAMP) st_sum[1] += spins;   // slot 1: failed tag checks (repeated requests) of this stream-step, not a time

      // ---- reduce the 4 K-quarters, cell update on waves 0/1, publish.  `red` is double-buffered
      // by stream, so the only barrier is write -> read (the readers of this buffer two
      // stream-steps ago have since passed the other stream's barrier).
      float* redb = red + sg * RED2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        redb[(wave * 16 + 4 * q + i) * RED_STRIDE + c16] = acc0[i];
        redb[(wave * 16 + 4 * q + i) * RED_STRIDE + 16 + c16] = acc1[i];
      }
      __syncthreads();
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[2] += now - st_prev; st_prev = now; }
      if (wave < 2) {
        float gsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = xg[g];
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) v += redb[(w2 * 16 + nl) * RED_STRIDE + g * 8 + u];
          gsum[g] = v;
        }
        float cnew, hnew;
        if (HARD) {
          cnew = clamp01(0.2f * gsum[1] + 0.5f) * c[sg] + clamp01(0.2f * gsum[0] + 0.5f) * clamp11(gsum[2]);
          hnew = clamp01(0.2f * gsum[3] + 0.5f) * clamp11(cnew);
        } else {
          cnew = fast_sigmoid(gsum[1]) * c[sg] + fast_sigmoid(gsum[0]) * fast_tanh(gsum[2]);
          hnew = fast_sigmoid(gsum[3]) * fast_tanh(cnew);
        }
        const bool active = t < len_n[sg];
        c[sg] = active ? cnew : c[sg];
        h[sg] = active ? hnew : h[sg];
        const int off = sg * STREAM + ec.wpar * KG * 256 + j * 256 + nl * 16;
        publish_split<P>(h[sg], wtag, hx_rsrc, off, PLANE + off, lane);
        if (n < N) {
          const size_t oidx = ((size_t)t * p.N_total + n_base + n) * (p.ndir * H) + d * H + unit;
          const float ov = active ? hnew : 0.f;
          if (p.out_hi) {   // exactly split_planes_kernel's arithmetic
            if (F16) {
              p.out_hi[oidx] = __builtin_bit_cast(unsigned short, (_Float16)ov);
            } else {
              unsigned hb, lb;
              ms::plane_split_bounded<HM>(ov, hb, lb);
              p.out_hi[oidx] = (unsigned short)hb;
              p.out_lo[oidx] = (unsigned short)lb;
            }
          } else {
            p.out[oidx] = ov;
          }
        }
      }
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[3] += now - st_prev; st_prev = now; }
    }
  }
  if (STAMP && (tid & 63) == 0) {
    // slots 0-3: the cell waves (0, 1); slots 4-7: waves 2, 3
    for (int k = 0; k < 4; ++k) atomicAdd(&p.dbg[(size_t)blockIdx.x * 8 + (wave >= 2 ? 4 : 0) + k], st_sum[k]);
  }
  if (wave < 2) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      if (n < N) {
        const size_t sidx = ((size_t)d * p.N_total + n_base + n) * H + unit;
        p.hn[sidx] = h[sg];
        p.cn[sidx] = c[sg];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ persistent LSTM, split-bf16, two streams, wide workgroups
//
// Round 3.  What binds lstm_persistent_split2_kernel is the delivery of h out of the XCDs' L2s: every one of the 256
// workgroups pulls the whole 64 KB of its direction's h per stream-step, for the 8 hidden units it owns (section 4 of
// DESIGN.md, "Round 3").  This form fuses two unit blocks into one workgroup of 8 waves (16 units, 64 gate rows, two
// waves per SIMD): a wave contracts one EIGHTH of K (4 k-steps) for all 64 gate rows, so the same 64 KB feed twice the
// arithmetic and a batch group of 32 rows needs 128 workgroups instead of 256.  The other half of the chip takes a SECOND
// batch group of 32 rows in the same launch (blockIdx -> group): two batches side by side for the L2 traffic of one.
// Same packed weights, same exchange layout and tags, same cell; the K split into eighths changes the order of the
// float32 partial sums, so results agree with the quarter-split kernel to rounding, not bit for bit.
// H = 1024 only (KS = 4 k-steps per wave), LSTM cell, bf16x3 operands.
constexpr int WIDE_RED_STRIDE = 72;                        // floats per reduction row: 64 gate columns + 8 (bank spread)
constexpr int WIDE_RED2 = 8 * 16 * WIDE_RED_STRIDE;        // 8 waves x 16 rows, per stream

// One wave's part: KS k-steps starting at k-step k0 of the workgroup's 32; CELL: the wave also applies the cell (waves 0-3).
// Uneven shares (waves 0-3 take KSC k-steps each, waves 4-7 the other 8 - KSC): the cell waves spend ~0.5 us of every
// stream-step on the cell and the publish before they request their own operands, waves 4-7 go straight from the barrier
// to the next request and share the SIMDs' matrix pipes with them.
// F16 (MS_PRECISION=fp16): one fp16 plane of W_hh and of h (the lo planes are neither loaded nor published), one MFMA pass.
template <int KS, bool CELL, bool HARD, bool STAMP = false, bool PACKED = false, int P = ms::PREC_BF16X3>
__device__ __forceinline__ void wide2_wave(const LstmP& p, float* red, const int k0, const int32_t* __restrict__ row_off) {
  constexpr bool F16 = ms::prec_one_plane(P), HM = P == ms::PREC_F16X3;
  constexpr int H = 1024, KG = H / 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int JJ = p.J / 2;                                  // workgroups per direction and group
  int grp = blockIdx.x / (p.ndir * JJ);
  const int rem = blockIdx.x % (p.ndir * JJ);
  int d = rem / JJ, jj = rem % JJ;
  if (p.xcd_map) {
    // experiment (MS_LSTM_WIDE_XCD=1; slower, see the launch site): two groups x two directions, each (group, direction) on
    // its own pair of XCDs (workgroups are dealt round-robin over the 8 XCDs -- observed, only speed depends on it)
    const int xcd = blockIdx.x & 7;
    grp = xcd >> 2;
    d = (xcd >> 1) & 1;
    jj = (blockIdx.x >> 3) * 2 + (xcd & 1);
  }
  const int n_base = p.n_base + 32 * grp;
  const int N = min(32, p.N - 32 * grp);                   // rows of this group (the host launches only groups with rows)
  // cell threads: waves 0-3; unit block b (0 / 1), batch row nl, unit u of the block
  const int cb = (tid >> 7) & 1, nl = (tid >> 3) & 15, u = tid & 7;
  const int jb = 2 * jj + cb;                              // 8-unit block of this cell thread
  const int unit = 8 * jb + u;

  // this wave's share of W_hh: its K-eighth x 64 gate rows (two blocks x two column tiles), hi and lo planes, 128 VGPRs
  u32x4 wh[2][2][KS], wl[2][2][F16 ? 1 : KS];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const char* wsrc = reinterpret_cast<const char*>(p.whh) + ((size_t)d * p.J + 2 * jj + b) * 128 * H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kg = 4 * (k0 + ks) + q;
      const char* wp = wsrc + (kg * 32 + c16) * 16;
      wh[b][0][ks] = *reinterpret_cast<const u32x4*>(wp);
      wh[b][1][ks] = *reinterpret_cast<const u32x4*>(wp + 256);
      if (!F16) {
        wl[b][0][ks] = *reinterpret_cast<const u32x4*>(wp + KG * 512);
        wl[b][1][ks] = *reinterpret_cast<const u32x4*>(wp + KG * 512 + 256);
      }
    }
  }

  const int rs = p.ring_shift;
  const int PLANE = (KG * 256) << rs;       // bytes: [slot][kg][16][8 bf16]
  const int STREAM = 2 * PLANE;
  char* hx_d = reinterpret_cast<char*>(p.hx) + ((size_t)grp * p.ndir + d) * 2 * STREAM;
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx_d, 0, 2 * STREAM, 0x00020000);

  float c[2] = {0.f, 0.f}, h[2] = {0.f, 0.f};
  int len_n[2] = {0, 0};
  if (CELL) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      const bool valid = n < N;
      const size_t sidx = ((size_t)d * p.N_total + n_base + n) * H + unit;
      h[sg] = (valid && p.h0) ? p.h0[sidx] : 0.f;
      c[sg] = (valid && p.c0) ? p.c0[sidx] : 0.f;
      len_n[sg] = valid ? (p.lens ? p.lens[n_base + n] : p.steps) : 0;
      // slot and tag the first step of this launch reads (s_begin = 0: epoch_par0 / epoch_tag0; a continuing segment re-publishes
      // what its predecessor's last step published there -- the same float32 h, the same bits)
      const EpochClock ec0 = epoch_clock(d, p.s_begin, p.steps, rs);
      const int off = sg * STREAM + ec0.par * KG * 256 + jb * 256 + nl * 16;
      publish_split<P>(h[sg], ec0.em & 1u, hx_rsrc, off, PLANE + off, lane, valid);
    }
  }
  __syncthreads();

  const int xcols = p.ndir * 4 * H;
  const int lane_off = q * 256 + c16 * 16;
  // Rows past the group's last sequence are padding of the 16-row MFMA tile: their lanes request an offset beyond the
  // buffer (a raw buffer load then returns zeros without touching memory) and take no part in the tag check, and nobody
  // publishes them: a single clip pulls 4 KB of h per stream-step instead of 64 KB (round 4; a full group is unchanged).
  const bool row_ok[2] = {c16 < N, 16 + c16 < N};
  const int loff[2] = {row_ok[0] ? lane_off : 0x40000000, row_ok[1] ? lane_off : 0x40000000};
  bool alive = true;
  // diagnostic build: [0] wait for h (first request .. tags fresh), [1] repeated requests (a count), [2] MFMAs + partial sums
  // to LDS, [3] wait at the barrier, [4] reduce + cell + publish (cell waves); summed per wave class over the launch
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_prev = 0;
  if (STAMP) st_prev = wall_clock64();

  // PACKED: first row of the step's frame in xproj / the planes; the next step's is fetched a step ahead
  // (a kernel argument of its own, const and restrict: the compiler may then fetch it with scalar loads, outside the vector
  // memory queue whose order the tag checks count on)
  int roff_next = (PACKED && CELL) ? row_off[d ? p.steps - 1 - p.s_begin : p.s_begin] : 0;
  for (int s = p.s_begin; s < p.s_end; ++s) {
    const int t = d ? (p.steps - 1 - s) : s;
    const int roff = roff_next;
    if (PACKED && CELL) roff_next = row_off[s + 1 < p.steps ? (d ? t - 1 : t + 1) : t];
    const EpochClock ec = epoch_clock(d, s, p.steps, rs);
    const int par = ec.par;
    const unsigned em = ec.em;
    const unsigned wtag = ec.wtag;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      float xg[4] = {0.f, 0.f, 0.f, 0.f};
      const int n = sg * 16 + nl;
      if (CELL && n < N) {
        // (PACKED: a row past its sequence's end does not exist; its thread reads the frame's first row instead -- every
        // frame < steps has one -- and discards the result as it always did)
        const size_t row = PACKED ? (size_t)(roff + (t < len_n[sg] ? n_base + n : 0)) : (size_t)t * p.N_total + n_base + n;
        const float* xp = p.xproj + row * xcols + d * 4 * H + jb * 32 + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = xp[g * 8];
      }

      // ---- h_{t-1} of this stream: the k-steps of the wave's K share, hi and lo planes (2 KS loads of 1 KB)
      const int base = sg * STREAM + par * KG * 256 + k0 * 1024;
      u32x4 ah[KS], al[F16 ? 1 : KS];
      const unsigned long long t_wait0 = wall_clock64();
      unsigned spins = 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {   // first request: agent scope; the re-requests below: system scope + volatile
        ah[ks] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, loff[sg], base + ks * 1024, /*aux: sc1*/ 16);
        if (!F16) al[ks] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, loff[sg], PLANE + base + ks * 1024, /*aux: sc1*/ 16);
      }
      f32x4v acc[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[b][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
      auto mfma_step = [&](int ks) {
        if (F16) {
          const f16x8v xf = __builtin_bit_cast(f16x8v, ah[ks]);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, __builtin_bit_cast(f16x8v, wh[b][0][ks]), acc[b][0], 0, 0, 0);
            acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, __builtin_bit_cast(f16x8v, wh[b][1][ks]), acc[b][1], 0, 0, 0);
          }
          return;
        }
        const u32x4 xh = ah[ks], xl = al[F16 ? 0 : ks];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const u32x4 bh0 = wh[b][0][ks], bh1 = wh[b][1][ks], bl0 = wl[b][0][F16 ? 0 : ks], bl1 = wl[b][1][F16 ? 0 : ks];
          acc[b][0] = ms::mfma_16x16x32<HM>(xh, bh0, acc[b][0]);
          acc[b][1] = ms::mfma_16x16x32<HM>(xh, bh1, acc[b][1]);
          acc[b][0] = ms::mfma_16x16x32<HM>(xl, bh0, acc[b][0]);
          acc[b][1] = ms::mfma_16x16x32<HM>(xl, bh1, acc[b][1]);
          acc[b][0] = ms::mfma_16x16x32<HM>(xh, bl0, acc[b][0]);
          acc[b][1] = ms::mfma_16x16x32<HM>(xh, bl1, acc[b][1]);
        }
      };
      for (;;) {
        unsigned bad = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) bad |= (ah[ks][e] ^ em) | (F16 ? 0u : (al[F16 ? 0 : ks][e] ^ em));
        if (!alive || !__any(row_ok[sg] && (bad & 0x00010001u) != 0)) break;
        if ((++spins & 63u) == 0) {
          const unsigned dead = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (dead != 0 || wall_clock64() - t_wait0 > SPIN_LIMIT_TICKS) {
            if (lane == 0) flag_timeout(p.status);
            alive = false;
            break;
          }
        }
        for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          ah[ks] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, loff[sg], base + ks * 1024, (int)(0x80000000u | 16u));
          if (!F16) al[ks] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, loff[sg], PLANE + base + ks * 1024, (int)(0x80000000u | 16u));
        }
      }
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[0] += now - st_prev; st_prev = now; st_sum[1] += spins; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) mfma_step(ks);

      // ---- reduce the 8 K-eighths, cell update on waves 0-3, publish.  `red` is double-buffered by stream (see
      // lstm_persistent_split2_kernel): one barrier per stream-step
      float* redb = red + sg * WIDE_RED2;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) redb[(wave * 16 + 4 * q + i) * WIDE_RED_STRIDE + b * 32 + ct * 16 + c16] = acc[b][ct][i];
      if (STAMP) {
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the partial sums are in the LDS
        const unsigned long long now = wall_clock64(); st_sum[2] += now - st_prev; st_prev = now;
      }
      __syncthreads();
      if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[3] += now - st_prev; st_prev = now; }
      if (CELL) {
        float gsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = xg[g];
#pragma unroll
          for (int w2 = 0; w2 < 8; ++w2) v += redb[(w2 * 16 + nl) * WIDE_RED_STRIDE + cb * 32 + g * 8 + u];
          gsum[g] = v;
        }
        float cnew, hnew;
        if (HARD) {
          cnew = clamp01(0.2f * gsum[1] + 0.5f) * c[sg] + clamp01(0.2f * gsum[0] + 0.5f) * clamp11(gsum[2]);
          hnew = clamp01(0.2f * gsum[3] + 0.5f) * clamp11(cnew);
        } else {
          cnew = fast_sigmoid(gsum[1]) * c[sg] + fast_sigmoid(gsum[0]) * fast_tanh(gsum[2]);
          hnew = fast_sigmoid(gsum[3]) * fast_tanh(cnew);
        }
        const bool active = t < len_n[sg];
        c[sg] = active ? cnew : c[sg];
        h[sg] = active ? hnew : h[sg];
        const int off = sg * STREAM + ec.wpar * KG * 256 + jb * 256 + nl * 16;
        publish_split<P>(h[sg], wtag, hx_rsrc, off, PLANE + off, lane, n < N);
        if (n < N) {
          const size_t oidx = ((size_t)t * p.N_total + n_base + n) * (p.ndir * H) + d * H + unit;
          const float ov = active ? hnew : 0.f;
          if (p.out_hi) {   // exactly split_planes_kernel's arithmetic
            // (PACKED: a row that does not exist is written to the planes' last row slot, steps * N - 1, which no packed
            // row of a batch with such a row reaches and nobody reads: a select on the address instead of a branch)
            const size_t prow = active ? (size_t)(roff + n_base + n) : (size_t)p.steps * p.N_total - 1;
            const size_t pidx = PACKED ? (prow * (p.ndir * H) + d * H + unit) : oidx;
            if (F16) {
              p.out_hi[pidx] = __builtin_bit_cast(unsigned short, (_Float16)ov);
            } else {
              unsigned hb, lb;
              ms::plane_split_bounded<HM>(ov, hb, lb);
              p.out_hi[pidx] = (unsigned short)hb;
              p.out_lo[pidx] = (unsigned short)lb;
            }
          } else {
            p.out[oidx] = ov;
          }
        }
        if (STAMP) { const unsigned long long now = wall_clock64(); st_sum[4] += now - st_prev; st_prev = now; }
      }
    }
  }
  if (STAMP && lane == 0) {
    for (int k = 0; k < 5; ++k) atomicAdd(&p.dbg[(size_t)blockIdx.x * 16 + (CELL ? 0 : 8) + k], st_sum[k]);
  }
  if (CELL) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      if (n < N) {
        const size_t sidx = ((size_t)d * p.N_total + n_base + n) * H + unit;
        p.hn[sidx] = h[sg];
        p.cn[sidx] = c[sg];
      }
    }
  }
}

template <bool HARD, int KSC, bool STAMP = false, bool PACKED = false, int P = ms::PREC_BF16X3>
__global__ __launch_bounds__(512, 2) void lstm_persistent_wide2_kernel(LstmP p, const int32_t* __restrict__ row_off) {
  static_assert(KSC >= 1 && KSC <= 7, "both wave sets need at least one k-step");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) wide2_wave<KSC, true, HARD, STAMP, PACKED, P>(p, smem, wave * KSC, row_off);
  else wide2_wave<8 - KSC, false, HARD, STAMP, false, P>(p, smem, 4 * KSC + (wave - 4) * (8 - KSC), row_off);
}


// ------------------------------------------------------------------------------------------------ persistent LSTM, float32 MFMA, two streams
//
// MS_PRECISION=f32 on the two-stream shapes.  The same decomposition as lstm_persistent_split2_kernel -- 8 units per
// workgroup, each wave one K-quarter, two interleaved streams of 16 batch rows -- with every product in float32:
// v_mfma_f32_16x16x4_f32 (16 batch rows x 16 gate columns x 4 k, a k-ordered fmaf chain).  A wave's K-quarter x 32 gate
// columns of W_hh is 32 * H / 4 floats = 128 VGPRs per lane at H = 1024 and stays in registers for the whole sequence
// (the one-stream kernel above re-reads it from LDS every step and waits for epoch flags before it may load h).
// h is exchanged as float32 whose mantissa LSB carries the 1-bit epoch tag -- the only place where this mode is not
// bit-faithful float32: the copy of h that OTHER workgroups multiply is off by at most one ulp (2^-24 relative; each
// workgroup's own state, the outputs and h_n / c_n are untouched).  Full-size config-2 logits: max error 3.9e-8 against
// the reference, 3.4e-8 with the exact exchange of the one-stream kernel (MS_LSTM_F32_ONE_STREAM=1).  An untagged
// alternative was built and measured first: 8-byte {value, step} packets double the bytes every CU pulls per step
// (128 KB per stream-step at ~65 GB/s) and ran 14 % SLOWER than the one-stream kernel; with 4-byte tagged elements
// the step drops from 8.2 to 5.9 us (2.97 vs 4.13 ms per layer, 75 % of the algorithmic roofline).
// Element layout per (direction, stream, slot): [k/16][k%4][row 16][(k%16)/4] floats -- lane (row = lane & 15,
// kq = lane >> 4) finds the A operands of four consecutive MFMAs (k = 16g + 4j + kq, j = 0..3) in one 16-byte load.
// Slots and tags follow the EpochClock above (sequence time, two slots); hx_init_kernel pre-sets every word of a slot to
// the tag that is NOT the first one expected there.

// dst[d][j][wave][g][c][lane][e] = w_hh[gate*H + 8j + u][wave*H/4 + 16g + 4e + (lane >> 4)],  gate*8 + u = 16c + (lane & 15)
__global__ void pack_whh_f32x2_kernel(const float* __restrict__ w, float* __restrict__ dst, int H) {
  const size_t total = (size_t)4 * H * H;
  const int G = H / 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 3, lane = (i >> 2) & 63, c = (i >> 8) & 1;
    const size_t rest = i >> 9;
    const int g = rest % G, wave = (rest / G) & 3;
    const int j = (int)(rest / G / 4);
    const int r = 16 * c + (lane & 15), gate = r >> 3, u = r & 7;
    const int k = wave * (H / 4) + 16 * g + 4 * e + (lane >> 4);
    dst[i] = w[(size_t)(gate * H + 8 * j + u) * H + k];
  }
}

template <int G, bool HARD>
__global__ __launch_bounds__(256, 1) void lstm_persistent_f32x2_kernel(LstmP p) {
  constexpr int H = 64 * G;
  constexpr int RED2 = 4 * 16 * RED_STRIDE;
  constexpr int SLOT = H * 16 * 4;   // bytes of one (stream, slot): H x 16 floats
  constexpr int STREAM = 2 * SLOT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;  // [stream][4 waves][16 rows][RED_STRIDE]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int d = blockIdx.x / p.J, j = blockIdx.x % p.J;
  const int nl = (tid >> 3) & 15, u = tid & 7;  // cell threads are waves 0 and 1
  const int unit = 8 * j + u;
  const int N = p.N;

  f32x4 w0[G], w1[G];
  {
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.whh) + ((((size_t)d * p.J + j) * 4 + wave) * G * 2) * 64 + lane;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      w0[g] = wsrc[(g * 2 + 0) * 64];
      w1[g] = wsrc[(g * 2 + 1) * 64];
    }
  }

  char* hx_d = reinterpret_cast<char*>(p.hx) + (size_t)d * 2 * STREAM;
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx_d, 0, 2 * STREAM, 0x00020000);
  const int pk_off = (((unit >> 4) * 4 + (unit & 3)) * 16 + nl) * 16 + ((unit >> 2) & 3) * 4;  // this thread's element
  auto publish = [&](float hval, unsigned tag, int off) {
    __builtin_amdgcn_raw_buffer_store_b32((__float_as_uint(hval) & ~1u) | tag, hx_rsrc, off, 0, /*aux: sc1*/ 16);
  };

  float c[2] = {0.f, 0.f}, h[2] = {0.f, 0.f};
  int len_n[2] = {0, 0};
  if (wave < 2) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      const bool valid = n < N;
      const size_t sidx = ((size_t)d * p.N_total + p.n_base + n) * H + unit;
      h[sg] = (valid && p.h0) ? p.h0[sidx] : 0.f;
      c[sg] = (valid && p.c0) ? p.c0[sidx] : 0.f;
      len_n[sg] = valid ? (p.lens ? p.lens[p.n_base + n] : p.steps) : 0;
      publish(h[sg], epoch_tag0(d, p.steps), sg * STREAM + epoch_par0(d, p.steps) * SLOT + pk_off);   // read by the first step
    }
  }
  __syncthreads();

  const int xcols = p.ndir * 4 * H;
  bool alive = true;

  for (int s = 0; s < p.steps; ++s) {
    const int t = d ? (p.steps - 1 - s) : s;
    // the clock of the tags is the sequence time (see EpochClock): a sequence's outputs do not depend on the batch's length
    const EpochClock ec = epoch_clock(d, s, p.steps);
    const unsigned em = (ec.em & 1u) ? 0xFFFFFFFFu : 0u;      // LSB of every element this step reads
    const unsigned wtag = ec.wtag;
    const int rslot = ec.par * SLOT, wslot = ec.wpar * SLOT;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      float xg[4] = {0.f, 0.f, 0.f, 0.f};
      const int n = sg * 16 + nl;
      if (wave < 2 && n < N) {
        const float* xp = p.xproj + ((size_t)t * p.N_total + p.n_base + n) * xcols + d * 4 * H + j * 32 + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = xp[g * 8];
      }

      // ---- h_{t-1} of this stream: all G k-groups (16 k each) of the wave's K-quarter requested at once (8 G VGPRs);
      // the first request in straight-line code, the loop only re-requests (see lstm_persistent_split2_kernel)
      const int base = sg * STREAM + rslot + (wave * G * 4 * 16 + q * 16 + c16) * 16;
      u32x4 pa[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        pa[g] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, base + g * 1024, 0, /*aux: sc1*/ 16);
      }
      {
        const unsigned long long t_wait0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
          unsigned bad = 0;
#pragma unroll
          for (int g = 0; g < G; ++g) bad |= (pa[g][0] ^ em) | (pa[g][1] ^ em) | (pa[g][2] ^ em) | (pa[g][3] ^ em);
          if (!alive || !__any((bad & 1u) != 0)) break;
          if ((++spins & 63u) == 0) {
            const unsigned dead = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dead != 0 || wall_clock64() - t_wait0 > SPIN_LIMIT_TICKS) {
              if (lane == 0) flag_timeout(p.status);
              alive = false;
              break;
            }
          }
          for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            pa[g] = load_sc1_u128(hx_rsrc, base + g * 1024);
          }
        }
      }
      f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float a0 = __uint_as_float(pa[g][0]), a1 = __uint_as_float(pa[g][1]);
        const float a2 = __uint_as_float(pa[g][2]), a3 = __uint_as_float(pa[g][3]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w0[g][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w1[g][0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w0[g][1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w1[g][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w0[g][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w1[g][2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w0[g][3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w1[g][3], acc1, 0, 0, 0);
      }

      // ---- reduce the 4 K-quarters (LDS, double-buffered by stream: one barrier), cell update on waves 0/1, publish
      float* redb = red + sg * RED2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        redb[(wave * 16 + 4 * q + i) * RED_STRIDE + c16] = acc0[i];
        redb[(wave * 16 + 4 * q + i) * RED_STRIDE + 16 + c16] = acc1[i];
      }
      __syncthreads();
      if (wave < 2) {
        float gsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = xg[g];
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) v += redb[(w2 * 16 + nl) * RED_STRIDE + g * 8 + u];
          gsum[g] = v;
        }
        float cnew, hnew;
        if (HARD) {
          cnew = clamp01(0.2f * gsum[1] + 0.5f) * c[sg] + clamp01(0.2f * gsum[0] + 0.5f) * clamp11(gsum[2]);
          hnew = clamp01(0.2f * gsum[3] + 0.5f) * clamp11(cnew);
        } else {
          cnew = fast_sigmoid(gsum[1]) * c[sg] + fast_sigmoid(gsum[0]) * fast_tanh(gsum[2]);
          hnew = fast_sigmoid(gsum[3]) * fast_tanh(cnew);
        }
        const bool active = t < len_n[sg];
        c[sg] = active ? cnew : c[sg];
        h[sg] = active ? hnew : h[sg];
        publish(h[sg], wtag, sg * STREAM + wslot + pk_off);
        if (n < N) p.out[((size_t)t * p.N_total + p.n_base + n) * (p.ndir * H) + d * H + unit] = active ? hnew : 0.f;
      }
    }
  }
  if (wave < 2) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      if (n < N) {
        const size_t sidx = ((size_t)d * p.N_total + p.n_base + n) * H + unit;
        p.hn[sidx] = h[sg];
        p.cn[sidx] = c[sg];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ persistent GRU, split-bf16
//
// The reference's SHIPPED DS2 config is 3 x GRU-2560: 78.6 MB of f32 recurrent weights per layer, far beyond the L2, so
// a per-step launch re-streams them from the Infinity Cache every step (rnn_step_mfma_kernel: 35 us / step).  With
// U = 10 units per workgroup the layer is H / 10 = 256 workgroups = one per CU, and a workgroup's slice -- 30 gate rows
// (r, z, n of its 10 units, padded to 32) x H, bf16 hi + lo = 307 KB at H = 2560 -- fits the CU's register file:
// each wave keeps its K-quarter x 32 rows x 2 planes in 16*KS VGPRs per lane for the whole sequence, exactly like
// lstm_persistent_split2_kernel, and the per-step traffic shrinks to the exchange of h (two streams of 16 batch rows,
// bf16 hi/lo planes, a 1-bit epoch tag in every element, sc1 stores / sc0 sc1 polls).  Differences from the LSTM
// kernel: the h operands of a stream-step (KS k-steps) do not fit beside the weights, so they move through two
// rotating register chunks of 5 k-steps; a producer's 10 units straddle the 8-element operand granules, so every
// (row, unit) element is published on its own (2-byte sc1 stores; the per-element tags make torn granules harmless).
struct GruP {
  const float* xproj;          // [steps*N_total][ndir*3H], natural column order (r, z, n), b_ih folded in
  const unsigned short* whh;   // [ndir][J][plane hi|lo][H/8][32 rows: g*U + u, 30 used][8 bf16]
  const float* bhh;            // [ndir][3H]
  const int32_t* lens;         // may be null
  const float* h0;             // [ndir][N_total][H] or null
  float* out;                  // [T][N_total][ndir*H]
  float* hn;                   // [ndir][N_total][H]
  float* hx;                   // exchange buffer, per direction [stream][plane][parity][H/8][16][8 bf16]
  unsigned* status;
  int steps, N, n_base, N_total, ndir, J, poll_sleep, ring_shift;
  int d_base;                  // direction of workgroup 0 (a bidirectional layer whose directions run as two launches)
  int grp_wgs;                 // workgroups per batch group of <= 32 rows when one launch holds two groups side by side (0: one group)
  unsigned short* out_hi;      // when set: the next layer's GEMM operand planes instead of float32 `out` (see LstmP)
  unsigned short* out_lo;
};

template <bool HM>
__device__ __forceinline__ void publish_elem(float hval, unsigned tag, __amdgpu_buffer_rsrc_t rsrc, int off_hi, int off_lo) {
  const unsigned hi = (ms::plane_bits_bounded<HM>(hval) & 0xFFFEu) | tag;
  const unsigned lo = (ms::plane_bits_bounded<HM>(hval - ms::plane_val<HM>(hi)) & 0xFFFEu) | tag;
  __builtin_amdgcn_raw_buffer_store_b16((unsigned short)hi, rsrc, off_hi, 0, /*aux: sc1*/ 16);
  __builtin_amdgcn_raw_buffer_store_b16((unsigned short)lo, rsrc, off_lo, 0, /*aux: sc1*/ 16);
}

template <int KS, int U, bool HM = false>
__global__ __launch_bounds__(256, 1) void gru_persistent_kernel(GruP p) {
  constexpr int H = 128 * KS, KG = H / 8, CH = KS % 5 == 0 ? 5 : KS % 4 == 0 ? 4 : 3, NCH = KS / CH;
  static_assert(KS % CH == 0 && 3 * U <= 32 && 16 * U <= 256 && H % U == 0, "unsupported GRU tiling");
  constexpr int RED2 = 4 * 16 * RED_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;  // [stream][4 waves][16 rows][RED_STRIDE]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  // (batch groups side by side: see lstm_persistent_split2_kernel)
  const int grp = p.grp_wgs ? (int)blockIdx.x / p.grp_wgs : 0;
  const int bx = p.grp_wgs ? (int)blockIdx.x % p.grp_wgs : (int)blockIdx.x;
  const int d = p.d_base + bx / p.J, j = bx % p.J;
  const int n_base = p.n_base + 32 * grp;
  const bool cell_thread = tid < 16 * U;
  const int nl = cell_thread ? tid / U : 0, u = cell_thread ? tid % U : 0;
  const int unit = U * j + u;
  const int N = min(32, p.N - 32 * grp);

  // this wave's K-quarter of the workgroup's 32 packed gate rows (hi and lo planes): 16*KS VGPRs per lane
  u32x4 wh0[KS], wh1[KS], wl0[KS], wl1[KS];
  {
    const char* wsrc = reinterpret_cast<const char*>(p.whh) + ((size_t)d * p.J + j) * 128 * H;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kg = wave * (KG / 4) + 4 * ks + q;
      const char* wp = wsrc + (kg * 32 + c16) * 16;
      wh0[ks] = *reinterpret_cast<const u32x4*>(wp);
      wh1[ks] = *reinterpret_cast<const u32x4*>(wp + 256);
      wl0[ks] = *reinterpret_cast<const u32x4*>(wp + KG * 512);
      wl1[ks] = *reinterpret_cast<const u32x4*>(wp + KG * 512 + 256);
    }
  }

  const int rs = p.ring_shift;
  const int PLANE = (KG * 256) << rs;       // bytes: [slot][kg][16][8 bf16]
  const int STREAM = 2 * PLANE;
  char* hx_d = reinterpret_cast<char*>(p.hx) + ((size_t)grp * p.ndir + d) * 2 * STREAM;
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx_d, 0, 2 * STREAM, 0x00020000);
  const int elem_off = ((unit >> 3) * 16 + nl) * 16 + (unit & 7) * 2;   // this thread's element inside a parity slab

  float h[2] = {0.f, 0.f};
  int len_n[2] = {0, 0};
  float bh[3] = {0.f, 0.f, 0.f};
  if (cell_thread) {
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = p.bhh[(size_t)d * 3 * H + g * H + unit];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      const bool valid = n < N;
      const size_t sidx = ((size_t)d * p.N_total + n_base + n) * H + unit;
      h[sg] = (valid && p.h0) ? p.h0[sidx] : 0.f;
      len_n[sg] = valid ? (p.lens ? p.lens[n_base + n] : p.steps) : 0;
      const int off = sg * STREAM + epoch_par0(d, p.steps, rs) * KG * 256 + elem_off;  // slot read by the first step
      publish_elem<HM>(h[sg], epoch_tag0(d, p.steps, rs), hx_rsrc, off, PLANE + off);
    }
  }
  __syncthreads();

  const int kg_base = wave * (KG / 4);
  const int xcols = p.ndir * 3 * H;
  bool alive = true;

  for (int s = 0; s < p.steps; ++s) {
    const int t = d ? (p.steps - 1 - s) : s;
    const EpochClock ec = epoch_clock(d, s, p.steps, rs);
    const int par = ec.par;
    const unsigned em = ec.em;
    const unsigned wtag = ec.wtag;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      float xg[3] = {0.f, 0.f, 0.f};
      const int n = sg * 16 + nl;
      if (cell_thread && n < N) {
        const float* xp = p.xproj + ((size_t)t * p.N_total + n_base + n) * xcols + d * 3 * H + unit;
#pragma unroll
        for (int g = 0; g < 3; ++g) xg[g] = xp[g * H];
      }
      const int base = sg * STREAM + par * KG * 256 + c16 * 16;
      u32x4 ah[2][CH], al[2][CH];
      auto issue = [&](int buf, int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int kg = kg_base + 4 * (c * CH + i) + q;
          ah[buf][i] = load_sc1_u128(hx_rsrc, base + kg * 256);
          al[buf][i] = load_sc1_u128(hx_rsrc, PLANE + base + kg * 256);
        }
      };
      f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      issue(0, 0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        if (c + 1 < NCH) issue(buf ^ 1, c + 1);
        // wait until every element of this chunk carries the step's tag (stale or torn -> load the chunk again)
        const unsigned long long t_wait0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
          unsigned bad = 0;
#pragma unroll
          for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= (ah[buf][i][e] ^ em) | (al[buf][i][e] ^ em);
          if (!alive || !__any((bad & 0x00010001u) != 0)) break;
          if ((++spins & 63u) == 0) {
            const unsigned dead = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dead != 0 || wall_clock64() - t_wait0 > SPIN_LIMIT_TICKS) {
              if (lane == 0) flag_timeout(p.status);
              alive = false;
              break;
            }
          }
          for (int z = 0; z < p.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
          issue(buf, c);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int ks = c * CH + i;
          const u32x4 bh0 = wh0[ks], bh1 = wh1[ks], bl0 = wl0[ks], bl1 = wl1[ks];
          const u32x4 xh = ah[buf][i], xl = al[buf][i];
          acc0 = ms::mfma_16x16x32<HM>(xh, bh0, acc0);
          acc1 = ms::mfma_16x16x32<HM>(xh, bh1, acc1);
          acc0 = ms::mfma_16x16x32<HM>(xl, bh0, acc0);
          acc1 = ms::mfma_16x16x32<HM>(xl, bh1, acc1);
          acc0 = ms::mfma_16x16x32<HM>(xh, bl0, acc0);
          acc1 = ms::mfma_16x16x32<HM>(xh, bl1, acc1);
        }
      }

      // reduce the 4 K-quarters (LDS, double-buffered by stream: one barrier), cell update, publish
      float* redb = red + sg * RED2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        redb[(wave * 16 + 4 * q + i) * RED_STRIDE + c16] = acc0[i];
        redb[(wave * 16 + 4 * q + i) * RED_STRIDE + 16 + c16] = acc1[i];
      }
      __syncthreads();
      if (cell_thread) {
        float hs[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          float v = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) v += redb[(w2 * 16 + nl) * RED_STRIDE + g * U + u];
          hs[g] = v + bh[g];
        }
        const float r = fast_sigmoid(xg[0] + hs[0]);
        const float z = fast_sigmoid(xg[1] + hs[1]);
        const float nn = fast_tanh(xg[2] + r * hs[2]);
        const float hnew = (1.0f - z) * nn + z * h[sg];
        const bool active = t < len_n[sg];
        h[sg] = active ? hnew : h[sg];
        const int off = sg * STREAM + ec.wpar * KG * 256 + elem_off;
        publish_elem<HM>(h[sg], wtag, hx_rsrc, off, PLANE + off);
        if (n < N) {
          const size_t oidx = ((size_t)t * p.N_total + n_base + n) * (p.ndir * H) + d * H + unit;
          const float ov = active ? hnew : 0.f;
          if (p.out_hi) {   // exactly split_planes_kernel's arithmetic
            unsigned hb, lb;
            ms::plane_split_bounded<HM>(ov, hb, lb);
            p.out_hi[oidx] = (unsigned short)hb;
            p.out_lo[oidx] = (unsigned short)lb;
          } else {
            p.out[oidx] = ov;
          }
        }
      }
    }
  }
  if (cell_thread) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int n = sg * 16 + nl;
      if (n < N) p.hn[((size_t)d * p.N_total + n_base + n) * H + unit] = h[sg];
    }
  }
}

// [d][j][plane][kg][32 rows][8]: row r < 3U is gate r / U of unit U*j + r % U, rows 3U..31 are zero
template <bool HM>
__global__ void pack_whh_gru_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int H, int U) {
  const int KG = H / 8, J = H / U;
  const size_t total = (size_t)J * KG * 256;  // elements per plane
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 7;
    const int r = (i >> 3) & 31;
    const int kg = (i >> 8) % KG;
    const int j = (i >> 8) / KG;
    float x = 0.f;
    if (r < 3 * U) x = w[(size_t)((r / U) * H + U * j + r % U) * H + 8 * kg + e];
    const size_t base = (size_t)j * 2 * KG * 256 + ((size_t)kg * 32 + r) * 8 + e;
    unsigned hb, lb;
    ms::plane_split<HM>(x, hb, lb);
    dst[base] = (unsigned short)hb;
    dst[base + (size_t)KG * 256] = (unsigned short)lb;
  }
}

}  // namespace

// ================================================================================================ launch timing

// ================================================================================================ residency
namespace {
// MS_RNN_FAKE_OCCUPANCY=<n> (tests): pretend the occupancy API answered n for every persistent kernel.
int fake_occupancy() {
  static const int v = getenv("MS_RNN_FAKE_OCCUPANCY") ? atoi(getenv("MS_RNN_FAKE_OCCUPANCY")) : -1;
  return v;
}

struct PersistentKernel { const void* fn; size_t lds; };

int min_blocks_per_cu(const PersistentKernel* ks, int n) {
  int best = 1 << 30;
  for (int i = 0; i < n; ++i) {
    if (ks[i].lds > 64 * 1024 &&
        hipFuncSetAttribute(ks[i].fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return 0;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ks[i].fn, 256, ks[i].lds) != hipSuccess) return 0;
    best = std::min(best, nb);
  }
  return best;
}

int persistent_blocks_per_cu(bool gru) {
  if (fake_occupancy() >= 0) return fake_occupancy();
  static std::atomic<int> cache[2][64];   // per family and device ordinal; 0 = not queried, else answer + 1
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::atomic<int>& slot = cache[gru ? 1 : 0][dev & 63];
  int v = slot.load(std::memory_order_relaxed);
  if (v == 0) {
    const size_t red = (size_t)RED_FLOATS * sizeof(float), big = ((size_t)1024 * 32 + RED_FLOATS) * sizeof(float);
    int nb;
    if (gru) {
      const PersistentKernel ks[] = {{(const void*)gru_persistent_kernel<20, 10>, red}, {(const void*)gru_persistent_kernel<10, 10>, red},
                                     {(const void*)gru_persistent_kernel<16, 8>, red},  {(const void*)gru_persistent_kernel<12, 8>, red},
                                     {(const void*)gru_persistent_kernel<8, 8>, red},   {(const void*)gru_persistent_kernel<6, 8>, red},
                                     {(const void*)gru_persistent_kernel<4, 8>, red}};
      nb = min_blocks_per_cu(ks, 7);
    } else {
      // the heaviest instantiation of every LSTM family (all are __launch_bounds__(256, 1); registers and LDS are
      // what the API prices): LDS-resident weights at H = 1024, register-resident split / f32 weights at H = 1024
      const PersistentKernel ks[] = {
          {(const void*)lstm_persistent_kernel<1, false, true>, big},
          {(const void*)lstm_persistent_kernel<2, false, true>, big},
          {(const void*)lstm_persistent_split_kernel<1, 4, false>, big},
          {(const void*)lstm_persistent_split_kernel<2, 0, false>, big},
          {(const void*)lstm_persistent_split2_kernel<8, false>, red},
          {(const void*)lstm_persistent_split2_kernel<16, false>, red + (size_t)4 * 16 * 2 * 64 * 16},
          {(const void*)lstm_persistent_split2_kernel<8, true>, red},
          {(const void*)lstm_persistent_split2_kernel<8, false, false, ms::PREC_F16>, red},
          {(const void*)lstm_persistent_split2_kernel<8, false, false, ms::PREC_F16X3>, red},
          {(const void*)lstm_persistent_split2_kernel<16, false, false, ms::PREC_F16X3>, red + (size_t)4 * 16 * 2 * 64 * 16},
          {(const void*)lstm_persistent_split_kernel<1, 4, false, false, true>, big},
          {(const void*)lstm_persistent_f32x2_kernel<16, false>, red},
          {(const void*)lstm_persistent_f32x2_kernel<16, true>, red}};
      nb = min_blocks_per_cu(ks, (int)(sizeof(ks) / sizeof(ks[0])));
    }
    v = nb + 1;
    slot.store(v, std::memory_order_relaxed);
  }
  return v - 1;
}

// Two persistent launches must never be resident together: each fills (up to) every CU with workgroups that wait for
// their own peers, so two half-resident grids on two streams would wait for each other until the spin limit.  Launches
// of this process are therefore chained across streams: a persistent launch on stream S waits (on the device) for the
// previous persistent launch if that one went to a different stream.  With a single stream -- the normal case -- this
// costs nothing: no event is recorded until a second stream shows up (that first hand-over synchronises the earlier
// stream on the host once; from then on every persistent launch records an event).
struct PersistentChain {
  std::mutex mu;
  hipStream_t last = nullptr;
  bool any = false, multi = false;
  hipEvent_t done = nullptr;
};
PersistentChain g_chain[64];

struct PersistentTurn {
  PersistentChain* c = nullptr;
  hipStream_t stream;
  int rc = MS_OK;
  explicit PersistentTurn(hipStream_t s) : stream(s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { rc = MS_ERR_HIP; return; }
    // A launch that is being CAPTURED into a HIP graph (streaming.py replays a chunk's launches as one graph) is not chained:
    // nothing runs now, and the capture stream is not the stream the graph will be launched on.  A graph that holds
    // persistent launches must therefore not be replayed beside persistent launches of another stream of this process.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) return;
    (void)hipGetLastError();
    c = &g_chain[dev & 63];
    c->mu.lock();
    if (c->any && c->last != stream) {
      if (!c->multi) {
        // (the earlier stream may have been destroyed by its owner since: then everything is waited for once)
        if (hipStreamSynchronize(c->last) != hipSuccess) {
          (void)hipGetLastError();
          if (hipDeviceSynchronize() != hipSuccess) rc = MS_ERR_HIP;
        }
        if (hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) rc = MS_ERR_HIP;
        c->multi = true;
      } else if (hipStreamWaitEvent(stream, c->done, 0) != hipSuccess) {
        rc = MS_ERR_HIP;
      }
    }
  }
  ~PersistentTurn() {
    if (!c) return;
    if (c->multi && c->done) (void)hipEventRecord(c->done, stream);
    c->last = stream;
    c->any = true;
    c->mu.unlock();
  }
};
}  // namespace

using ms::ProfScope;

// ================================================================================================ C ABI

// ---- one exchange initialisation for a whole stack (five hx_init launches of ~4.8 us each sat on a
// streaming chunk's critical path).  Every layer of a stack runs the same kernel on the same (steps, N, H, ndir), so their
// exchange regions start out identical: with a region per layer (workspace regions 0 .. 7, selected by flags bits 8 .. 11)
// ONE launch initialises all of them -- and zeroes the per-call status / epoch words once -- before the first layer, and the
// layer calls carry MS_RNN_HX_PREINIT.  Served: the tagged-exchange kernels with ONE launch per layer call -- the wide-workgroup
// LSTM (<= 64 sequences), the two-stream LSTM and the persistent GRU at <= 32 sequences.
bool use_wide(int cell, int H, int ndir, int N);
bool hx_preinit_ok(int cell, int N, int H, int ndir) {
  if (use_wide(cell, H, ndir, N)) return true;
  if (N > 32) return false;
  if (use_fast(cell, H, ndir)) {
    static const bool one_stream = getenv("MS_LSTM_ONE_STREAM") && getenv("MS_LSTM_ONE_STREAM")[0] == '1';
    return use_split(cell, H, ndir) && two_stream_shape(H) && (!one_stream || use_f16(cell, H, ndir) || H > 1024);
  }
  return use_gru_persistent(cell, H, ndir);
}

extern "C" int ms_rnn_hx_preinit(int cell, int T, int N, int In, int H, int ndir, int max_len, int nregions, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
  MS_REQUIRE(cell >= 0 && cell <= MS_CELL_HARD_LSTM && T > 0 && N > 0 && In > 0 && H > 0 && (ndir == 1 || ndir == 2), "bad shape");
  MS_REQUIRE(workspace && max_len >= 1 && max_len <= T && nregions >= 1 && nregions <= HX_REGIONS, "bad argument");
  if (!hx_preinit_ok(cell, N, H, ndir)) return MS_ERR_UNSUPPORTED;     // (not an error: the caller lets every layer initialise its own)
  const WsLayout W = ws_layout(cell, T, N, H, ndir, In);
  MS_REQUIRE(workspace_bytes >= W.total, "workspace too small");
  char* ws = (char*)workspace;
  const bool wide = use_wide(cell, H, ndir, N);
  const int rs = wide ? 1 : lstm_ring_shift();
  const int groups = wide ? ms::cdiv(N, 32) : 1;
  const size_t words_per_dir = (size_t)32 * H << rs;
  MS_REQUIRE(words_per_dir * ndir * groups * sizeof(unsigned) <= W.hx_bytes, "exchange region smaller than the kernel's ring");
  hipStream_t stream = (hipStream_t)stream_;
  for (int r = 0; r < nregions; ++r) {
    // (regions are W.hx_bytes apart, which may exceed what the kernel uses: one launch per region would be nregions launches,
    // so the kernel walks "directions" of a combined index space only when the regions are exactly contiguous)
    if (words_per_dir * ndir * groups * sizeof(unsigned) == W.hx_bytes) {
      hipLaunchKernelGGL(hx_init_kernel, dim3(blocks_for(words_per_dir * ndir * groups * nregions)), dim3(256), 0, stream,
                         (unsigned*)(ws + W.hx), words_per_dir, (size_t)8 * H, ndir * groups * nregions, max_len, rs,
                         (unsigned*)(ws + W.status), (int)((W.xproj - W.status) / sizeof(unsigned)), ndir);
      MS_LAUNCH_CHECK();
      return MS_OK;
    }
    hipLaunchKernelGGL(hx_init_kernel, dim3(blocks_for(words_per_dir * ndir * groups)), dim3(256), 0, stream,
                       (unsigned*)(ws + W.hx + (size_t)r * W.hx_bytes), words_per_dir, (size_t)8 * H, ndir * groups, max_len, rs,
                       (unsigned*)(ws + W.status), (int)((W.xproj - W.status) / sizeof(unsigned)), ndir);
    MS_LAUNCH_CHECK();
  }
  return MS_OK;
}

// Hidden sizes without a persistent kernel (LSTM: not a multiple of 64; GRU: not one of gru_units()'s widths) used to fall to
// one launch per step -- GRU-800 68 ms, LSTM-1000 100 ms per layer at [501, 32, .] against 1.7 .. 3 ms for their persistent
// neighbours (VERDICT r4 missing 3).  The caller pads such a layer to the width returned here with zero weight rows /
// columns and zero biases: a padded unit's gates are then exactly sigmoid(0), tanh(0), so its c and h stay exactly 0 from a
// zero initial state (LSTM: c = .5 * 0 + .5 * 0, h = .5 * tanh(0); GRU: n = tanh(0 + .5 * 0) = 0, h = .5 * 0 + .5 * 0;
// hard cells: clamp(.2 * 0 + .5) = .5, hardtanh(0) = 0), it contributes 0 * w = 0 to every real unit's sums, and the real
// units' k-ordered sums only gain exact zeros.  Returns H itself when H has a persistent kernel or no wider one exists.
extern "C" int ms_rnn_padded_hidden(int cell, int H, int ndir) {
  static const bool off = getenv("MS_RNN_PAD_HIDDEN") && getenv("MS_RNN_PAD_HIDDEN")[0] == '0';
  if (off || cell < 0 || cell > MS_CELL_HARD_LSTM || H <= 0 || ndir < 1 || ndir > 2 || force_generic()) return H;
  if (cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM) {
    auto ok = [&](int h) { return use_fast(cell, h, ndir) && (!want_split() || use_split(cell, h, ndir)); };
    // (round 6) in the two-plane modes a width of 129 .. 1 024 units that is not one of the two-stream kernel's (256, 512, 768,
    // 1 024) runs at the NEXT of those: the one-stream kernel that serves the other multiples of 64 takes 1.8 .. 4.9 ms per
    // bidirectional layer at [501, 32, H] for H = 320 .. 960 where its two-stream neighbours take 1.6 / 1.9 / 2.4 (tools/
    // width_sweep.py) -- LSTM-640 3.19 -> 1.93 ms, LSTM-832 (what 800 was padded to) 4.23 -> 2.44.  64 and 128 stay (0.8 / 1.0 ms).
    static const bool two_stream_pad_off = getenv("MS_RNN_PAD_TWO_STREAM") && getenv("MS_RNN_PAD_TWO_STREAM")[0] == '0';
    if (want_split() && !two_stream_pad_off && H > 128 && H <= 1024 && !two_stream_shape(H)) {
      for (int h : {256, 512, 768, 1024})
        if (h >= H && ok(h)) return h;
    }
    if (ok(H)) return H;
    for (int h = ms::cdiv(H, 64) * 64; h <= 1024; h += 64)
      if (ok(h)) return h;
    for (int h : {1280, 1536, 2048})
      if (h >= H && ok(h)) return h;
  } else if (cell == MS_CELL_GRU) {
    if (use_gru_persistent(cell, H, ndir)) return H;
    for (int h : {512, 768, 1024, 1280, 1536, 2048, 2560})
      if (h >= H && use_gru_persistent(cell, h, ndir)) return h;
  }
  // no persistent kernel at or above this width: at least the MFMA step kernel (H % 64 == 0) instead of the scalar one, which
  // takes 0.5 .. 1.5 ms PER STEP at H = 3 000 (tools/fallback_audit.py)
  return (H % 64 != 0) ? ms::cdiv(H, 64) * 64 : H;
}

extern "C" size_t ms_rnn_packed_bytes(int cell, int In, int H, int ndir) {
  if (cell < 0 || cell > MS_CELL_HARD_LSTM || In <= 0 || H <= 0 || ndir < 1 || ndir > 2) return 0;
  return pack_layout(cell, In, H, ndir).total;
}

extern "C" int ms_rnn_pack(int cell, int In, int H, int ndir, const float* const* w_ih, const float* const* w_hh,
                           const float* const* b_ih, const float* const* b_hh, void* packed, void* stream_) {
  MS_REQUIRE(cell >= 0 && cell <= MS_CELL_HARD_LSTM, "unknown cell");
  MS_REQUIRE(In > 0 && H > 0 && (ndir == 1 || ndir == 2), "bad shape");
  MS_REQUIRE(w_ih && w_hh && packed, "null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int G = gates_of(cell);
  const size_t GH = (size_t)G * H;
  const PackLayout L = pack_layout(cell, In, H, ndir);
  char* base = (char*)packed;
  const bool fast = use_fast(cell, H, ndir);
  for (int d = 0; d < ndir; ++d) {
    MS_REQUIRE(w_ih[d] && w_hh[d], "null weight pointer");
    float* wih_d = (float*)(base + L.wih) + (size_t)d * GH * In;
    float* bx_d = (float*)(base + L.bias_x) + (size_t)d * GH;
    float* whh_d = (float*)(base + L.whh) + (size_t)d * GH * H;
    float* bhh_d = (float*)(base + L.bhh) + (size_t)d * GH;
    const float* bi = b_ih ? b_ih[d] : nullptr;
    const float* bh = b_hh ? b_hh[d] : nullptr;
    if (fast) {
      if (use_split_gemm(cell, H, ndir, In)) {
        unsigned short* hi0 = (unsigned short*)(base + L.wih);
        unsigned short* lo0 = hi0 + (size_t)ndir * GH * In;
        const int prec = layer_prec(cell, H, ndir);
        auto kern = prec == ms::PREC_F16 ? pack_rows_split_kernel<ms::PREC_F16>
                    : prec == ms::PREC_F16X3 ? pack_rows_split_kernel<ms::PREC_F16X3> : pack_rows_split_kernel<ms::PREC_BF16X3>;
        hipLaunchKernelGGL(kern, dim3(4 * H), dim3(128), 0, stream, w_ih[d], hi0 + (size_t)d * GH * In, lo0 + (size_t)d * GH * In, H, In);
      } else {
        hipLaunchKernelGGL(pack_rows_fast_kernel, dim3(4 * H), dim3(128), 0, stream, w_ih[d], wih_d, H, In);
      }
      hipLaunchKernelGGL(pack_bias_fast_kernel, dim3(ms::cdiv(4 * H, 256)), dim3(256), 0, stream, bi, bh, bx_d, H);
      if (use_split(cell, H, ndir)) {
        const int prec = layer_prec(cell, H, ndir);
        auto kern = prec == ms::PREC_F16 ? pack_whh_split_kernel<ms::PREC_F16>
                    : prec == ms::PREC_F16X3 ? pack_whh_split_kernel<ms::PREC_F16X3> : pack_whh_split_kernel<ms::PREC_BF16X3>;
        hipLaunchKernelGGL(kern, dim3(blocks_for(GH * H)), dim3(256), 0, stream, w_hh[d], (unsigned short*)whh_d, H);
      } else if (use_f32x2(cell, H, ndir))
        hipLaunchKernelGGL(pack_whh_f32x2_kernel, dim3(blocks_for(GH * H)), dim3(256), 0, stream, w_hh[d], whh_d, H);
      else
        hipLaunchKernelGGL(pack_whh_fast_kernel, dim3(blocks_for(GH * H)), dim3(256), 0, stream, w_hh[d], whh_d, H);
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH)), dim3(256), 0, stream, (const float*)nullptr, bhh_d,
                         GH);
    } else {
      if (use_split_gemm(cell, H, ndir, In)) {
        // natural row order, bf16 hi / lo planes (plane stride = ndir * GH * In elements)
        unsigned short* hi0 = (unsigned short*)(base + L.wih);
        unsigned short* lo0 = hi0 + (size_t)ndir * GH * In;
        int rc = ms::split_planes_launch(w_ih[d], hi0 + (size_t)d * GH * In, lo0 + (size_t)d * GH * In, GH * In,
                                         two_plane_mode(), stream);
        if (rc != MS_OK) return rc;
      } else {
        hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH * In)), dim3(256), 0, stream, w_ih[d], wih_d, GH * In);
      }
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH)), dim3(256), 0, stream, bi, bx_d, GH);
      if (use_gru_persistent(cell, H, ndir)) {
        const int U = gru_units(H);
        unsigned short* dst = (unsigned short*)(base + L.whh) + (size_t)d * (H / U) * 64 * H;
        auto kern = two_plane_mode() == ms::PREC_F16X3 ? pack_whh_gru_kernel<true> : pack_whh_gru_kernel<false>;
        hipLaunchKernelGGL(kern, dim3(blocks_for((size_t)(H / U) * (H / 8) * 256)), dim3(256), 0, stream, w_hh[d], dst, H, U);
      } else {
        hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH * H)), dim3(256), 0, stream, w_hh[d], whh_d, GH * H);
      }
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH)), dim3(256), 0, stream, bh, bhh_d, GH);
    }
    MS_LAUNCH_CHECK();
  }
  return MS_OK;
}

extern "C" size_t ms_rnn_workspace_bytes(int cell, int T, int N, int In, int H, int ndir) {
  if (cell < 0 || cell > MS_CELL_HARD_LSTM || T <= 0 || N <= 0 || In <= 0 || H <= 0 || ndir < 1 || ndir > 2) return 0;
  return ws_layout(cell, T, N, H, ndir, In).total;
}

template <int NB, bool HARD, bool PIPE, bool STAMP = false>
static int launch_persistent(const LstmP& p, hipStream_t stream) {
  const size_t lds = ((size_t)p.H * 32 + RED_FLOATS) * sizeof(float);
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_kernel<NB, HARD, PIPE, STAMP>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  hipLaunchKernelGGL((lstm_persistent_kernel<NB, HARD, PIPE, STAMP>), dim3(p.ndir * p.J), dim3(256), lds, stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

template <int NB, int NCH, bool HARD, bool STAMP = false, bool HM = false>
static int launch_split_hm(const LstmP& p, hipStream_t stream) {
  const size_t lds = ((size_t)p.H * 32 + RED_FLOATS) * sizeof(float);
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_split_kernel<NB, NCH, HARD, STAMP, HM>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  hipLaunchKernelGGL((lstm_persistent_split_kernel<NB, NCH, HARD, STAMP, HM>), dim3(p.ndir * p.J), dim3(256), lds, stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
template <int NB, int NCH, bool HARD, bool STAMP = false>
static int launch_split(const LstmP& p, hipStream_t stream) {
  return two_plane_mode() == ms::PREC_F16X3 ? launch_split_hm<NB, NCH, HARD, STAMP, true>(p, stream)
                                            : launch_split_hm<NB, NCH, HARD, STAMP, false>(p, stream);
}

constexpr size_t SPLIT2_WL_LDS_BYTES = (size_t)4 * 16 * 2 * 64 * 16;       // H = 2048, two-plane modes: the lo plane of W_hh (128 KB)
template <int KS, bool HARD, bool STAMP = false, int P = ms::PREC_BF16X3>
static int launch_split2(const LstmP& p, hipStream_t stream) {
  const size_t lds = (size_t)RED_FLOATS * sizeof(float) + ((KS == 16 && P != ms::PREC_F16) ? SPLIT2_WL_LDS_BYTES : 0);
  if (lds > 64 * 1024) {
    static ms::DeviceOnce attr_once;
    if (attr_once.need()) {
      MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_split2_kernel<KS, HARD, STAMP, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_once.done();
    }
  }
  if (p.ndir * p.J > ms::num_cus()) {
    // every workgroup of a launch has to be resident: the directions of a wide bidirectional layer run one after the other
    for (int d = 0; d < p.ndir; ++d) {
      LstmP q = p;
      q.d_base = d;
      hipLaunchKernelGGL((lstm_persistent_split2_kernel<KS, HARD, STAMP, P>), dim3(p.J), dim3(256), lds, stream, q);
      MS_LAUNCH_CHECK();
    }
    return MS_OK;
  }
  const int groups = p.grp_wgs ? ms::cdiv(p.N, 32) : 1;
  hipLaunchKernelGGL((lstm_persistent_split2_kernel<KS, HARD, STAMP, P>), dim3(groups * p.ndir * p.J), dim3(256), lds, stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

// wide workgroups (16 units, 8 waves), `groups` batch groups of <= 32 rows side by side in one launch
template <int P>
static int launch_wide2_p(const LstmP& p, bool hard, int groups, hipStream_t stream) {
  const size_t lds = (size_t)2 * WIDE_RED2 * sizeof(float);
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<false, 3, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<true, 3, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<false, 4, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (P == ms::PREC_F16) {
      MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<false, 2, false, false, ms::PREC_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
      MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<true, 4, false, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<false, 3, true, false, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<false, 3, false, true, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<true, 3, false, true, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    attr_once.done();
  }
  const dim3 grid(groups * p.ndir * (p.J / 2));
  if constexpr (P == ms::PREC_F16) {   // MS_PRECISION=fp16: one fp16 plane of h (32 KB pulled per workgroup and stream-step) and one MFMA pass
    // cell waves take MS_LSTM_WIDE_KSC_F16 of the 8 k-steps of a pair each (default 3, as in the bf16x3 form)
    static const int kf = getenv("MS_LSTM_WIDE_KSC_F16") ? atoi(getenv("MS_LSTM_WIDE_KSC_F16")) : 3;
    if (hard) hipLaunchKernelGGL((lstm_persistent_wide2_kernel<true, 3, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    else if (kf == 4) hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 4, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    else if (kf == 2) hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 2, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    else hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 3, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    MS_LAUNCH_CHECK();
    return MS_OK;
  } else {
  static const bool stamps = getenv("MS_LSTM_STAMPS") && getenv("MS_LSTM_STAMPS")[0] == '1';
  if (p.row_off != nullptr) {   // packed rows (ragged batch): xproj and the planes hold only the rows that exist
    if (hard) hipLaunchKernelGGL((lstm_persistent_wide2_kernel<true, 3, false, true, P>), grid, dim3(512), lds, stream, p, p.row_off);
    else hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 3, false, true, P>), grid, dim3(512), lds, stream, p, p.row_off);
    MS_LAUNCH_CHECK();
    return MS_OK;
  }
  if (stamps && !hard) {   // diagnostic build (tools/wide_stamps.py): the shipped arithmetic with wall-clock stamps around its phases
    hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 3, true, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    MS_LAUNCH_CHECK();
    return MS_OK;
  }
  // k-steps per cell wave: 3 (waves 4-7: 5) measured 1.96 ms per layer of two batches against 2.11 for equal eighths and
  // 3.1 for 2 / 6, which spills (profiles/r03z_*); MS_LSTM_WIDE_KSC=4 keeps the equal split for A/B runs
  static const int ksc = getenv("MS_LSTM_WIDE_KSC") ? atoi(getenv("MS_LSTM_WIDE_KSC")) : 3;
  if (hard) {
    if (ksc == 4) hipLaunchKernelGGL((lstm_persistent_wide2_kernel<true, 4, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    else hipLaunchKernelGGL((lstm_persistent_wide2_kernel<true, 3, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
  } else {
    if (ksc == 4) hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 4, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
    else hipLaunchKernelGGL((lstm_persistent_wide2_kernel<false, 3, false, false, P>), grid, dim3(512), lds, stream, p, p.row_off);
  }
  MS_LAUNCH_CHECK();
  return MS_OK;
  }
}
static int launch_wide2(const LstmP& p, bool hard, int groups, int prec, hipStream_t stream) {
  if (prec == ms::PREC_F16) return launch_wide2_p<ms::PREC_F16>(p, hard, groups, stream);
  if (prec == ms::PREC_F16X3) return launch_wide2_p<ms::PREC_F16X3>(p, hard, groups, stream);
  return launch_wide2_p<ms::PREC_BF16X3>(p, hard, groups, stream);
}

// The wide-workgroup kernel serves H = 1024 bf16x3 layers of up to 64 sequences: one batch group of <= 32 rows on 128
// workgroups (1.70 ms per layer against 1.74 for the 8-unit kernel on 256, and HALF of the CUs stay free -- for the other
// batch's projection GEMM when two batches are in flight, pipeline.py), two groups side by side on 256 (1.96 ms against
// 3.5 ms for two launches; profiles/r03w_*, r03ad_*).  Every utterance goes through the same arithmetic whichever group it
// is in, so one batch at a time, two batches in flight and two batches per forward give the same bits.  MS_LSTM_WIDE=0
// switches it off (the 8-unit kernel everywhere; A/B runs).
bool use_wide(int cell, int H, int ndir, int N) {
  static const int mode = getenv("MS_LSTM_WIDE") ? atoi(getenv("MS_LSTM_WIDE")) : -1;    // -1: default (on)
  if (mode == 0) return false;
  if (H != 1024 || N > 64 || !use_split(cell, H, ndir) || ms::precision_mode() == ms::PREC_F32) return false;
  // the occupancy / LDS answer is a property of the device and is asked once; whether THIS call's grid (batch groups x
  // directions x H/16 workgroups, all of which wait for each other) fits the CUs is evaluated on every call
  static std::atomic<int> ok[64];     // per device: 0 = not asked, 1 = a workgroup fits a CU, 2 = it does not
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  int v = ok[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    int nb = 0;
    const size_t lds = (size_t)2 * WIDE_RED2 * sizeof(float);
    const bool fits = hipFuncSetAttribute((const void*)lstm_persistent_wide2_kernel<false, 3, false, false, ms::PREC_F16X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)lstm_persistent_wide2_kernel<false, 3, false, false, ms::PREC_F16X3>, 512, lds) == hipSuccess &&
                      nb >= 1;
    v = fits ? 1 : 2;
    ok[dev & 63].store(v, std::memory_order_relaxed);
  }
  if (v != 1) return false;
  return ms::cdiv(N, 32) * ndir * (H / 16) <= ms::num_cus();
}

template <int G, bool HARD>
static int launch_f32x2(const LstmP& p, hipStream_t stream) {
  const size_t lds = (size_t)RED_FLOATS * sizeof(float);
  hipLaunchKernelGGL((lstm_persistent_f32x2_kernel<G, HARD>), dim3(p.ndir * p.J), dim3(256), lds, stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

static int launch_f32x2_any(const LstmP& p, bool hard, hipStream_t stream) {
  switch (p.H) {
    case 256: return hard ? launch_f32x2<4, true>(p, stream) : launch_f32x2<4, false>(p, stream);
    case 512: return hard ? launch_f32x2<8, true>(p, stream) : launch_f32x2<8, false>(p, stream);
    case 768: return hard ? launch_f32x2<12, true>(p, stream) : launch_f32x2<12, false>(p, stream);
    default: return hard ? launch_f32x2<16, true>(p, stream) : launch_f32x2<16, false>(p, stream);
  }
}

template <int P>
static int launch_split2_prec(const LstmP& p, bool hard, bool stamps, hipStream_t stream) {
  switch (p.H) {
    case 256: return hard ? launch_split2<2, true, false, P>(p, stream) : launch_split2<2, false, false, P>(p, stream);
    case 512: return hard ? launch_split2<4, true, false, P>(p, stream) : launch_split2<4, false, false, P>(p, stream);
    case 768: return hard ? launch_split2<6, true, false, P>(p, stream) : launch_split2<6, false, false, P>(p, stream);
    default: break;
  }
  if constexpr (P != ms::PREC_F16) {
    switch (p.H) {
      case 1280: return hard ? launch_split2<10, true, false, P>(p, stream) : launch_split2<10, false, false, P>(p, stream);
      case 1536: return hard ? launch_split2<12, true, false, P>(p, stream) : launch_split2<12, false, false, P>(p, stream);
      case 2048: return hard ? launch_split2<16, true, false, P>(p, stream) : launch_split2<16, false, false, P>(p, stream);
      default: break;
    }
    if (stamps && !hard) return launch_split2<8, false, true, P>(p, stream);
  }
  return hard ? launch_split2<8, true, false, P>(p, stream) : launch_split2<8, false, false, P>(p, stream);
}
static int launch_split2_any(const LstmP& p, bool hard, bool stamps, int prec, hipStream_t stream) {
  if (prec == ms::PREC_F16) return launch_split2_prec<ms::PREC_F16>(p, hard, stamps, stream);
  if (prec == ms::PREC_F16X3) return launch_split2_prec<ms::PREC_F16X3>(p, hard, stamps, stream);
  return launch_split2_prec<ms::PREC_BF16X3>(p, hard, stamps, stream);
}

// Does a layer of this kind hand its output to the next layer as GEMM operand planes inside the shared workspace
// (MS_RNN_X_PLANES_IN_WS / MS_RNN_OUT_PLANES_TO_WS)?  The two-stream LSTM and the persistent GRU kernels write planes.
extern "C" int ms_rnn_layer_chains_planes(int cell, int H, int ndir) {
  static const bool off = getenv("MS_RNN_CHAIN_PLANES") && getenv("MS_RNN_CHAIN_PLANES")[0] == '0';
  if (off || cell < 0 || cell > MS_CELL_HARD_LSTM) return 0;
  if (use_gru_persistent(cell, H, ndir)) return use_split_gemm(cell, H, ndir, ndir * H) ? 1 : 0;
  return use_split(cell, H, ndir) && two_stream_shape(H) && use_split_gemm(cell, H, ndir, ndir * H) &&
         !(getenv("MS_LSTM_ONE_STREAM") && getenv("MS_LSTM_ONE_STREAM")[0] == '1') ? 1 : 0;
}

extern "C" int ms_rnn_layer_is_wide(int cell, int H, int ndir, int N) {
  if (cell < 0 || cell > MS_CELL_HARD_LSTM || H <= 0 || ndir < 1 || ndir > 2 || N <= 0) return 0;
  return use_fast(cell, H, ndir) && use_wide(cell, H, ndir, N) ? 1 : 0;
}

extern "C" int ms_rnn_layer_forward(int cell, const void* packed, const float* x, const int32_t* lens, int max_len,
                                    const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N,
                                    int In, int H, int ndir, void* workspace, size_t workspace_bytes, void* stream_) {
  return ms_rnn_layer_forward_ex(cell, packed, x, lens, max_len, h0, c0, out, hn, cn, T, N, In, H, ndir, 0, workspace,
                                 workspace_bytes, stream_);
}

// Two batch groups of the 8-unit two-stream LSTM kernel in one launch (LstmP::grp_wgs): both groups' workgroups must be resident
// together (they spin on their own group's peers only, but a launch is dispatched as a whole), the exchange region holds
// exactly two groups (ws_layout: min(npad, 64) rows).  MS_LSTM_PAIR_GROUPS=0 keeps one group per launch (A/B runs).
static bool two_stream_pairs_groups(int cell, int H, int ndir) {
  static const bool off = getenv("MS_LSTM_PAIR_GROUPS") && getenv("MS_LSTM_PAIR_GROUPS")[0] == '0';
  static const bool one_stream = getenv("MS_LSTM_ONE_STREAM") && getenv("MS_LSTM_ONE_STREAM")[0] == '1';
  if (off || !use_fast(cell, H, ndir) || !use_split(cell, H, ndir) || !two_stream_shape(H)) return false;
  if (one_stream && !use_f16(cell, H, ndir) && H <= 1024) return false;
  return 2 * ndir * (H / 8) <= ms::num_cus();
}

// Packed rows need the wide-workgroup recurrence (the only kernel that reads them) and the LDS-DMA GEMM (the only one that
// takes its row count from device memory); the answer must not depend on In, since a stack's layers chain their planes.
static bool layer_packs_rows(int cell, int steps, int N, int In, int H, int ndir) {
  static const bool off = getenv("MS_RNN_PACKED") && getenv("MS_RNN_PACKED")[0] == '0';
  if (off || !(cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM)) return false;
  const int GH4 = 4 * H;
  // (batches beyond 64 rows run as wide launches of 64: the kernel's packed-row index is roff[t] + n_base + n in every launch)
  return use_fast(cell, H, ndir) && use_wide(cell, H, ndir, std::min(N, 64)) && use_split_gemm(cell, H, ndir, In) && !use_f16(cell, H, ndir) &&
         ms::gemm_rows_from_device_ok(steps * N, 32, ndir * GH4) && (size_t)steps * N * std::max(In, ndir * H) * 2 < ((size_t)1 << 31);
}

extern "C" int ms_rnn_layer_packs_rows(int cell, int T, int N, int In, int H, int ndir) {
  if (cell < 0 || cell > MS_CELL_HARD_LSTM || T <= 0 || N <= 0 || In <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return 0;
  return layer_packs_rows(cell, T, N, In, H, ndir) ? 1 : 0;
}

extern "C" int ms_rnn_layer_forward_ex(int cell, const void* packed, const float* x, const int32_t* lens, int max_len,
                                       const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N,
                                       int In, int H, int ndir, int flags, void* workspace, size_t workspace_bytes,
                                       void* stream_) {
  MS_REQUIRE(cell >= 0 && cell <= MS_CELL_HARD_LSTM, "unknown cell");
  MS_REQUIRE(T > 0 && N > 0 && In > 0 && H > 0 && (ndir == 1 || ndir == 2), "bad shape");
  const bool x_in_ws = (flags & MS_RNN_X_PLANES_IN_WS) != 0, out_to_ws = (flags & MS_RNN_OUT_PLANES_TO_WS) != 0;
  MS_REQUIRE(!(x_in_ws || out_to_ws) || ms_rnn_layer_chains_planes(cell, H, ndir), "this layer kind does not chain planes");
  MS_REQUIRE(!x_in_ws || use_split_gemm(cell, H, ndir, In), "input planes need the split GEMM (In % 32 == 0)");
  MS_REQUIRE(packed && (x || x_in_ws) && (out || out_to_ws) && hn && workspace, "null pointer");
  const bool lstm_like = (cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM);
  MS_REQUIRE(!lstm_like || cn, "cn required for LSTM cells");
  MS_REQUIRE(max_len >= 1 && max_len <= T, "max_len must be in [1, T]");
  hipStream_t stream = (hipStream_t)stream_;
  const int G = gates_of(cell);
  const size_t GH = (size_t)G * H;
  const WsLayout W = ws_layout(cell, T, N, H, ndir, In);
  // (the next layer's planes are ndir*H wide and share the xsplit region, which is sized for THIS layer's In)
  const size_t need = out_to_ws ? std::max(W.total, W.xsplit + ms::align_up((size_t)T * N * ndir * H * 4, 256)) : W.total;
  if (workspace_bytes < need) {
    ms::set_error("ms_rnn_layer_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  const PackLayout L = pack_layout(cell, In, H, ndir);
  const char* pk = (const char*)packed;
  char* ws = (char*)workspace;
  const int steps = max_len;
  const bool fast = use_fast(cell, H, ndir);
  const int hx_region = (flags >> 8) & 15;
  MS_REQUIRE(hx_region < HX_REGIONS, "exchange region out of range");
  const bool hx_preinit = (flags & MS_RNN_HX_PREINIT) != 0;      // ms_rnn_hx_preinit initialised this call's region (and zeroed the flags)
  MS_REQUIRE(!hx_preinit || hx_preinit_ok(cell, N, H, ndir), "MS_RNN_HX_PREINIT on a layer kind ms_rnn_hx_preinit does not serve");
  char* hx_ptr = ws + W.hx + (size_t)hx_region * W.hx_bytes;
  // packed rows: only the rows (t, n) with t < lens[n] go through the projection and the planes (see row_offsets_kernel)
  const bool packed_rows = (flags & MS_RNN_PACKED_ROWS) != 0 && lens != nullptr && layer_packs_rows(cell, steps, N, In, H, ndir);
  MS_REQUIRE(packed_rows || !(flags & MS_RNN_PACKED_ROWS) || lens == nullptr || !(x_in_ws || out_to_ws),
             "MS_RNN_PACKED_ROWS on a chained layer that cannot pack its rows (ask ms_rnn_layer_packs_rows for every layer of the stack)");
  int32_t* row_off = (int32_t*)(ws + W.row_off);
  if (packed_rows) {
    hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(256), 0, stream, lens, N, steps, row_off);
    MS_LAUNCH_CHECK();
  }

  // status + epoch flags are re-zeroed on every call (cdna_hip_programming.md G16): by hx_init_kernel on the paths that
  // launch it anyway, by a memset otherwise
  const bool hx_path = use_split(cell, H, ndir) || use_f32x2(cell, H, ndir) || use_gru_persistent(cell, H, ndir);
  const bool projection_only = (flags & MS_RNN_PROJECTION_ONLY) != 0;
  MS_REQUIRE(!(projection_only && (flags & MS_RNN_RECURRENCE_ONLY)), "MS_RNN_PROJECTION_ONLY and MS_RNN_RECURRENCE_ONLY exclude each other");
  if (!hx_path && !projection_only) MS_HIP(hipMemsetAsync(ws + W.status, 0, W.xproj - W.status, stream));
  auto zero_base = [&](int n0) { return (unsigned*)(ws + (n0 == 0 ? W.status : W.flags)); };
  auto zero_words = [&](int n0) { return (int)((W.xproj - (n0 == 0 ? W.status : W.flags)) / sizeof(unsigned)); };
  // frames t >= max_len are all padding (the planes of a chained layer are only read up to max_len)
  if (steps < T && out && !projection_only)
    MS_HIP(hipMemsetAsync(out + (size_t)steps * N * ndir * H, 0, (size_t)(T - steps) * N * ndir * H * sizeof(float),
                          stream));

  // (i) input projection for every frame of every direction: [steps*N, In] x [ndir*GH, In]^T
  float* xproj = (float*)(ws + W.xslot(0));
  int rc;
  if (flags & MS_RNN_RECURRENCE_ONLY) {
    rc = MS_OK;       // the projection region holds what an MS_RNN_PROJECTION_ONLY call left there
  } else {
    ProfScope prof(MS_PROF_PROJECTION, stream);
    if (use_split_gemm(cell, H, ndir, In)) {
      unsigned short* xh = (unsigned short*)(ws + W.xsplit);
      unsigned short* xl = xh + (size_t)steps * N * In;
      const unsigned short* wh = (const unsigned short*)(pk + L.wih);
      const unsigned short* wl = wh + (size_t)ndir * GH * In;
      const int prec = layer_prec(cell, H, ndir);
      if (x_in_ws) {
        rc = MS_OK;
      } else if (packed_rows) {
        auto kern = prec == ms::PREC_F16X3 ? split_planes_packed_kernel<true> : split_planes_packed_kernel<false>;
        hipLaunchKernelGGL(kern, dim3(steps * N), dim3(256), 0, stream, x, xh, xl, lens, row_off, N, In);
        rc = hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_HIP;
      } else {
        rc = ms::split_planes_launch(x, xh, xl, (size_t)steps * N * In, prec, stream);
      }
      if (rc == MS_OK) {
        ProfScope gemm_only(In >= 1024 ? MS_PROF_GEMM_K_LARGE : MS_PROF_GEMM_K_SMALL, stream);   // the split GEMM kernel alone, by contraction length
        rc = ms::gemm_bf16x3_launch_rows(xh, xl, wh, wl, (const float*)(pk + L.bias_x), xproj, steps * N, In, (int)(ndir * GH),
                                         MS_ACT_NONE, 0.f, 0.f, prec, stream, packed_rows ? row_off + steps : nullptr);
      }
    } else {
      ProfScope gemm_only(In >= 1024 ? MS_PROF_GEMM_K_LARGE : MS_PROF_GEMM_K_SMALL, stream);
      rc = ms::linear_launch(x, (const float*)(pk + L.wih), (const float*)(pk + L.bias_x), xproj, steps * N, In,
                             (int)(ndir * GH), MS_ACT_NONE, 0.f, 0.f, stream);
    }
  }
  if (rc != MS_OK) return rc;
  if (projection_only) return MS_OK;

  if (fast) {
    PersistentTurn turn(stream);   // never resident together with another stream's persistent launch
    if (turn.rc != MS_OK) { ms::set_error("ms_rnn_layer_forward: cross-stream hand-over of the persistent launch failed"); return turn.rc; }
    ProfScope prof_rec(MS_PROF_RECURRENCE, stream); // after the hand-over's stream wait: the span is the launch(es), not the queueing behind another stream
    // batch groups of <= 64 sequences, one persistent launch each (stream-ordered; the epoch
    // flags are re-zeroed in between, the status word is kept so any time-out is reported)
    const bool f32x2 = use_f32x2(cell, H, ndir);
    // (round 6) batches of more than 64 sequences on a wide-workgroup shape: launches of 64 rows (two groups of 32 side by
    // side), not of 32 on the 8-unit kernel -- [501, 128, 2048] 12.8 -> 9.5 ms per layer, and an utterance then gets the bits
    // it gets in a batch of 32 (tools/batch_sweep.py)
    const bool wide_rows = use_wide(cell, H, ndir, std::min(N, 64));
    // ... and two groups of 32 rows side by side in one launch of the 8-unit two-stream kernel where their workgroups fit
    // the CUs together (LstmP::grp_wgs): H <= 512 bidirectional, H <= 1024 unidirectional
    const bool pair_groups = !wide_rows && two_stream_pairs_groups(cell, H, ndir);
    const int group = (wide_rows || pair_groups) ? 64 : ((use_split(cell, H, ndir) && two_stream_shape(H)) || f32x2) ? 32 : 64;
    for (int n0 = 0; n0 < N; n0 += group) {
      const int ng = std::min(group, N - n0);
      if (n0 > 0 && !hx_path) MS_HIP(hipMemsetAsync(ws + W.flags, 0, W.xproj - W.flags, stream));
      LstmP p;
      p.xproj = xproj;
      p.whh = (const float*)(pk + L.whh);
      p.lens = lens;
      p.h0 = h0; p.c0 = c0; p.out = out; p.hn = hn; p.cn = cn;
      p.out_hi = p.out_lo = nullptr;
      p.row_off = packed_rows ? row_off : nullptr;
      if (out_to_ws) {   // next layer: [steps*N][ndir*H] hi plane, then the lo plane
        p.out_hi = (unsigned short*)(ws + W.xsplit);
        p.out_lo = p.out_hi + (size_t)steps * N * ndir * H;
      }
      p.hx = (float*)hx_ptr;
      p.flags = (unsigned*)(ws + W.flags);
      p.status = (unsigned*)(ws + W.status);
      p.steps = steps; p.N = ng; p.n_base = n0; p.N_total = N; p.H = H; p.ndir = ndir; p.J = H / 8;
      p.d_base = 0;
      p.grp_wgs = 0;
      p.s_begin = 0; p.s_end = steps;
      p.NPAD = ms::cdiv(ng, 32) * 32;
      {
        static const int ps = getenv("MS_LSTM_POLL_SLEEP") ? atoi(getenv("MS_LSTM_POLL_SLEEP")) : 1;
        p.poll_sleep = ps > 0 ? ps : 1;
      }
      const bool hard = (cell == MS_CELL_HARD_LSTM);
      p.dbg = (unsigned long long*)(ws + W.dbg);
      const bool pipe = (H % 256 == 0);
      static const bool stamps = getenv("MS_LSTM_STAMPS") && getenv("MS_LSTM_STAMPS")[0] == '1';
      if (wide_rows) {
        // this launch's batch groups (one or two of <= 32 rows) in ONE launch of the wide-workgroup kernel
        const int groups = ms::cdiv(ng, 32);
        const int rs = 1;
        const size_t words_per_dir = (size_t)32 * H << rs;
        p.ring_shift = rs;
        p.N = ng;                      // the kernel cuts it into groups of 32 rows
        {
          // measured SLOWER (2.2 against 1.9 .. 2.06 ms per layer, profiles/r03y_*): with one h vector per XCD all 32 CUs of
          // the XCD ask the same L2 channels for the same lines at the same time; four vectors per XCD spread the requests
          static const bool on = getenv("MS_LSTM_WIDE_XCD") && getenv("MS_LSTM_WIDE_XCD")[0] == '1';
          p.xcd_map = (on && groups == 2 && ndir == 2 && p.J == 128) ? 1 : 0;
        }
        // one launch for all batch groups: a group's exchange region is `ndir` directions long, so group g's direction d is
        // "direction" g * ndir + d of a region with groups * ndir of them, and the tag a slot starts with depends on the
        // direction's parity only (hx_init_kernel: d & 1 -- forward / backward -- when ndir == 2, forward when ndir == 1)
        if (!hx_preinit || n0 > 0) {
          hipLaunchKernelGGL(hx_init_kernel, dim3(blocks_for(words_per_dir * ndir * groups)), dim3(256), 0, stream,
                             (unsigned*)hx_ptr, words_per_dir, (size_t)8 * H, ndir * groups, steps, rs, zero_base(n0), zero_words(n0), ndir);
          MS_LAUNCH_CHECK();
        }
        if (stamps) MS_HIP(hipMemsetAsync(ws + W.dbg, 0, W.row_off - W.dbg, stream));   // the stamp area only: row_off (packed rows) follows it
        rc = launch_wide2(p, cell == MS_CELL_HARD_LSTM, groups, layer_prec(cell, H, ndir), stream);
        if (rc != MS_OK) return rc;
        continue;
      }
      if (use_split(cell, H, ndir)) {
        const bool hard_ = (cell == MS_CELL_HARD_LSTM);
        if (stamps) MS_HIP(hipMemsetAsync(ws + W.dbg, 0, W.row_off - W.dbg, stream));
        static const bool one_stream = getenv("MS_LSTM_ONE_STREAM") && getenv("MS_LSTM_ONE_STREAM")[0] == '1';
        const bool two_stream = (p.NPAD == 32 || pair_groups) && two_stream_shape(H) && (!one_stream || use_f16(cell, H, ndir) || H > 1024);
        const int groups = (two_stream && pair_groups) ? ms::cdiv(ng, 32) : 1;     // batch groups side by side in this launch
        p.grp_wgs = groups > 1 ? ndir * p.J : 0;
        {
          // every word of every slot starts with the tag that is NOT the first one expected there
          const int rs = two_stream ? lstm_ring_shift() : 1;
          const size_t slab_words = two_stream ? (size_t)8 * H : (size_t)H * p.NPAD / 2;
          const size_t words_per_dir = two_stream ? ((size_t)32 * H << rs) : (size_t)2 * H * p.NPAD;
          p.ring_shift = rs;
          MS_REQUIRE(words_per_dir * ndir * groups * sizeof(unsigned) <= W.hx_bytes, "exchange region smaller than the launch's batch groups");
          if (!(hx_preinit && two_stream) || groups > 1) {
            // (group g's direction d is region g * ndir + d; a region's first tags depend on its direction only)
            hipLaunchKernelGGL(hx_init_kernel, dim3(blocks_for(words_per_dir * ndir * groups)), dim3(256), 0, stream,
                               (unsigned*)hx_ptr, words_per_dir, slab_words, ndir * groups, steps, rs, zero_base(n0), zero_words(n0), ndir);
            MS_LAUNCH_CHECK();
          }
        }
        if (two_stream) {
          rc = launch_split2_any(p, hard_, stamps, layer_prec(cell, H, ndir), stream);
        } else if (p.NPAD == 32 && H == 1024) {
          if (stamps && !hard_) rc = launch_split<1, 4, false, true>(p, stream);
          else rc = hard_ ? launch_split<1, 4, true>(p, stream) : launch_split<1, 4, false>(p, stream);
        } else if (p.NPAD == 32) {
          rc = hard_ ? launch_split<1, 0, true>(p, stream) : launch_split<1, 0, false>(p, stream);
        } else {
          rc = hard_ ? launch_split<2, 0, true>(p, stream) : launch_split<2, 0, false>(p, stream);
        }
        if (rc != MS_OK) return rc;
        continue;
      }
      if (f32x2) {
        // every word of every slot starts with the tag that is NOT the first one expected there (also between groups)
        hipLaunchKernelGGL(hx_init_kernel, dim3(blocks_for((size_t)64 * H * ndir)), dim3(256), 0, stream,
                           (unsigned*)hx_ptr, (size_t)64 * H, (size_t)16 * H, ndir, steps, 1, zero_base(n0), zero_words(n0), ndir);
        MS_LAUNCH_CHECK();
        rc = launch_f32x2_any(p, hard, stream);
        if (rc != MS_OK) return rc;
        continue;
      }
      if (stamps && pipe && p.NPAD == 32 && !(cell == MS_CELL_HARD_LSTM)) {
        MS_HIP(hipMemsetAsync(ws + W.dbg, 0, W.row_off - W.dbg, stream));
        rc = launch_persistent<1, false, true, true>(p, stream);
        if (rc != MS_OK) return rc;
        continue;
      }
      if (p.NPAD == 32) {
        if (pipe) rc = hard ? launch_persistent<1, true, true>(p, stream) : launch_persistent<1, false, true>(p, stream);
        else rc = hard ? launch_persistent<1, true, false>(p, stream) : launch_persistent<1, false, false>(p, stream);
      } else {
        if (pipe) rc = hard ? launch_persistent<2, true, true>(p, stream) : launch_persistent<2, false, true>(p, stream);
        else rc = hard ? launch_persistent<2, true, false>(p, stream) : launch_persistent<2, false, false>(p, stream);
      }
      if (rc != MS_OK) return rc;
    }
    return MS_OK;
  }

  if (use_gru_persistent(cell, H, ndir)) {
    PersistentTurn turn(stream);
    if (turn.rc != MS_OK) { ms::set_error("ms_rnn_layer_forward: cross-stream hand-over of the persistent launch failed"); return turn.rc; }
    ProfScope prof_rec(MS_PROF_RECURRENCE, stream);
    // one persistent launch per group of 32 sequences (two interleaved streams of 16) -- or (round 6) per TWO groups side by
    // side where both groups' workgroups fit the CUs together (GruP::grp_wgs; MS_LSTM_PAIR_GROUPS=0: one group per launch)
    static const bool pair_off = getenv("MS_LSTM_PAIR_GROUPS") && getenv("MS_LSTM_PAIR_GROUPS")[0] == '0';
    const bool pair_groups = !pair_off && 2 * ndir * (H / gru_units(H)) <= ms::num_cus();
    const int rows_per_launch = pair_groups ? 64 : 32;
    for (int n0 = 0; n0 < N; n0 += rows_per_launch) {
      const int rs = lstm_ring_shift();
      const int ng = std::min(rows_per_launch, N - n0), groups = ms::cdiv(ng, 32);
      MS_REQUIRE(((size_t)32 * H << rs) * ndir * groups * sizeof(unsigned) <= W.hx_bytes, "exchange region smaller than the launch's batch groups");
      if (!hx_preinit || groups > 1) {
        hipLaunchKernelGGL(hx_init_kernel, dim3(blocks_for(((size_t)32 * H << rs) * ndir * groups)), dim3(256), 0, stream,
                           (unsigned*)hx_ptr, (size_t)32 * H << rs, (size_t)8 * H, ndir * groups, steps, rs, zero_base(n0), zero_words(n0), ndir);
        MS_LAUNCH_CHECK();
      }
      GruP g;
      g.xproj = xproj;
      g.whh = (const unsigned short*)(pk + L.whh);
      g.bhh = (const float*)(pk + L.bhh);
      g.lens = lens;
      g.h0 = h0; g.out = out; g.hn = hn;
      g.out_hi = g.out_lo = nullptr;
      if (out_to_ws) {
        g.out_hi = (unsigned short*)(ws + W.xsplit);
        g.out_lo = g.out_hi + (size_t)steps * N * ndir * H;
      }
      g.hx = (float*)hx_ptr;
      g.status = (unsigned*)(ws + W.status);
      g.steps = steps; g.N = ng; g.n_base = n0; g.N_total = N; g.ndir = ndir; g.J = H / gru_units(H);
      g.grp_wgs = groups > 1 ? ndir * g.J : 0;
      g.ring_shift = rs;
      {
        static const int ps = getenv("MS_LSTM_POLL_SLEEP") ? atoi(getenv("MS_LSTM_POLL_SLEEP")) : 1;
        g.poll_sleep = ps > 0 ? ps : 1;
      }
      const size_t lds = (size_t)RED_FLOATS * sizeof(float);
      // every workgroup of a launch has to be resident: directions that do not fit the CUs together run one after the other
      const int launches = ndir * g.J > ms::num_cus() ? ndir : 1, dirs = (ndir / launches) * groups;     // (groups > 1 only where launches == 1)
      for (int l = 0; l < launches; ++l) {
        g.d_base = l;
        switch (H) {
#define MS_GRU_CASE(HH, KS_, U_) \
          case HH:                                                                                                         \
            if (two_plane_mode() == ms::PREC_F16X3) hipLaunchKernelGGL((gru_persistent_kernel<KS_, U_, true>), dim3(dirs * g.J), dim3(256), lds, stream, g); \
            else hipLaunchKernelGGL((gru_persistent_kernel<KS_, U_, false>), dim3(dirs * g.J), dim3(256), lds, stream, g);  \
            break;
          MS_GRU_CASE(2560, 20, 10) MS_GRU_CASE(1280, 10, 10) MS_GRU_CASE(2048, 16, 8) MS_GRU_CASE(1536, 12, 8)
          MS_GRU_CASE(1024, 8, 8) MS_GRU_CASE(768, 6, 8) MS_GRU_CASE(512, 4, 8)
#undef MS_GRU_CASE
          default: ms::set_error("persistent GRU: unsupported hidden size"); return MS_ERR_UNSUPPORTED;
        }
        MS_LAUNCH_CHECK();
      }
    }
    return MS_OK;
  }

  // generic: one launch per time step
  ProfScope prof_rec(MS_PROF_RECURRENCE, stream);
  float* sh = (float*)(ws + W.state_h);
  float* sc = (float*)(ws + W.state_c);
  const size_t st = (size_t)ndir * N * H;
  hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(st)), dim3(256), 0, stream, h0, sh, st);
  if (lstm_like) hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(st)), dim3(256), 0, stream, c0, sc, st);
  MS_LAUNCH_CHECK();
  StepP p;
  p.xproj = xproj; p.whh = (const float*)(pk + L.whh); p.bhh = (const float*)(pk + L.bhh); p.lens = lens;
  p.c_state = sc; p.out = out; p.hn = hn; p.cn = cn; p.steps = steps; p.N = N; p.H = H; p.ndir = ndir;
  for (int s = 0; s < steps; ++s) {
    p.s = s;
    p.h_prev = sh + (size_t)(s & 1) * st;
    p.h_next = sh + (size_t)((s + 1) & 1) * st;
    if (!force_generic() && H % 64 == 0) {
      switch (cell) {
        case MS_CELL_LSTM: launch_step_mfma<MS_CELL_LSTM>(p, stream); break;
        case MS_CELL_GRU: launch_step_mfma<MS_CELL_GRU>(p, stream); break;
        case MS_CELL_RNN_TANH: launch_step_mfma<MS_CELL_RNN_TANH>(p, stream); break;
        default: launch_step_mfma<MS_CELL_HARD_LSTM>(p, stream); break;
      }
      continue;
    }
    dim3 grid(H, ndir);
    switch (cell) {
      case MS_CELL_LSTM: hipLaunchKernelGGL(rnn_step_generic_kernel<MS_CELL_LSTM>, grid, dim3(64), 0, stream, p); break;
      case MS_CELL_GRU: hipLaunchKernelGGL(rnn_step_generic_kernel<MS_CELL_GRU>, grid, dim3(64), 0, stream, p); break;
      case MS_CELL_RNN_TANH: hipLaunchKernelGGL(rnn_step_generic_kernel<MS_CELL_RNN_TANH>, grid, dim3(64), 0, stream, p); break;
      default: hipLaunchKernelGGL(rnn_step_generic_kernel<MS_CELL_HARD_LSTM>, grid, dim3(64), 0, stream, p); break;
    }
  }
  MS_LAUNCH_CHECK();
  return MS_OK;
}

// ------------------------------------------------------------------------------------------------ overlapped stack (round 6)
//
// The literal batch-32 step ran its layers strictly one after the other: projection (whole chip, 1.2 ms), recurrence (128 of
// the 256 CUs, 1.75 ms), ... -- half of the chip idle for 9 of 15 ms.  Here layer l's recurrence runs as S launches over
// consecutive time segments (state in place in hn / cn, exchange tags carried by the sequence-time clock) while a second
// stream, behind events, computes layer l+1's projection on the idle CUs IN TWO K HALVES: W_ih(l+1) = [W_f | W_b] along K, the
// forward half needs only layer l's forward outputs and the backward half only its backward outputs.
//   after segment k (forward direction has produced times < F_k, backward direction times >= B_k = steps - F_k):
//     GEMM_K_FIRST   rows of times [F_{k-1}, F_k):  acc = h_fwd . W_f^T                       -> projection buffer (raw accumulators)
//     GEMM_K_SECOND  rows of times [B_k, F_k) not done yet (the range grows outward from the middle of the utterance):
//                                                   acc += h_bwd . W_b^T, + bias               -> projection buffer
// A row's accumulators pass through memory as exact float32 copies between the two launches, so every gate pre-activation is
// the k-ordered chain of the ONE K = 2H GEMM of the layer-by-layer path: the results are its bits (tests/test_gpu_pipeline.py),
// in every mode and for every batch -- which is why the halves are not two buffers summed afterwards (measured: two K = H
// GEMMs writing two buffers cost every mode 0.29 ms per layer, profiles/r06f_overlap_ab.txt) and why the order is forward half
// first for EVERY row (a per-row choice would depend on the batch's longest utterance).  The price of the fixed order: the
// backward half of the early rows and the forward half of the late rows arrive last, so the second launches bunch up in the
// second half of the layer (1.5 of 2 GEMM halves per segment there) and finish ~0.3 ms after the recurrence.
// Stream events only: the failure mode of this schedule is a slow run, never a hung GPU.  Timing emulation before it was
// built: tools/overlap_emulation.py, profiles/r06e_overlap_emulation.txt.
namespace {
struct OverlapCtx {
  std::mutex mu;
  hipStream_t side = nullptr;      // the projection pieces of a layer's segments, in segment order
  hipStream_t urgent = nullptr;    // ... except those of its LAST segment(s): the rows the next layer needs first
  std::vector<hipEvent_t> ev;
};
OverlapCtx g_overlap[64];

// Steps per time segment of a layer, about steps / want: a segment's GEMM pieces have L * N rows, and the side stream's GEMM
// gets the CUs the recurrence leaves (128 of 256), one 256 x 256 tile per CU -- so L * N is made a multiple of 1 024 rows
// (4 row tiles x 32 column tiles = whole rounds of 128 tiles; measured at N = 32: 62-step segments 13.3 ms per stack, 71-step
// segments 14.6, profiles/r06h_overlap_ab.txt) and large enough for the kernels that can take a contraction cut along K
// (ms::gemm_k_halves_ok).  0 = no segmentation that qualifies.
int overlap_segment_steps(int steps, int N, int NG, int want) {
  want = std::max(2, want);
  const long rows = (long)steps * N;
  long j = std::max(1L, (rows + (long)want * 512) / ((long)want * 1024));      // round(rows / (want * 1024))
  for (; j >= 1; --j) {
    const int L = (int)((1024 * j) / N);
    if (L >= 1 && L < steps && ms::gemm_k_halves_ok(L * N, NG)) return L;
  }
  return 0;
}
}  // namespace

extern "C" int ms_rnn_stack_overlap_ok(int cell, int T, int N, int In, int H, int ndir, int nl) {
  static const bool off = getenv("MS_RNN_OVERLAP") && getenv("MS_RNN_OVERLAP")[0] == '0';
  if (off || cell < 0 || cell > MS_CELL_HARD_LSTM || T <= 0 || N <= 0 || In <= 0 || nl < 2 || nl > HX_REGIONS) return 0;
  if (!overlap_shape(cell, H, ndir) || !use_fast(cell, H, ndir) || !use_wide(cell, H, ndir, N) || N > 32) return 0;
  if (!use_split_gemm(cell, H, ndir, In) || !use_split_gemm(cell, H, ndir, ndir * H)) return 0;
  if (!ms_rnn_layer_chains_planes(cell, H, ndir) || !hx_preinit_ok(cell, N, H, ndir)) return 0;
  return overlap_segment_steps(T, N, ndir * 4 * H, 8) > 0 ? 1 : 0;      // (T: the steps the call will run, max_len)
}

extern "C" int ms_rnn_stack_forward(int cell, const void* const* packed_host, const float* x, const int32_t* lens, int max_len,
                                    const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N, int In, int H,
                                    int ndir, int nl, int segments, void* workspace, size_t workspace_bytes, void* stream_) {
  MS_REQUIRE(max_len >= 1 && max_len <= T, "max_len must be in [1, T]");
  MS_REQUIRE(ms_rnn_stack_overlap_ok(cell, max_len, N, In, H, ndir, nl), "this stack does not run on the overlapped schedule (ask ms_rnn_stack_overlap_ok)");
  MS_REQUIRE(packed_host && x && out && hn && cn && workspace, "null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int steps = max_len;
  const size_t GH = (size_t)4 * H;
  const int NG = (int)(ndir * GH);                       // projection columns (both directions' gates)
  const int SL = overlap_segment_steps(steps, N, NG, segments);      // steps per segment
  MS_REQUIRE(SL > 0, "no time segmentation of this stack qualifies (ask ms_rnn_stack_overlap_ok with T = max_len)");
  const int S = std::max(1, steps / SL);                 // the last segment takes the remainder (SL <= its length < 2 SL)
  const WsLayout W = ws_layout(cell, T, N, H, ndir, std::max(In, ndir * H));
  if (workspace_bytes < W.total) {
    ms::set_error("ms_rnn_stack_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  int dev = 0;
  MS_HIP(hipGetDevice(&dev));
  OverlapCtx& oc = g_overlap[dev & 63];
  std::lock_guard<std::mutex> lock(oc.mu);
  if (oc.side == nullptr) MS_HIP(hipStreamCreateWithFlags(&oc.side, hipStreamNonBlocking));
  if (oc.urgent == nullptr) {
    int lo_pri = 0, hi_pri = 0;
    MS_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
    MS_HIP(hipStreamCreateWithPriority(&oc.urgent, hipStreamNonBlocking, hi_pri));
  }
  while ((int)oc.ev.size() < 4 * S) {
    hipEvent_t e;
    MS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    oc.ev.push_back(e);
  }
  char* ws = (char*)workspace;
  const bool hard = cell == MS_CELL_HARD_LSTM;
  const int prec = layer_prec(cell, H, ndir);
  int rc = ms_rnn_hx_preinit(cell, T, N, std::max(In, ndir * H), H, ndir, max_len, nl, workspace, workspace_bytes, stream_);
  if (rc != MS_OK) return rc;
  if (steps < T) MS_HIP(hipMemsetAsync(out + (size_t)steps * N * ndir * H, 0, (size_t)(T - steps) * N * ndir * H * sizeof(float), stream));
  unsigned short* xh = (unsigned short*)(ws + W.xsplit);
  // ---- layer 0's projection: it depends on the caller's x (the convolutions), nothing runs beside it
  {
    const PackLayout L0 = pack_layout(cell, In, H, ndir);
    const char* pk = (const char*)packed_host[0];
    ProfScope prof(MS_PROF_PROJECTION, stream);
    unsigned short* xl = xh + (size_t)steps * N * In;
    rc = ms::split_planes_launch(x, xh, xl, (size_t)steps * N * In, prec, stream);
    if (rc != MS_OK) return rc;
    ProfScope gemm_only(In >= 1024 ? MS_PROF_GEMM_K_LARGE : MS_PROF_GEMM_K_SMALL, stream);
    const unsigned short* wh = (const unsigned short*)(pk + L0.wih);
    rc = ms::gemm_bf16x3_launch_rows(xh, xl, wh, wh + (size_t)ndir * GH * In, (const float*)(pk + L0.bias_x), (float*)(ws + W.xslot(0)),
                                     steps * N, In, NG, MS_ACT_NONE, 0.f, 0.f, prec, stream, nullptr);
    if (rc != MS_OK) return rc;
  }
  const int Kc = ndir * H;                               // width of a chained layer's input planes
  unsigned short* ph = xh;                               // planes [steps * N][Kc]: hi, then lo
  unsigned short* pl = ph + (size_t)steps * N * Kc;
  const size_t st = (size_t)ndir * N * H;                // one layer's (h, c) state
  // What the side stream has finished, for the layer that consumes it: after `ev`, the projection rows of times [a0, a1) and
  // [b0, b1) are complete (a batch = the pieces issued after one segment).  The rows at the two ends of the utterance, which the
  // next layer needs FIRST, are made LAST, behind the backlog of the layer's second half (the side stream's GEMM is longer than
  // the recurrence beside it).  Two experiments live behind switches here (both measured worse than waiting for everything):
  // per-batch waits, and the last batch on a second, high-priority stream (MS_RNN_OVERLAP_URGENT=<segments>).
  struct Batch { int a0, a1, b0, b1; hipEvent_t ev; bool waited; };
  std::vector<Batch> made_prev, made;
  static const int urgent_segments = getenv("MS_RNN_OVERLAP_URGENT") ? atoi(getenv("MS_RNN_OVERLAP_URGENT")) : 0;   // (experiment, see below)
  for (int l = 0; l < nl; ++l) {
    const int in_l = l == 0 ? In : Kc;
    const PackLayout L = pack_layout(cell, in_l, H, ndir);
    const char* pk = (const char*)packed_host[l];
    const bool last_layer = l + 1 == nl;
    hipEvent_t* evl = oc.ev.data() + (size_t)(l & 1) * 2 * S;      // this layer's events: [2 k] segment done, [2 k + 1] its pieces done
    LstmP p;
    p.xproj = (const float*)(ws + W.xslot(l & 1));
    p.whh = (const float*)(pk + L.whh);
    p.lens = lens;
    p.out = last_layer ? out : nullptr;
    p.hn = hn + l * st; p.cn = cn + l * st;
    p.out_hi = last_layer ? nullptr : ph;
    p.out_lo = last_layer ? nullptr : pl;
    p.row_off = nullptr;
    p.grp_wgs = 0;
    p.hx = (float*)(ws + W.hx + (size_t)l * W.hx_bytes);
    p.flags = (unsigned*)(ws + W.flags);
    p.status = (unsigned*)(ws + W.status);
    p.steps = steps; p.N = N; p.n_base = 0; p.N_total = N; p.H = H; p.ndir = ndir; p.J = H / 8;
    p.d_base = 0; p.NPAD = 32; p.xcd_map = 0; p.ring_shift = 1;
    p.dbg = (unsigned long long*)(ws + W.dbg);
    {
      static const int ps = getenv("MS_LSTM_POLL_SLEEP") ? atoi(getenv("MS_LSTM_POLL_SLEEP")) : 1;
      p.poll_sleep = ps > 0 ? ps : 1;
    }
    const PackLayout Ln = pack_layout(cell, Kc, H, ndir);
    const char* pkn = last_layer ? nullptr : (const char*)packed_host[l + 1];
    float* xnext = (float*)(ws + W.xslot((l + 1) & 1));
    made_prev.swap(made);
    made.clear();
    int done_lo = 0, done_hi = 0;                        // times [done_lo, done_hi) have both halves (empty at first)
    int part_hi = 0;                                     // times [0, part_hi) hold at least the forward half's accumulators
    hipEvent_t first_half_ev = nullptr;                  // after it, every forward-half piece issued so far (side stream) is done
    // (profiling spans: ONE around the layer's segment launches and one per segment around its pieces; a span is two event
    // records, and an event per launch cost this schedule 0.6 ms per step of its gain when the spans were on)
    {
    ProfScope prof_rec(MS_PROF_RECURRENCE, stream);
    for (int k = 0; k < S; ++k) {
      const int s0 = k * SL, s1 = k + 1 == S ? steps : (k + 1) * SL;
      // ---- this segment reads the projection rows of times [s0, s1) (forward) and [steps - s1, steps - s0) (backward)
      // A layer starts when ALL of its rows exist (default).  MS_RNN_OVERLAP_WAIT_ALL=0: a segment waits only for the batches
      // that made its rows, so the side stream's backlog runs beside the next layer's first segments -- measured WORSE (13.2
      // against 12.6 ms per stack, same box; with the end rows' batch on a high-priority stream 13.1): a persistent launch that
      // starts while GEMM tiles still hold CUs has part of its workgroups spinning on peers that are not resident yet
      // (profiles/r06_overlap_schedule_ab.txt)
      static const bool wait_all = !(getenv("MS_RNN_OVERLAP_WAIT_ALL") && getenv("MS_RNN_OVERLAP_WAIT_ALL")[0] == '0');
      for (Batch& m : made_prev) {
        auto hits = [&](int q0, int q1) { return (q0 < m.a1 && m.a0 < q1) || (q0 < m.b1 && m.b0 < q1); };
        if (!m.waited && (wait_all || hits(s0, s1) || hits(steps - s1, steps - s0))) {
          MS_HIP(hipStreamWaitEvent(stream, m.ev, 0));
          m.waited = true;
        }
      }
      p.s_begin = s0; p.s_end = s1;
      p.h0 = s0 == 0 ? (h0 ? h0 + l * st : nullptr) : p.hn;     // a continuing segment starts from the state its predecessor left
      p.c0 = s0 == 0 ? (c0 ? c0 + l * st : nullptr) : p.cn;
      {
        PersistentTurn turn(stream);
        if (turn.rc != MS_OK) { ms::set_error("ms_rnn_stack_forward: cross-stream hand-over of the persistent launch failed"); return turn.rc; }
        rc = launch_wide2(p, hard, 1, prec, stream);
        if (rc != MS_OK) return rc;
      }
      if (last_layer) continue;
      // ---- the next layer's projection of what this segment made available, on the idle CUs
      // (the urgent stream never gets a forward-half piece: second launches on the side stream find their accumulators in order)
      hipStream_t ss = (k + urgent_segments >= S && steps - s1 < s1 && part_hi >= steps - s1) ? oc.urgent : oc.side;
      hipEvent_t seg_done = evl[2 * k], pieces_done = evl[2 * k + 1];
      MS_HIP(hipEventRecord(seg_done, stream));
      MS_HIP(hipStreamWaitEvent(ss, seg_done, 0));
      Batch made_now{0, 0, 0, 0, pieces_done, false};
      {
        ProfScope prof(MS_PROF_PROJECTION, ss);
        const unsigned short* wh = (const unsigned short*)(pkn + Ln.wih);
        const unsigned short* wl = wh + (size_t)ndir * GH * Kc;
        const float* bias = (const float*)(pkn + Ln.bias_x);
        // rows of times [t0, t1): GEMM_K_FIRST = the forward half of K (accumulators out), GEMM_K_SECOND = the backward half on top
        // of them (+ bias), GEMM_K_WHOLE = both in one launch -- the same k-ordered chain either way
        auto piece = [&](int t0, int t1, int kmode) -> int {
          if (t1 <= t0) return (int)MS_OK;
          if (kmode == ms::GEMM_K_SECOND && ss != oc.side && first_half_ev != nullptr)
            MS_HIP(hipStreamWaitEvent(ss, first_half_ev, 0));      // its accumulators were written on the other stream
          const size_t r0 = (size_t)t0 * N;
          const int koff = kmode == ms::GEMM_K_SECOND ? H : 0;
          return ms::gemm_bf16x3_launch_ld(ph + r0 * Kc + koff, pl + r0 * Kc + koff, wh + koff, wl + koff, bias, xnext + r0 * NG,
                                           (t1 - t0) * N, kmode == ms::GEMM_K_WHOLE ? Kc : H, NG, MS_ACT_NONE, 0.f, 0.f, prec, ss,
                                           nullptr, Kc, Kc, kmode);
        };
        auto cut_ok = [&](int t0, int t1) { return t1 > t0 && ms::gemm_k_halves_ok((t1 - t0) * N, NG); };
        // rows [t0, t1) become complete: those below part_hi hold the forward half's accumulators already (second launch), the
        // others get the whole contraction at once -- no accumulator round trip for them; a second-launch piece too small for
        // the kernels that take one is simply recomputed whole (the same bits)
        auto complete = [&](int t0, int t1) {
          int rc2 = MS_OK;
          const int mid = std::min(std::max(part_hi, t0), t1);
          if (mid > t0 && cut_ok(t0, mid)) {
            rc2 = piece(t0, mid, ms::GEMM_K_SECOND);
            if (rc2 == MS_OK) rc2 = piece(mid, t1, ms::GEMM_K_WHOLE);
          } else {
            rc2 = piece(t0, t1, ms::GEMM_K_WHOLE);
          }
          return rc2;
        };
        const int f_k = s1, b_k = steps - s1;              // forward done below f_k, backward done from b_k on
        bool first_half_piece = false;
        if (b_k >= f_k) {                                  // nothing the backward direction has produced is among the forward's rows yet
          if (cut_ok(part_hi, f_k)) {                      // (a sliver waits for the next segment)
            rc = piece(part_hi, f_k, ms::GEMM_K_FIRST);
            part_hi = f_k;
            first_half_piece = true;
          }
        } else {
          // rows the forward direction has produced that the backward direction has not reached yet and that hold nothing so
          // far: the forward half now (a sliver stays untouched and is computed whole when its time comes: `complete`
          // decides by part_hi, which is not raised past rows without accumulators)
          if (part_hi < b_k && cut_ok(part_hi, b_k)) {
            rc = piece(part_hi, b_k, ms::GEMM_K_FIRST);
            part_hi = b_k;
            first_half_piece = true;
          }
          if (rc == MS_OK) {
            if (done_hi <= done_lo) {                      // first meeting of the two fronts
              rc = complete(b_k, f_k);
              made_now.a0 = b_k; made_now.a1 = f_k;
            } else {
              rc = complete(b_k, done_lo);
              if (rc == MS_OK) rc = complete(done_hi, f_k);
              made_now.a0 = b_k; made_now.a1 = done_lo; made_now.b0 = done_hi; made_now.b1 = f_k;
            }
          }
          done_lo = b_k; done_hi = f_k;
        }
        if (rc != MS_OK) return rc;
        if (first_half_piece) first_half_ev = pieces_done;
      }
      MS_HIP(hipEventRecord(pieces_done, ss));
      if (made_now.a1 > made_now.a0 || made_now.b1 > made_now.b0) made.push_back(made_now);
    }
    }   // (the recurrence span ends here)
  }
  // (the last layer's segments have waited for every batch of the layer below it; nothing is outstanding on the side streams)
  return MS_OK;
}

extern "C" size_t ms_rnn_debug_offset(int cell, int T, int N, int In, int H, int ndir) {
  return ws_layout(cell, T, N, H, ndir, In).dbg;
}

extern "C" int ms_rnn_status(const void* workspace, void* stream) {
  MS_REQUIRE(workspace, "null pointer");
  MS_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned st = 0;
  MS_HIP(hipMemcpy(&st, workspace, sizeof(st), hipMemcpyDeviceToHost));
  if (st != 0) {
    MS_HIP(hipMemset(const_cast<void*>(workspace), 0, sizeof(st)));  // reported once
    ms::set_error("ms_rnn_status: persistent LSTM kernel timed out waiting for a peer workgroup");
    return MS_ERR_TIMEOUT;
  }
  return MS_OK;
}
