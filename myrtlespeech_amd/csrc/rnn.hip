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
#include <vector>

#include "common.h"

namespace ms {
int linear_launch(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, float lo,
                  float hi, hipStream_t stream);
}

namespace {

using ms::f32x16;
using ms::f32x4;

constexpr int RED_STRIDE = 40;                    // floats per reduction row (conflict-free, see cell read)
constexpr int RED_FLOATS = 4 * 32 * RED_STRIDE;   // 4 waves x 32 batch rows
constexpr size_t STATUS_BYTES = 256;
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

bool use_fast(int cell, int H, int ndir) {
  if (force_generic()) return false;
  if (!(cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM)) return false;
  if (H % 32 != 0 || H > 1024) return false;
  return ndir * (H / 8) <= ms::num_cus();
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
  L.whh = o; o += ms::align_up(ndir * GH * H * sizeof(float), 256);
  L.bhh = o; o += ms::align_up(ndir * GH * sizeof(float), 256);
  L.total = o;
  return L;
}

struct WsLayout {
  size_t status, flags, xproj, hx, state_h, state_c, total;
};
WsLayout ws_layout(int cell, int T, int N, int H, int ndir) {
  const size_t GH = (size_t)gates_of(cell) * H;
  const int npad = ms::cdiv(N, 32) * 32;
  WsLayout L;
  size_t o = 0;
  L.status = o; o += STATUS_BYTES;
  L.flags = o; o += ms::align_up((size_t)ndir * std::max(H / 8, 1) * sizeof(unsigned), 256);
  L.xproj = o; o += ms::align_up((size_t)T * N * ndir * GH * sizeof(float), 256);
  L.hx = o; o += ms::align_up((size_t)ndir * 2 * H * std::min(npad, 64) * sizeof(float), 256);
  L.state_h = o; o += ms::align_up((size_t)2 * ndir * N * H * sizeof(float), 256);
  L.state_c = o; o += ms::align_up((size_t)ndir * N * H * sizeof(float), 256);
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

__global__ void copy_or_zero_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src ? src[i] : 0.f;
}

int blocks_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }

// ------------------------------------------------------------------------------------------------ generic step

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
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
  int steps, N, n_base, N_total, H, ndir, J, NPAD;
};

__device__ __forceinline__ f32x4 load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, /*aux: sc1*/ 16);
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ void store_sc1_f32(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int NB, bool HARD, bool PIPE>
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
    // publish h_s: every storing wave drains, then one lane raises this producer's epoch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(s + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = b * 32 + nl;
      if (n < N) p.out[((size_t)t * p.N_total + p.n_base + n) * (p.ndir * H) + d * H + unit] = hout[b];
    }
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

}  // namespace

// ================================================================================================ launch timing

namespace {
struct ProfSpan { hipEvent_t a, b; int kind; };
bool g_prof_on = false;
std::vector<ProfSpan> g_spans;      // recorded, not yet read
std::vector<ProfSpan> g_free;       // event pairs ready for re-use

struct ProfScope {
  ProfSpan s{};
  hipStream_t stream;
  bool on;
  ProfScope(int kind, hipStream_t st) : stream(st), on(g_prof_on) {
    if (!on) return;
    if (!g_free.empty()) { s = g_free.back(); g_free.pop_back(); }
    else if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) { on = false; return; }
    s.kind = kind;
    (void)hipEventRecord(s.a, stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(s.b, stream);
    g_spans.push_back(s);
  }
};
}  // namespace

extern "C" int ms_prof_enable(int on) {
  g_prof_on = on != 0;
  return MS_OK;
}

extern "C" int ms_prof_read(float* out_ms, int* out_n) {
  MS_REQUIRE(out_ms && out_n, "null pointer");
  out_ms[0] = out_ms[1] = 0.f;
  out_n[0] = out_n[1] = 0;
  for (auto& s : g_spans) {
    MS_HIP(hipEventSynchronize(s.b));
    float ms = 0.f;
    MS_HIP(hipEventElapsedTime(&ms, s.a, s.b));
    out_ms[s.kind] += ms;
    out_n[s.kind] += 1;
    g_free.push_back(s);
  }
  g_spans.clear();
  return MS_OK;
}

// ================================================================================================ C ABI

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
      hipLaunchKernelGGL(pack_rows_fast_kernel, dim3(4 * H), dim3(128), 0, stream, w_ih[d], wih_d, H, In);
      hipLaunchKernelGGL(pack_bias_fast_kernel, dim3(ms::cdiv(4 * H, 256)), dim3(256), 0, stream, bi, bh, bx_d, H);
      hipLaunchKernelGGL(pack_whh_fast_kernel, dim3(blocks_for(GH * H)), dim3(256), 0, stream, w_hh[d], whh_d, H);
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH)), dim3(256), 0, stream, (const float*)nullptr, bhh_d,
                         GH);
    } else {
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH * In)), dim3(256), 0, stream, w_ih[d], wih_d, GH * In);
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH)), dim3(256), 0, stream, bi, bx_d, GH);
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH * H)), dim3(256), 0, stream, w_hh[d], whh_d, GH * H);
      hipLaunchKernelGGL(copy_or_zero_kernel, dim3(blocks_for(GH)), dim3(256), 0, stream, bh, bhh_d, GH);
    }
    MS_LAUNCH_CHECK();
  }
  return MS_OK;
}

extern "C" size_t ms_rnn_workspace_bytes(int cell, int T, int N, int In, int H, int ndir) {
  (void)In;
  if (cell < 0 || cell > MS_CELL_HARD_LSTM || T <= 0 || N <= 0 || H <= 0 || ndir < 1 || ndir > 2) return 0;
  return ws_layout(cell, T, N, H, ndir).total;
}

template <int NB, bool HARD, bool PIPE>
static int launch_persistent(const LstmP& p, hipStream_t stream) {
  const size_t lds = ((size_t)p.H * 32 + RED_FLOATS) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MS_HIP(hipFuncSetAttribute((const void*)lstm_persistent_kernel<NB, HARD, PIPE>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_persistent_kernel<NB, HARD, PIPE>), dim3(p.ndir * p.J), dim3(256), lds, stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_rnn_layer_forward(int cell, const void* packed, const float* x, const int32_t* lens, int max_len,
                                    const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N,
                                    int In, int H, int ndir, void* workspace, size_t workspace_bytes, void* stream_) {
  MS_REQUIRE(cell >= 0 && cell <= MS_CELL_HARD_LSTM, "unknown cell");
  MS_REQUIRE(T > 0 && N > 0 && In > 0 && H > 0 && (ndir == 1 || ndir == 2), "bad shape");
  MS_REQUIRE(packed && x && out && hn && workspace, "null pointer");
  const bool lstm_like = (cell == MS_CELL_LSTM || cell == MS_CELL_HARD_LSTM);
  MS_REQUIRE(!lstm_like || cn, "cn required for LSTM cells");
  MS_REQUIRE(max_len >= 1 && max_len <= T, "max_len must be in [1, T]");
  hipStream_t stream = (hipStream_t)stream_;
  const int G = gates_of(cell);
  const size_t GH = (size_t)G * H;
  const WsLayout W = ws_layout(cell, T, N, H, ndir);
  if (workspace_bytes < W.total) {
    ms::set_error("ms_rnn_layer_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  const PackLayout L = pack_layout(cell, In, H, ndir);
  const char* pk = (const char*)packed;
  char* ws = (char*)workspace;
  const int steps = max_len;
  const bool fast = use_fast(cell, H, ndir);

  // status + epoch flags are re-zeroed on every call (cdna_hip_programming.md G16)
  MS_HIP(hipMemsetAsync(ws + W.status, 0, W.xproj - W.status, stream));
  // frames t >= max_len are all padding
  if (steps < T)
    MS_HIP(hipMemsetAsync(out + (size_t)steps * N * ndir * H, 0, (size_t)(T - steps) * N * ndir * H * sizeof(float),
                          stream));

  // (i) input projection for every frame of every direction: [steps*N, In] x [ndir*GH, In]^T
  float* xproj = (float*)(ws + W.xproj);
  int rc;
  {
    ProfScope prof(0, stream);
    rc = ms::linear_launch(x, (const float*)(pk + L.wih), (const float*)(pk + L.bias_x), xproj, steps * N, In,
                           (int)(ndir * GH), MS_ACT_NONE, 0.f, 0.f, stream);
  }
  if (rc != MS_OK) return rc;
  ProfScope prof_rec(1, stream);

  if (fast) {
    // batch groups of <= 64 sequences, one persistent launch each (stream-ordered; the epoch
    // flags are re-zeroed in between, the status word is kept so any time-out is reported)
    for (int n0 = 0; n0 < N; n0 += 64) {
      const int ng = std::min(64, N - n0);
      if (n0 > 0) MS_HIP(hipMemsetAsync(ws + W.flags, 0, W.xproj - W.flags, stream));
      LstmP p;
      p.xproj = xproj;
      p.whh = (const float*)(pk + L.whh);
      p.lens = lens;
      p.h0 = h0; p.c0 = c0; p.out = out; p.hn = hn; p.cn = cn;
      p.hx = (float*)(ws + W.hx);
      p.flags = (unsigned*)(ws + W.flags);
      p.status = (unsigned*)(ws + W.status);
      p.steps = steps; p.N = ng; p.n_base = n0; p.N_total = N; p.H = H; p.ndir = ndir; p.J = H / 8;
      p.NPAD = ms::cdiv(ng, 32) * 32;
      const bool hard = (cell == MS_CELL_HARD_LSTM);
      const bool pipe = (H % 256 == 0);
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

  // generic: one launch per time step
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

extern "C" int ms_rnn_status(const void* workspace, void* stream) {
  MS_REQUIRE(workspace, "null pointer");
  MS_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned st = 0;
  MS_HIP(hipMemcpy(&st, workspace, sizeof(st), hipMemcpyDeviceToHost));
  if (st != 0) {
    ms::set_error("ms_rnn_status: persistent LSTM kernel timed out waiting for a peer workgroup");
    return MS_ERR_TIMEOUT;
  }
  return MS_OK;
}
