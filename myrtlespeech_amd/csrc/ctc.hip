// CTC loss (log-space alpha recursion) and CTC greedy decode.  No MFMA: these are
// scan / compaction kernels built from wavefront ballot + prefix counts (64-wide waves).
//
//   ms_ctc_loss_forward   <- loss/ctc_loss.py:95-101 (LogSoftmax + torch.nn.CTCLoss)
//   ms_ctc_greedy_decode  <- post_process/ctc_greedy_decoder.py:74-92
#include <math.h>
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int CTC_THREADS = 256;

__device__ __forceinline__ float neg_inf() { return -INFINITY; }

// log(exp(a)+exp(b)+exp(c)) with the max trick; all -inf -> -inf.  The hardware forms (v_exp_f32 / v_log_f32 behind
// __expf / __logf): the arguments are differences to the maximum (<= 0) and the sum lies in [1, 3], where their absolute error
// is ~1e-7 -- against losses of order 10^2 .. 10^3 and a stated tolerance of 2e-5 relative; the precise library forms made
// this function ~150 instructions and the alpha recursion's frame 0.67 us (round 4: 0.34 -> see DESIGN 4, CTC).
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == neg_inf()) return neg_inf();
  return __logf(__expf(a - m) + __expf(b - m) + __expf(c - m)) + m;
}

// One workgroup per utterance.
//   phase 1: logZ[t] = logsumexp_v logits[t,n,:]            (threads over t)
//   phase 2: alpha recursion over the 2L+1 extended states  (threads over states, one barrier per frame)
// LDS: ext labels [S], alpha double buffer [2][S]; logZ lives in the global workspace.
__global__ __launch_bounds__(CTC_THREADS) void ctc_alpha_kernel(const float* __restrict__ logits,
                                                                const int32_t* __restrict__ in_lens,
                                                                const int32_t* __restrict__ targets,
                                                                const int32_t* __restrict__ tgt_offsets,
                                                                const int32_t* __restrict__ tgt_lens,
                                                                float* __restrict__ nll, float* __restrict__ logz_ws,
                                                                int T, int N, int V, int S_max, int blank,
                                                                int log_probs_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tid = threadIdx.x;
  int* ext = reinterpret_cast<int*>(smem);   // [S_max]
  float* alpha0 = smem + S_max;              // [S_max]
  float* alpha1 = alpha0 + S_max;            // [S_max]
  const int Tn = min(max(in_lens[n], 0), T);
  const int L = max(tgt_lens[n], 0);
  const int S = 2 * L + 1;
  float* logz = logz_ws + (size_t)n * T;
  const int32_t* tg = targets + tgt_offsets[n];

  for (int t = tid; t < Tn; t += CTC_THREADS) {
    if (log_probs_in) { logz[t] = 0.f; continue; }   // the caller normalised over another axis (CTCLoss(dim != -1))
    const float* row = logits + ((size_t)t * N + n) * V;
    float m = neg_inf();
    for (int v = 0; v < V; ++v) m = fmaxf(m, row[v]);
    float sum = 0.f;
    for (int v = 0; v < V; ++v) sum += expf(row[v] - m);
    logz[t] = logf(sum) + m;
  }
  for (int s = tid; s < S; s += CTC_THREADS) {
    ext[s] = (s & 1) ? tg[s >> 1] : blank;
    alpha0[s] = neg_inf();
  }
  __syncthreads();
  if (Tn > 0) {
    const float* row = logits + (size_t)n * V;
    if (tid == 0) alpha0[0] = row[blank] - logz[0];
    if (tid == 1 && S > 1) alpha0[1] = row[ext[1]] - logz[0];
  }
  __syncthreads();
  float* cur = alpha0;
  float* nxt = alpha1;
  // The frame's log-probability of this thread's first state (s = tid) does not depend on the recursion: it is fetched one
  // frame AHEAD, so that the global load's latency (~0.5 us out of L2) runs under the previous frame's barrier instead of
  // in series with every one of the T steps (round 4: 0.73 -> 0.2 us per frame at T = 501, 120 labels).  States past the
  // first 256 (targets longer than 127 labels) keep the direct load.
  const int lab0 = tid < S ? ext[tid] : blank;
  const bool skip0 = tid >= 2 && tid < S && lab0 != blank && lab0 != ext[tid - 2];
  float lp_next = 0.f;
  if (Tn > 1 && tid < S) lp_next = logits[((size_t)1 * N + n) * V + lab0] - logz[1];
  for (int t = 1; t < Tn; ++t) {
    const float* row = logits + ((size_t)t * N + n) * V;
    const float lp0 = lp_next;
    if (t + 1 < Tn && tid < S) lp_next = logits[((size_t)(t + 1) * N + n) * V + lab0] - logz[t + 1];
    if (tid < S) {
      const float a0 = cur[tid];
      const float a1 = (tid >= 1) ? cur[tid - 1] : neg_inf();
      const float a2 = skip0 ? cur[tid - 2] : neg_inf();
      const float l = lse3(a0, a1, a2);
      nxt[tid] = (l == neg_inf()) ? neg_inf() : l + lp0;
    }
    if (S > CTC_THREADS) {
      const float lz = logz[t];
      for (int s = tid + CTC_THREADS; s < S; s += CTC_THREADS) {
        const int lab = ext[s];
        const float a0 = cur[s];
        const float a1 = cur[s - 1];
        const float a2 = (lab != blank && lab != ext[s - 2]) ? cur[s - 2] : neg_inf();
        const float l = lse3(a0, a1, a2);
        nxt[s] = (l == neg_inf()) ? neg_inf() : l + (row[lab] - lz);
      }
    }
    // LDS-only barrier: __syncthreads() also drains the vector-memory queue (its workgroup fence waits vmcnt(0)), i.e. it
    // would wait for the log-probability just requested for the NEXT frame and put the L2 latency back into every step
    // (compiler fences on both sides: without them nothing tells the compiler that the LDS stores above must be issued before
    // the barrier and the loads of the next frame after it -- ADVICE r4)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's alpha row is in the LDS
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  __syncthreads();
  if (tid == 0) {
    float ll;
    if (Tn == 0) {
      ll = (S == 1) ? 0.f : neg_inf();
    } else {
      const float l1 = cur[S - 1];
      const float l2 = (S > 1) ? cur[S - 2] : neg_inf();
      const float m = fmaxf(l1, l2);
      ll = (m == neg_inf()) ? neg_inf() : logf(expf(l1 - m) + expf(l2 - m)) + m;
    }
    nll[n] = -ll;
  }
}

// ---- the alpha recursion as a four-wave systolic pipeline (round 4).  The kernel above spends 0.43 us per frame, almost all
// of it the round trip "alpha row to LDS -> barrier -> three dependent LDS reads" that every frame pays.  Here a lane keeps
// its K consecutive states' alphas in REGISTERS for the whole utterance, the neighbours' alpha(s-1) / alpha(s-2) arrive by DPP
// (wave_shr:1), and only the two states at a wave's upper edge cross to the next wave, through an LDS mailbox with one entry per
// frame: wave w computes frame t as soon as wave w-1 has published frame t-1, so the waves run
// skewed by about one frame and nobody waits at a barrier.  The mailbox read for frame t is issued a frame ahead (the upstream
// wave is ahead: the LDS latency is off the chain).  Dependencies only run upwards (wave 0 never waits) and every wave is
// resident (one workgroup), so the waits terminate whatever the data.
// An utterance has ONE wave per SIMD, which issues one instruction of any kind per four cycles: a frame costs its instruction
// count, so the frame is written for few instructions --
//   * phase 1 (all threads, a frame's symbols in registers) writes the frame's log-probabilities once, normalised, in the log2
//     domain and clamped to a finite "log zero" (CTC_NEG) into the workspace: the recursion is max3 / sub / v_exp / add /
//     v_log / add on finite numbers, no -inf cases, no per-frame normaliser, no multiplications by log2(e) / ln 2;
//   * a state's log-probability arrives through a D-deep register ring, one buffer load per frame whose frame offset is a
//     scalar register (an L2 round trip is ~8 frames long);
//   * the main loop runs whole blocks of D frames with no exit inside (an exit per frame makes the compiler rotate the ring
//     through copies, and a copy waits for every load in flight); the last < D frames run predicated and fetch nothing;
//   * the edge waves are separate instantiations (no upstream / no downstream code at all).
// Natural-log losses differ from ctc_alpha_kernel's by rounding only (1e-6 relative in tests/test_gpu_parity.py, which also
// holds both against the oracle).  K = states per lane: S <= 256 K.
constexpr float CTC_NEG = -1.0e30f;
constexpr size_t CTC_STATUS_BYTES = 256;   // sticky time-out word at the start of every CTC workspace (ms_ctc_status)

struct alignas(8) MbEntry { float top, below; };

// REV (the beta recursion): the same recursion on the time-reversed, label-reversed utterance -- frame t' reads row
// Tn - 1 - t', state s' carries the label of state S - 1 - s' -- which IS the beta recursion of loss/ctc_loss.py's backward
// (beta_t(s) = lp[t, l'_s] + lse(beta_{t+1}(s), beta_{t+1}(s+1), [skip] beta_{t+1}(s+2)), started from the last two states).
// ROWS: every frame's values are also written to rows[t][s] (unreversed indices) for the gradient kernel.
template <int K, int D, bool UP, bool DOWN, bool ROWS>
__device__ __forceinline__ void alpha_wave_body(const float* __restrict__ lpn, const int32_t* __restrict__ tg, float* fin,
                                                MbEntry* mb, int T, int Tn, int V, int S, int blank, int tid, float* rows,
                                                int S_max, int rev) {
#define CTC_FENCE() asm volatile("" ::: "memory")
  const int lane = tid & 63, w = tid >> 6;
  const int s0 = tid * K;
  // `lpn` rows: [T][V] log-probabilities (a state reads its label's column), or -- wide alphabets, `gathered` -- [T][S_max]
  // with a column per STATE (a wave's load is then 256 contiguous bytes instead of 64 cache lines of a 4 KB row)
  const int gathered = S_max < 0;
  const int RS = gathered ? -S_max : V;
  S_max = gathered ? -S_max : S_max;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lpn), 0, (int)((size_t)T * RS * sizeof(float)), 0x00020000);
  const int V4 = RS * 4;
  const int L = (S - 1) >> 1;
  int voff[K], voff_st[K];
  bool skip[K];
  float own[K], ring[D][K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int s = s0 + j;
    const bool act = s < S;
    const int li = s >> 1;                                             // label index of an odd state, in recursion order
    const int lab = (act && (s & 1)) ? tg[rev ? L - 1 - li : li] : blank;
    skip[j] = act && (s & 1) && s >= 3 && lab != blank && lab != tg[rev ? L - li : li - 1];
    voff[j] = gathered ? (act ? (rev ? S - 1 - s : s) : 0) * 4 : lab * 4;
    voff_st[j] = act ? (rev ? S - 1 - s : s) * 4 : 0x7fffffff;         // states past S: dropped by the buffer's range check
  }
  // frame t' of the recursion reads row t' (or Tn - 1 - t'); the scalar offset of a buffer access is NOT part of the range
  // check, so it is kept inside the utterance's rows (the edge row is then fetched again; those values are never used)
  const int dV4 = rev ? -V4 : V4;
  const int soff_lo = 0, soff_hi = (T - 1) * V4;
  auto clamp_off = [&](int o) { return min(max(o, soff_lo), soff_hi); };
  const int soff0 = rev ? (Tn - 1) * V4 : 0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j)
      ring[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], clamp_off(soff0 + i * dV4), 0));
  int soff = clamp_off(soff0 + D * dV4);     // byte offset of the next frame to fetch
  // row store: rows[t][s], t = t' or Tn - 1 - t'
  const __amdgpu_buffer_rsrc_t rsrc_st =
      __builtin_amdgcn_make_buffer_rsrc(rows, 0, ROWS ? (int)((size_t)T * S_max * sizeof(float)) : 0, 0x00020000);
  const int dS4 = (rev ? -S_max : S_max) * 4;
  int soff_st = rev ? (Tn - 1) * S_max * 4 : 0;
  float a1in = CTC_NEG, a2in = CTC_NEG;
  int gave_up = 0;
  MbEntry pv{CTC_NEG, CTC_NEG};
  const MbEntry* mb_up = mb + (size_t)(UP ? w - 1 : 0) * T;
  MbEntry* mb_me = mb + (size_t)(DOWN ? w : 0) * T;
  // The mailbox: entry [t] of wave w = the alphas of its two topmost states after frame t, written ONCE by its lane 63 over
  // words that phase 1 set to all ones -- a bit pattern no alpha can have (the log-probabilities are clamped to finite values
  // in phase 1, NaN included, so the recursion stays finite; an arithmetic NaN would be 0x7fc00000 anyway).  An entry is valid
  // when neither word is the sentinel: no counter, one LDS write and one LDS read per frame.  The accesses are plain LDS
  // operations fenced for the COMPILER (volatile turns them into flat accesses that drain the vector-memory queue, i.e. the ring)
  // (two 32-bit reads: a 64-bit one lands in a register pair, and the pair then has to be copied apart for the DPP's tied
  // operands -- a copy that waits for the read where it is issued)
  auto fetch = [&](int t) {
    CTC_FENCE();
    pv.top = reinterpret_cast<const float*>(mb_up + t)[0];
    pv.below = reinterpret_cast<const float*>(mb_up + t)[1];
    CTC_FENCE();
  };

  // nw[] = this frame's alphas -> own[]; the edge values move on (DPP inside the wave, mailbox across waves)
  auto exchange = [&](int t, const float (&nw)[K]) {
    float ex = CTC_NEG, ey = CTC_NEG;
    if (UP) {
      // One asm statement, tied to the frame's result, so that the look at the entry stays BEHIND the arithmetic (left to
      // itself the compiler tests right after the request: the LDS latency exposed and the publication missed).  The spin is
      // bounded (~0.5 s): a wave that does not get its entry goes on with what it has instead of hanging the queue, and the
      // utterance's loss comes out as NaN.
      // The hazard recogniser does not see into the statement: gfx950 needs one wait state between a VALU write of a VGPR
      // and a v_readfirstlane of it (without it the test reads the OLD register and lets a sentinel through -- measured).
      unsigned x = __float_as_uint(pv.top), y = __float_as_uint(pv.below), tmp;
      int spins;
      const unsigned addr = (unsigned)(size_t)(mb_up + t);   // LDS byte address (the low 32 bits of the generic pointer)
      asm volatile(
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_max_u32 %[tmp], %[x], %[y]\n\t"
          "s_mov_b32 %[spins], 0\n\t"   // also the wait state gfx950 wants between a VALU write and a v_readfirstlane of it
          "v_readfirstlane_b32 vcc_lo, %[tmp]\n\t"
          "s_cmp_eq_u32 vcc_lo, -1\n\t"
          "s_cbranch_scc0 1f\n\t"
          "0:\n\t"
          "s_sleep 1\n\t"
          "ds_read_b32 %[x], %[addr]\n\t"
          "ds_read_b32 %[y], %[addr] offset:4\n\t"
          "s_add_u32 %[spins], %[spins], 1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_max_u32 %[tmp], %[x], %[y]\n\t"
          "s_bitcmp1_b32 %[spins], 22\n\t"
          "s_cbranch_scc1 1f\n\t"
          "v_readfirstlane_b32 vcc_lo, %[tmp]\n\t"
          "s_cmp_eq_u32 vcc_lo, -1\n\t"
          "s_cbranch_scc1 0b\n\t"
          "1:\n\t"
          : [x] "+v"(x), [y] "+v"(y), [tmp] "=&v"(tmp), [spins] "=&s"(spins)
          : [addr] "v"(addr), "v"(nw[K - 1])
          : "vcc", "scc", "memory");
      ex = __uint_as_float(x);
      ey = __uint_as_float(y);
      gave_up |= spins;          // bit 22 set: the bounded spin ran out (the loss is then reported as NaN, not as a number)
    }
    const float top = nw[K - 1];
    float below;
    a1in = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ex), __float_as_int(top), 0x138, 0xf, 0xf, false));
    if (K >= 2) {
      below = nw[K >= 2 ? K - 2 : 0];
      a2in = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ey), __float_as_int(below), 0x138, 0xf, 0xf, false));
    } else {
      a2in = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ey), __float_as_int(a1in), 0x138, 0xf, 0xf, false));
      below = a1in;   // lane 63: the alpha of lane 62's state
    }
    if (DOWN && lane == 63) {
      CTC_FENCE();
      mb_me[t] = MbEntry{top, below};
      CTC_FENCE();
    }
    if (UP) fetch(t + 1);   // the next frame's entry, a frame ahead (after the last frame: a word inside the allocation, not used)
    if (ROWS) {
#pragma unroll
      for (int j = 0; j < K; ++j) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nw[j]), rsrc_st, voff_st[j], soff_st, 0);
      soff_st = max(soff_st + dS4, 0);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      own[j] = nw[j];
      // the next frame's arithmetic starts from here, i.e. BEHIND the request above (asm statements keep their order; without
      // this the compiler sinks the request to just in front of its test and exposes the LDS latency)
      if (UP) asm volatile("" : "+v"(own[j]));
    }
  };
  auto frame = [&](int t, float (&slot)[K], bool refill) {
    float nw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float lp = slot[j];
      const float a0 = own[j];
      const float a1 = (j >= 1) ? own[j >= 1 ? j - 1 : 0] : a1in;
      if (K % 2 == 0 && j % 2 == 0) {
        // an even state of an even K is a blank: two terms, one of them exp2(0)
        const float m = fmaxf(a0, a1);
        const float e = __builtin_amdgcn_exp2f(-fabsf(a0 - a1));
        nw[j] = (__builtin_amdgcn_logf(1.f + e) + m) + lp;
      } else {
        const float a2r = (j >= 2) ? own[j >= 2 ? j - 2 : 0] : (j == 1 ? a1in : a2in);
        const float a2 = skip[j] ? a2r : CTC_NEG;
        const float m = fmaxf(fmaxf(a0, a1), a2);
        const float e = (__builtin_amdgcn_exp2f(a0 - m) + __builtin_amdgcn_exp2f(a1 - m)) + __builtin_amdgcn_exp2f(a2 - m);
        nw[j] = (__builtin_amdgcn_logf(e) + m) + lp;
      }
    }
    // the slot is refilled once its value has been USED (requested earlier, the new value would need a second register
    // while the old one is still live, and the compiler then copies the ring back at every block end -- behind a vmcnt(0))
    if (refill) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        asm volatile("" : : "v"(nw[j]));
        slot[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], soff, 0));
      }
      soff = clamp_off(soff + dV4);
    }
    exchange(t, nw);
  };
  {  // frame 0: only the first blank and the first label can start a path
    float nw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int s = s0 + j;
      nw[j] = (s < 2 && s < S) ? ring[0][j] : CTC_NEG;
      ring[0][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], soff, 0));
    }
    soff = clamp_off(soff + dV4);
    if (UP) fetch(0);
    exchange(0, nw);
  }
  int tb = 1;   // frame t's log-probabilities sit in ring slot t % D
  for (; tb + D <= Tn; tb += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) frame(tb + i, ring[(1 + i) % D], true);
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
    if (tb + i < Tn) frame(tb + i, ring[(1 + i) % D], false);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (s0 + j == S - 1) fin[0] = own[j];
    if (s0 + j == S - 2) fin[1] = own[j];
  }
  if ((gave_up >> 22) & 1) fin[2] = 1.f;   // an entry never arrived (cannot happen while every wave of the workgroup runs)
#undef CTC_FENCE
}

// phase 1 of the pipeline kernel: lpn[t][v] = clamp((x[t, n, v] - logsumexp_v x[t, n, :]) log2(e), +-CTC_NEG) (NaN -> CTC_NEG); a thread takes
// a frame with its VB >= V symbols in registers (every load of the frame in flight at once)
template <int VB>
__device__ __forceinline__ bool alpha_wave_normalise(const float* __restrict__ logits, float* __restrict__ lpn, int n, int Tn,
                                                     int N, int V, int log_probs_in, int tid) {
  constexpr float LOG2E = 1.4426950408889634f;
  // a frame's normaliser is NaN or infinite.  An int in a VGPR, not a bool: a loop-carried bool lives in an SGPR lane mask that
  // hipcc merges per iteration -- the shape tools/isa_lanemask_audit.py watches for; per-lane integer OR needs no mask at all
  int bad = 0;
  for (int t = tid; t < Tn; t += CTC_THREADS) {
    const float* row = logits + ((size_t)t * N + n) * V;
    float* out = lpn + (size_t)t * V;
    float r[VB];
#pragma unroll
    for (int v = 0; v < VB; ++v) r[v] = v < V ? row[v] : neg_inf();
    float lz = 0.f;
    if (!log_probs_in) {
      float m = neg_inf();
#pragma unroll
      for (int v = 0; v < VB; ++v) m = fmaxf(m, r[v]);
      float sum = 0.f;
#pragma unroll
      for (int v = 0; v < VB; ++v) sum += expf(r[v] - m);
      lz = logf(sum) + m;
      bad |= (fabsf(lz) < INFINITY) ? 0 : 1;
    } else {
#pragma unroll
      for (int v = 0; v < VB; ++v) bad |= (r[v] != r[v]) ? 1 : 0;      // (columns >= V hold -inf, not NaN: no `v < V` needed)
    }
#pragma unroll
    for (int v = 0; v < VB; ++v)
      if (v < V) out[v] = fminf(fmaxf((r[v] - lz) * LOG2E, CTC_NEG), -CTC_NEG);
  }
  return bad != 0;
}

// One frame of phase 1 for a wide alphabet, by ONE WAVE (lanes across the symbols): the row's normaliser, the per-state copy
// lps_row[s] of the normalised log2-domain log-probabilities the recursion reads and (ROWS) every symbol's value lpn_row[v] for
// the gradient kernel.  Returns 1 when the normaliser is not finite.  Called by the pipeline kernel's own four waves (phase 1
// inside the launch) and by ctc_normalise_wide_kernel (phase 1 as a chip-wide launch of its own).
template <bool ROWS>
__device__ __forceinline__ int ctc_normalise_wide_frame(const float* __restrict__ row, float* __restrict__ lpn_row,
                                                        float* __restrict__ lps_row, const int32_t* __restrict__ tg, int S, int V,
                                                        int blank, int log_probs_in, int lane) {
  constexpr int NV = 16;
  int bad = 0;
  float lz = 0.f;
  float r[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) r[i] = (lane + 64 * i < V) ? row[lane + 64 * i] : neg_inf();
  if (!log_probs_in) {
    float m = neg_inf();
#pragma unroll
    for (int i = 0; i < NV; ++i) m = fmaxf(m, r[i]);
    for (int v0 = 64 * NV; v0 < V; v0 += 64 * NV) {
      float q[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) q[i] = (v0 + lane + 64 * i < V) ? row[v0 + lane + 64 * i] : neg_inf();
#pragma unroll
      for (int i = 0; i < NV; ++i) m = fmaxf(m, q[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += expf(r[i] - m);
    for (int v0 = 64 * NV; v0 < V; v0 += 64 * NV) {
      float q[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) q[i] = (v0 + lane + 64 * i < V) ? row[v0 + lane + 64 * i] : neg_inf();
#pragma unroll
      for (int i = 0; i < NV; ++i) sum += expf(q[i] - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    lz = logf(sum) + m;
    bad |= (fabsf(lz) < INFINITY) ? 0 : 1;
  }
  if (ROWS) {                                   // every symbol's value: only the gradient kernel reads these
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < V) lpn_row[lane + 64 * i] = fminf(fmaxf((r[i] - lz) * 1.4426950408889634f, CTC_NEG), -CTC_NEG);
    for (int v = 64 * NV + lane; v < V; v += 64)
      lpn_row[v] = fminf(fmaxf((row[v] - lz) * 1.4426950408889634f, CTC_NEG), -CTC_NEG);
  }
  for (int s2 = lane; s2 < S; s2 += 64) {
    const int lab = (s2 & 1) ? tg[s2 >> 1] : blank;
    lps_row[s2] = fminf(fmaxf((row[lab] - lz) * 1.4426950408889634f, CTC_NEG), -CTC_NEG);
  }
  return bad;
}

// Phase 1 of a wide-alphabet loss as its own launch (round 6): a wave per (frame, utterance) over the whole chip.  Inside the
// pipeline kernel the utterance's four waves walked its T frames of V symbols alone -- 10 MB per workgroup at [501, 32, 5000],
// 1.4 ms for the forward, 3.8 .. 4.9 with the backward (tools/ctc_sweep.py).  bad[n] |= 1 where a frame's normaliser is not
// finite (the pipeline kernel reads it in place of its own phase-1 flag: mode bit 1).
template <bool ROWS>
__global__ __launch_bounds__(256) void ctc_normalise_wide_kernel(const float* __restrict__ logits, const int32_t* __restrict__ in_lens,
                                                                 const int32_t* __restrict__ targets,
                                                                 const int32_t* __restrict__ tgt_offsets,
                                                                 const int32_t* __restrict__ tgt_lens, float* __restrict__ lpn_ws,
                                                                 float* __restrict__ lps_ws, int* __restrict__ bad, int T, int N,
                                                                 int V, int S_max, int blank, int log_probs_in) {
  const int lane = threadIdx.x & 63;
  const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // frame-major: consecutive waves read consecutive rows
  if (f >= (long)T * N) return;
  const int t = (int)(f / N), n = (int)(f - (long)t * N);
  if (t >= min(max(in_lens[n], 0), T)) return;
  const int S = 2 * max(tgt_lens[n], 0) + 1;
  const int b = ctc_normalise_wide_frame<ROWS>(logits + (size_t)f * V, ROWS ? lpn_ws + ((size_t)n * T + t) * V : nullptr,
                                               lps_ws + ((size_t)n * T + t) * S_max, targets + tgt_offsets[n], S, V, blank,
                                               log_probs_in, lane);
  if (b && lane == 0) atomicOr(&bad[n], 1);
}

// mode: bit 0 = the reversed recursion (beta), bit 1 = the workspace already holds this call's normalised log-probabilities
// (a second launch of the same backward).  ROWS: rows_ws [N][T][S_max] receives every frame's values; ll2_out [N] the
// log2-domain log-likelihood (CTC_NEG or below: no path).
template <int K, int D, bool ROWS = false>
__global__ __launch_bounds__(CTC_THREADS) void ctc_alpha_wave_kernel(const float* __restrict__ logits,
                                                                     const int32_t* __restrict__ in_lens,
                                                                     const int32_t* __restrict__ targets,
                                                                     const int32_t* __restrict__ tgt_offsets,
                                                                     const int32_t* __restrict__ tgt_lens,
                                                                     float* __restrict__ nll, float* __restrict__ lpn_ws,
                                                                     int T, int N, int V, int blank, int log_probs_in,
                                                                     float* __restrict__ rows_ws = nullptr, int S_max = 0,
                                                                     int mode = 0, float* __restrict__ ll2_out = nullptr,
                                                                     float* __restrict__ rows_ws_rev = nullptr,
                                                                     float* __restrict__ lps_ws = nullptr,
                                                                     unsigned* __restrict__ status = nullptr,
                                                                     const int* __restrict__ bad_in = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
  int bad_row = 0;        // a frame whose normaliser is not finite (a NaN / +inf logit, a row of -inf): the loss is NaN (torch); an int
                          // in a VGPR rather than a loop-carried bool (alpha_wave_normalise)
  MbEntry* mb = reinterpret_cast<MbEntry*>(smem);   // [3][T] {alpha(top), alpha(top - 1)} of waves 0..2, per frame
  float* fin = smem + (size_t)6 * T;                // [2] the last two states' values, [2] = a wave gave up waiting
  const int Tn = min(max(in_lens[n], 0), T);
  const int L = max(tgt_lens[n], 0);
  const int S = 2 * L + 1;
  const int32_t* tg = targets + tgt_offsets[n];
  float* lpn = lpn_ws + (size_t)n * T * V;
  // a grid of (N, 2) runs the forward recursion (y = 0) and the reversed one (y = 1: beta, rows to rows_ws_rev) side by side:
  // both normalise the utterance's log-probabilities into the same workspace rows (the same bits from either)
  if (blockIdx.y == 1) { mode |= 1; rows_ws = rows_ws_rev; ll2_out = nullptr; nll = nullptr; }
  float* rows = ROWS ? rows_ws + (size_t)n * T * S_max : nullptr;
  const int rev = mode & 1;
  const bool wide_v = V > 64;                       // the recursion reads a per-state copy (lps) of the log-probabilities
  float* lps = wide_v ? lps_ws + (size_t)n * T * S_max : nullptr;

  if (mode & 2) {
    if (bad_in != nullptr) bad_row = bad_in[n];        // phase 1 ran as its own launch (ctc_normalise_wide_kernel)
  } else if (V <= 32) bad_row = (int)alpha_wave_normalise<32>(logits, lpn, n, Tn, N, V, log_probs_in, tid);
  else if (V <= 64) bad_row = (int)alpha_wave_normalise<64>(logits, lpn, n, Tn, N, V, log_probs_in, tid);
  else {
    // wide alphabets (word pieces): a WAVE per frame, lanes across the symbols (coalesced row reads, wave reductions), a
    // lane's share of the row in registers with every load of the row in flight at once -- a thread per frame, or one load
    // per loop iteration, waited for each L2 round trip in turn: 1.07 ms per launch at V = 1 000, 0.3 ms now.  Rows beyond
    // 64 x 16 symbols are walked in slabs of that size, the later passes re-reading them.  The recursion then reads a
    // per-STATE copy of the row (lps: a wave's load is 256 contiguous bytes, not 64 cache lines of a 4 KB row)
    const int lane = tid & 63;
    for (int t = w; t < Tn; t += CTC_THREADS / 64)
      bad_row |= ctc_normalise_wide_frame<ROWS>(logits + ((size_t)t * N + n) * V, ROWS ? lpn + (size_t)t * V : nullptr,
                                                lps + (size_t)t * S_max, tg, S, V, blank, log_probs_in, lane);
  }
  for (int i = tid; i < 3 * T; i += CTC_THREADS) mb[i] = MbEntry{__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu)};
  if (tid < 2) fin[tid] = CTC_NEG;
  if (tid == 2) fin[2] = 0.f;
  if (tid == 3) fin[3] = 0.f;
  __syncthreads();   // the workgroup's own stores (one CU, one L1) are visible to its loads behind this
  if (bad_row) fin[3] = 1.f;   // (read behind the barrier that follows the recursion)
  if (Tn == 0) {
    if (tid == 0) {
      if (nll != nullptr) nll[n] = (S == 1) ? -0.f : INFINITY;
      if (ll2_out != nullptr) ll2_out[n] = (S == 1) ? 0.f : CTC_NEG;
    }
    return;
  }
  if (w * 64 * K < S) {   // wave-uniform: a wave without states has nothing downstream of it either
    if (w == 0) alpha_wave_body<K, D, false, true, ROWS>(wide_v ? lps : lpn, tg, fin, mb, T, Tn, V, S, blank, tid, rows, wide_v ? -S_max : S_max, rev);
    else if (w == 3) alpha_wave_body<K, D, true, false, ROWS>(wide_v ? lps : lpn, tg, fin, mb, T, Tn, V, S, blank, tid, rows, wide_v ? -S_max : S_max, rev);
    else alpha_wave_body<K, D, true, true, ROWS>(wide_v ? lps : lpn, tg, fin, mb, T, Tn, V, S, blank, tid, rows, wide_v ? -S_max : S_max, rev);
  }
  __syncthreads();
  if (tid == 0) {
    const float l1 = fin[0], l2 = fin[1];   // log2 domain; CTC_NEG or below = no path; NaN = a wave gave up waiting
    // fin[2]: an LDS mailbox entry never arrived (cannot happen while every wave of the workgroup runs) -- NaN in band AND the
    // sticky status word, which ms_ctc_status turns into MS_ERR_TIMEOUT; fin[3]: a non-finite normaliser -- NaN, as
    // torch.nn.CTCLoss gives for such logits (the clamp to "log zero" would otherwise turn it into +inf, which zero_infinity
    // silently replaces by 0: ADVICE r4)
    if (fin[2] != 0.f && status != nullptr) atomicOr(status, 1u);
    const bool poisoned = fin[2] != 0.f || fin[3] != 0.f;
    const float m = fmaxf(l1, l2);
    const float ll2 = log2f(exp2f(l1 - m) + exp2f(l2 - m)) + m;
    if (nll != nullptr) nll[n] = poisoned ? __uint_as_float(0x7fc00000u) : (m < 0.5f * CTC_NEG) ? INFINITY : -(ll2 * 0.6931471805599453f);
    if (ll2_out != nullptr) ll2_out[n] = poisoned ? __uint_as_float(0x7fc00000u) : (m < 0.5f * CTC_NEG) ? CTC_NEG : ll2;
  }
}

// ---- the gradient rows (round 4).  With alpha and beta of every frame in the workspace (two launches of the pipeline kernel:
// forward, and reversed = beta) a frame's gradient row is independent of every other frame:
//   grad[t, n, k] = go[n] (2^lp2[t, k] - 2^(log2 sum_{s: l'_s = k} 2^(alpha_t(s) + beta_t(s)) - ll2 - lp2[t, k]))
// (everything in the log2 domain the pipeline works in; alpha and beta both contain lp2[t, l'_s]).  One wave per frame: its
// lanes hold the frame's states, the blank's states (every even s) are summed by a wave reduction, a label's few occurrences
// by the lane of that label walking its occurrence list (built once per workgroup in LDS) -- fixed orders, so the result does not
// depend on scheduling.  Sums are taken relative to the frame's maximum of alpha + beta: a bucket more than 2^-126 below it
// contributes nothing, as it does to any float32 result.  ctc_grad_kernel did this with V threads each scanning all S states twice
// between three barriers per frame, serially over the frames: 26 ms at [501, 32, 29] x 120 labels (52 us per frame).
constexpr int GR_FRAMES = 4;    // frames per wave (several waves per SIMD: a frame is a chain of dependent loads and reductions)

__global__ __launch_bounds__(CTC_THREADS) void ctc_grad_rows_kernel(const float* __restrict__ lpn_ws, const float* __restrict__ alpha_ws,
                                                                    const float* __restrict__ beta_ws, const float* __restrict__ ll2,
                                                                    const int32_t* __restrict__ in_lens,
                                                                    const int32_t* __restrict__ targets,
                                                                    const int32_t* __restrict__ tgt_offsets,
                                                                    const int32_t* __restrict__ tgt_lens,
                                                                    const float* __restrict__ grad_nll, float* __restrict__ grad,
                                                                    int T, int N, int V, int S_max, int blank, int zero_infinity) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Tn = min(max(in_lens[n], 0), T);
  const int L = max(tgt_lens[n], 0), S = 2 * L + 1;
  const int32_t* tg = targets + tgt_offsets[n];
  int* occ = reinterpret_cast<int*>(smem);              // [L] label positions grouped by label
  int* start = occ + S_max;                              // [V + 1]
  float* ab = smem + S_max + (V + 1) + (size_t)w * S_max;   // [4][S_max] this wave's alpha + beta row
  // occurrence lists: the positions of label k in the target, in order (a few entries each; built once per workgroup)
  for (int k = tid; k <= V; k += CTC_THREADS) {
    int c = 0;
    if (k < V && k != blank)
      for (int i = 0; i < L; ++i) c += (tg[i] == k) ? 1 : 0;
    start[k] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int c = 0;
    for (int k = 0; k <= V; ++k) { const int cnt = start[k]; start[k] = c; c += cnt; }
  }
  __syncthreads();
  for (int k = tid; k < V; k += CTC_THREADS) {
    if (k == blank) continue;
    int c = start[k];
    for (int i = 0; i < L; ++i)
      if (tg[i] == k) occ[c++] = i;
  }
  __syncthreads();
  const float go = grad_nll[n];
  const float l2 = ll2[n];
  const bool infeasible = l2 < 0.5f * CTC_NEG;
  const float* lpn = lpn_ws + (size_t)n * T * V;
  const float* al = alpha_ws + (size_t)n * T * S_max;
  const float* be = beta_ws + (size_t)n * T * S_max;
  const int t0 = (blockIdx.x * 4 + w) * GR_FRAMES;
  for (int t = t0; t < min(t0 + GR_FRAMES, T); ++t) {
    float* grow = grad + ((size_t)t * N + n) * V;
    if (t >= Tn || (infeasible && zero_infinity)) {       // padding frames (and, with zero_infinity, impossible targets): zeros
      for (int k = lane; k < V; k += 64) grow[k] = 0.f;
      continue;
    }
    if (l2 != l2) {   // a NaN loss (a non-finite log-softmax normaliser, or a wave that gave up): NaN gradients, as torch's are
      for (int k = lane; k < V; k += 64) grow[k] = l2;
      continue;
    }
    // alpha + beta of the frame's states into LDS, their maximum and the blank's sum by wave reductions
    float m = CTC_NEG;
    for (int s = lane; s < S; s += 64) {
      const float v = al[(size_t)t * S_max + s] + be[(size_t)t * S_max + s];
      ab[s] = v;
      m = fmaxf(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // (the row is this wave's own: a wave's LDS operations execute in order, no barrier between its writes and reads)
    float bsum = 0.f;
    for (int s = 2 * lane; s < S; s += 128) bsum += __builtin_amdgcn_exp2f(ab[s] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o, 64);
    const bool dead = m < 0.5f * CTC_NEG;                  // no state of this frame lies on a path (an impossible target)
    for (int k = lane; k < V; k += 64) {
      const float lp2 = lpn[(size_t)t * V + k];
      float res = __builtin_amdgcn_exp2f(lp2);
      if (!dead) {
        float sum = 0.f;
        if (k == blank) sum = bsum;
        else
          for (int i = start[k]; i < start[k + 1]; ++i) sum += __builtin_amdgcn_exp2f(ab[2 * occ[i] + 1] - m);
        if (sum > 0.f) res -= __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(sum) + m - l2 - lp2);
      }
      grow[k] = res * go;
    }
  }
}

size_t alpha_wave_lds(int T) { return ((size_t)6 * T + 4) * sizeof(float); }

// Gradient of the per-utterance losses with respect to the logits (alpha-beta posteriors; loss/ctc_loss.py:95-101 under
// autograd = LogSoftmax backward o torch ctc_loss backward, which at valid frames collapses to one expression):
//   grad[t,n,k] = go[n] * ( softmax(x)[t,k] - exp( logsumexp_{s: l'_s = k}(alpha_t(s) + beta_t(s)) + nll - lp[t,k] ) )
// One workgroup per utterance: the alpha rows go to the global workspace on the way forward, the beta recursion walks
// back with a double buffer in LDS and emits one gradient row per frame; frames t >= input length get zeros.
__global__ __launch_bounds__(CTC_THREADS) void ctc_grad_kernel(const float* __restrict__ logits,
                                                               const int32_t* __restrict__ in_lens,
                                                               const int32_t* __restrict__ targets,
                                                               const int32_t* __restrict__ tgt_offsets,
                                                               const int32_t* __restrict__ tgt_lens,
                                                               const float* __restrict__ grad_nll, float* __restrict__ grad,
                                                               float* __restrict__ logz_ws, float* __restrict__ alpha_ws,
                                                               int T, int N, int V, int S_max, int blank,
                                                               int zero_infinity, int log_probs_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x, tid = threadIdx.x;
  int* ext = reinterpret_cast<int*>(smem);   // [S_max]
  float* buf0 = smem + S_max;                // [S_max] alpha / beta double buffer
  float* buf1 = buf0 + S_max;                // [S_max]
  float* ab = buf1 + S_max;                  // [S_max] alpha_t + beta_t
  __shared__ float nll_s;
  const int Tn = min(max(in_lens[n], 0), T);
  const int L = max(tgt_lens[n], 0);
  const int S = 2 * L + 1;
  float* logz = logz_ws + (size_t)n * T;
  float* alpha = alpha_ws + (size_t)n * T * S_max;
  const int32_t* tg = targets + tgt_offsets[n];
  const float go = grad_nll[n];

  for (int t = tid; t < Tn; t += CTC_THREADS) {
    // CTCLoss(dim != -1): the values are what torch.nn.CTCLoss takes as log-probabilities; its backward still returns
    // exp(lp) - exp(log sum alpha beta + nll - lp) (aten/native/LossCTC.cpp), which is what is written here -- the
    // log-softmax over the other axis is chained behind it by ms_log_softmax_axis_backward
    if (log_probs_in) { logz[t] = 0.f; continue; }
    const float* row = logits + ((size_t)t * N + n) * V;
    float m = neg_inf();
    for (int v = 0; v < V; ++v) m = fmaxf(m, row[v]);
    float sum = 0.f;
    for (int v = 0; v < V; ++v) sum += expf(row[v] - m);
    logz[t] = logf(sum) + m;
  }
  for (int s = tid; s < S; s += CTC_THREADS) {
    ext[s] = (s & 1) ? tg[s >> 1] : blank;
    buf0[s] = neg_inf();
  }
  // padding frames carry no gradient
  for (size_t i = (size_t)Tn * V + tid; i < (size_t)T * V; i += CTC_THREADS) {
    const size_t t = i / V, v = i - t * V;
    grad[(t * N + n) * V + v] = 0.f;
  }
  __syncthreads();
  if (Tn == 0) return;
  {
    const float* row = logits + (size_t)n * V;
    if (tid == 0) buf0[0] = row[blank] - logz[0];
    if (tid == 1 && S > 1) buf0[1] = row[ext[1]] - logz[0];
  }
  __syncthreads();
  float* cur = buf0;
  float* nxt = buf1;
  for (int s = tid; s < S; s += CTC_THREADS) alpha[s] = cur[s];
  for (int t = 1; t < Tn; ++t) {
    const float* row = logits + ((size_t)t * N + n) * V;
    const float lz = logz[t];
    for (int s = tid; s < S; s += CTC_THREADS) {
      const int lab = ext[s];
      const float a0 = cur[s];
      const float a1 = (s >= 1) ? cur[s - 1] : neg_inf();
      const float a2 = (s >= 2 && lab != blank && lab != ext[s - 2]) ? cur[s - 2] : neg_inf();
      const float l = lse3(a0, a1, a2);
      const float v = (l == neg_inf()) ? neg_inf() : l + (row[lab] - lz);
      nxt[s] = v;
      alpha[(size_t)t * S_max + s] = v;
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  if (tid == 0) {
    const float l1 = cur[S - 1];
    const float l2 = (S > 1) ? cur[S - 2] : neg_inf();
    const float m = fmaxf(l1, l2);
    nll_s = (m == neg_inf()) ? INFINITY : -(logf(expf(l1 - m) + expf(l2 - m)) + m);
  }
  __syncthreads();
  const float nll = nll_s;
  if (zero_infinity && isinf(nll)) {
    for (size_t i = tid; i < (size_t)Tn * V; i += CTC_THREADS) {
      const size_t t = i / V, v = i - t * V;
      grad[(t * N + n) * V + v] = 0.f;
    }
    return;
  }
  // beta_{Tn-1}: only the last blank and the last label can end a path
  __syncthreads();
  {
    const float* row = logits + ((size_t)(Tn - 1) * N + n) * V;
    const float lz = logz[Tn - 1];
    for (int s = tid; s < S; s += CTC_THREADS) {
      float v = neg_inf();
      if (s == S - 1 || s == S - 2) v = row[ext[s]] - lz;
      cur[s] = v;
    }
  }
  __syncthreads();
  for (int t = Tn - 1; t >= 0; --t) {
    const float* row = logits + ((size_t)t * N + n) * V;
    const float lz = logz[t];
    if (t < Tn - 1) {
      // beta_t(s) = lp[t, l'_s] + logsumexp(beta_{t+1}(s), beta_{t+1}(s+1), [l'_{s+2} != l'_s] beta_{t+1}(s+2))
      for (int s = tid; s < S; s += CTC_THREADS) {
        const int lab = ext[s];
        const float b0 = nxt[s];
        const float b1 = (s + 1 < S) ? nxt[s + 1] : neg_inf();
        const float b2 = (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != lab) ? nxt[s + 2] : neg_inf();
        const float l = lse3(b0, b1, b2);
        cur[s] = (l == neg_inf()) ? neg_inf() : l + (row[lab] - lz);
      }
      __syncthreads();
    }
    for (int s = tid; s < S; s += CTC_THREADS) ab[s] = alpha[(size_t)t * S_max + s] + cur[s];
    __syncthreads();
    for (int k = tid; k < V; k += CTC_THREADS) {
      float m = neg_inf();
      for (int s = 0; s < S; ++s)
        if (ext[s] == k) m = fmaxf(m, ab[s]);
      const float lp = row[k] - lz;
      float res = expf(lp);
      if (m != neg_inf()) {
        float sum = 0.f;
        for (int s = 0; s < S; ++s)
          if (ext[s] == k) sum += expf(ab[s] - m);
        res -= expf(logf(sum) + m + nll - lp);
      }
      grad[((size_t)t * N + n) * V + k] = res * go;
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;   // beta_t becomes "next" for frame t-1
  }
}

// Deterministic single-workgroup reduction of the per-utterance losses.
__global__ void ctc_reduce_kernel(float* __restrict__ nll, const int32_t* __restrict__ tgt_lens,
                                  float* __restrict__ reduced, int N, int reduction, int zero_infinity) {
  __shared__ float part[256];
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int n = tid; n < N; n += 256) {
    float v = nll[n];
    if (zero_infinity && isinf(v)) {
      v = 0.f;
      nll[n] = 0.f;
    }
    if (reduction == 1) v = v / fmaxf((float)tgt_lens[n], 1.f);
    acc += v;
  }
  part[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) part[tid] += part[tid + o];
    __syncthreads();
  }
  if (tid == 0 && reduced) reduced[0] = (reduction == 1) ? part[0] / (float)N : part[0];
}

// One workgroup (256 threads = 4 waves) per utterance.  Frames are processed in chunks of
// 256: argmax per frame (first maximum wins, NaN counts as maximum like torch.argmax),
// keep = sym != blank && (t == 0 || sym != sym[t-1]); kept symbols are compacted with a
// wave ballot + popcount prefix and a 4-entry cross-wave scan.
// PRE (round 6; alphabets beyond 64 symbols): the arg max of frame t already lies in out_idx[n * T + t]
// (ctc_argmax_wave_kernel) and is compacted in place -- a kept symbol's place is never behind its frame, a chunk's threads
// have all read their frames before the first of them writes, and a chunk writes in front of the next chunk's frames.
// Without it a THREAD walks its frame's row alone: 29 loads for the reference's alphabet, 5 000 dependent ones for a
// 5 000-symbol alphabet (2.0 ms at [501, 32, 5000]; tools/ctc_sweep.py).
template <bool PRE>
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                         int32_t* __restrict__ out_idx, int32_t* __restrict__ out_len,
                                                         int T, int N, int V, int blank) {
  __shared__ int wave_cnt[4];
  __shared__ int wave_last[4];
  __shared__ int last_sym;   // argmax of the last frame of the previous chunk
  __shared__ int base;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = min(max(lens[n], 0), T);
  if (tid == 0) { last_sym = -1; base = 0; }
  __syncthreads();
  for (int t0 = 0; t0 < len; t0 += 256) {
    const int t = t0 + tid;
    int sym = -1;
    if (t < len) {
      if (PRE) {
        sym = out_idx[(size_t)n * T + t];
      } else {
        const float* row = x + ((size_t)t * N + n) * V;
        float best = row[0];
        sym = 0;
        for (int v = 1; v < V; ++v) {
          const float c = row[v];
          if (c > best || (c != c && best == best)) { best = c; sym = v; }
        }
      }
    }
    int prev = __shfl_up(sym, 1, 64);
    // cross-wave / cross-chunk predecessor
    if (lane == 63) wave_last[wave] = sym;
    __syncthreads();
    if (lane == 0) prev = (wave == 0) ? last_sym : wave_last[wave - 1];
    const bool keep = (t < len) && (sym != blank) && (t == 0 || sym != prev);
    const unsigned long long mask = __ballot(keep);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(mask);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) out_idx[(size_t)n * T + off + before] = sym;
    __syncthreads();
    if (tid == 255) last_sym = sym;
    if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) out_len[n] = base;
}

// One wave per (frame, utterance): lanes stride the row (coalesced), every lane keeps the first maximum of its columns
// (NaN counts as the maximum, the first NaN wins: torch.argmax), the lanes' candidates meet in a butterfly under the same
// order -- NaN before numbers, then the larger value, then the lower index.  Result into out_idx[n * T + t].
__global__ __launch_bounds__(256) void ctc_argmax_wave_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                              int32_t* __restrict__ out_idx, int T, int N, int V) {
  const int lane = threadIdx.x & 63;
  const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // frame-major pair index t * N + n
  if (f >= (long)T * N) return;
  const int t = (int)(f / N), n = (int)(f - (long)t * N);
  if (t >= min(max(lens[n], 0), T)) return;
  const float* row = x + (size_t)f * V;
  float best = 0.f;
  int sym = 0x7fffffff;                                          // a lane without a column never wins
  for (int v = lane; v < V; v += 64) {
    const float c = row[v];
    if (sym == 0x7fffffff || c > best || (c != c && best == best)) { best = c; sym = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int os = __shfl_xor(sym, o, 64);
    const bool mine_nan = best != best, other_nan = ob != ob;
    bool take;
    if (os == 0x7fffffff) take = false;
    else if (sym == 0x7fffffff) take = true;
    else if (mine_nan || other_nan) take = other_nan && (!mine_nan || os < sym);
    else take = ob > best || (ob == best && os < sym);
    if (take) { best = ob; sym = os; }
  }
  if (lane == 0) out_idx[(size_t)n * T + t] = sym;
}

}  // namespace

extern "C" size_t ms_ctc_loss_workspace_bytes(int T, int N, int V, int S_max) {
  (void)S_max;
  if (T <= 0 || N <= 0 || V <= 0) return 0;
  // per-frame normalisers [T][N] (ctc_alpha_kernel) + the normalised log-probabilities [N][T][V] of the pipeline kernel
  // (wide alphabets, V > 64: the recursion reads a per-state copy [N][T][S_max] instead)
  return CTC_STATUS_BYTES + ms::align_up((size_t)T * N * sizeof(float), 256) +
         ms::align_up((size_t)T * N * (V > 64 ? std::max(S_max, 1) : V) * sizeof(float), 256);
}

// The first CTC_STATUS_BYTES of a CTC workspace: a sticky word the pipeline kernel sets when a wave gave up waiting for a
// mailbox entry (the loss of that utterance is NaN).  Synchronises `stream`, returns MS_ERR_TIMEOUT once and clears the word.
extern "C" int ms_ctc_status(const void* workspace, void* stream) {
  MS_REQUIRE(workspace, "null pointer");
  unsigned v = 0;
  MS_HIP(hipMemcpyAsync(&v, workspace, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
  MS_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (v == 0) return MS_OK;
  MS_HIP(hipMemsetAsync(const_cast<void*>(workspace), 0, sizeof(unsigned), (hipStream_t)stream));
  ms::set_error("CTC loss: a wave of the alpha pipeline gave up waiting for its neighbour's frame (0.5 s); that utterance's loss is NaN");
  return MS_ERR_TIMEOUT;
}

extern "C" int ms_ctc_loss_forward(const float* logits, const int32_t* in_lens, const int32_t* targets,
                                   const int32_t* tgt_offsets, const int32_t* tgt_lens, float* nll, float* reduced,
                                   int T, int N, int V, int S_max, int blank, int reduction, int zero_infinity,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  MS_REQUIRE(logits && in_lens && targets && tgt_offsets && tgt_lens && nll && workspace, "null pointer");
  MS_REQUIRE(T > 0 && N > 0 && V > 0 && S_max >= 1, "bad shape");
  MS_REQUIRE(blank >= 0 && blank < V, "blank out of range");
  MS_REQUIRE(reduction >= 0 && reduction <= 2, "reduction must be 0 (none), 1 (mean) or 2 (sum)");
  MS_REQUIRE(reduction == 0 || reduced, "reduced output required");
  MS_REQUIRE(zero_infinity >= 0 && zero_infinity <= (1 | MS_CTC_LOG_PROBS_IN), "zero_infinity takes 0 / 1, optionally OR-ed with MS_CTC_LOG_PROBS_IN");
  if (workspace_bytes < ms_ctc_loss_workspace_bytes(T, N, V, S_max)) {
    ms::set_error("ms_ctc_loss_forward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  unsigned* status = (unsigned*)workspace;
  workspace = (char*)workspace + CTC_STATUS_BYTES;
  const size_t lds = (size_t)3 * S_max * sizeof(float);
  MS_REQUIRE(lds <= 160 * 1024, "target too long for the LDS-resident alpha rows");
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_wave_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_wave_kernel<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_wave_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  const int lpi = (zero_infinity & MS_CTC_LOG_PROBS_IN) ? 1 : 0;
  // the four-wave pipeline: alphas in registers (targets of <= 511 labels), the per-frame mailbox in LDS (T <= ~5 800 frames);
  // MS_CTC_WAVE=0 (read per call: an A/B switch for the tests) keeps the LDS-row kernel
  const char* we = getenv("MS_CTC_WAVE");
  const bool wave_ok = !(we && we[0] == '0') && S_max <= 1024 && alpha_wave_lds(T) <= 160 * 1024 &&
                       (size_t)T * V * sizeof(float) < (1ull << 31);
  float* lpn_ws = (float*)((char*)workspace + ms::align_up((size_t)T * N * sizeof(float), 256));
  if (wave_ok) {
    const size_t wl = alpha_wave_lds(T);
    // wide alphabets: phase 1 (every frame's log-softmax and the per-state copy of it) as a chip-wide launch of its own; the
    // per-utterance "a normaliser was not finite" words live where ctc_alpha_kernel keeps its normalisers (unused on this path).
    // MS_CTC_WIDE_PHASE1=0 (read per call) keeps phase 1 inside the pipeline kernel (A/B runs)
    int mode = 0;
    int* bad = nullptr;
    const char* pe = getenv("MS_CTC_WIDE_PHASE1");
    if (V > 64 && !(pe && pe[0] == '0')) {
      bad = (int*)workspace;
      MS_HIP(hipMemsetAsync(bad, 0, (size_t)N * sizeof(int), (hipStream_t)stream));
      const long pairs = (long)T * N;
      hipLaunchKernelGGL((ctc_normalise_wide_kernel<false>), dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits,
                         in_lens, targets, tgt_offsets, tgt_lens, (float*)nullptr, lpn_ws, bad, T, N, V, S_max, blank, lpi);
      MS_LAUNCH_CHECK();
      mode = 2;
    }
    if (S_max <= 256)
      hipLaunchKernelGGL((ctc_alpha_wave_kernel<1, 16>), dim3(N), dim3(CTC_THREADS), wl, (hipStream_t)stream, logits, in_lens,
                         targets, tgt_offsets, tgt_lens, nll, lpn_ws, T, N, V, blank, lpi, (float*)nullptr, S_max, mode, (float*)nullptr,
                         (float*)nullptr, lpn_ws, status, (const int*)bad);
    else if (S_max <= 512)
      hipLaunchKernelGGL((ctc_alpha_wave_kernel<2, 16>), dim3(N), dim3(CTC_THREADS), wl, (hipStream_t)stream, logits, in_lens,
                         targets, tgt_offsets, tgt_lens, nll, lpn_ws, T, N, V, blank, lpi, (float*)nullptr, S_max, mode, (float*)nullptr,
                         (float*)nullptr, lpn_ws, status, (const int*)bad);
    else
      hipLaunchKernelGGL((ctc_alpha_wave_kernel<4, 8>), dim3(N), dim3(CTC_THREADS), wl, (hipStream_t)stream, logits, in_lens,
                         targets, tgt_offsets, tgt_lens, nll, lpn_ws, T, N, V, blank, lpi, (float*)nullptr, S_max, mode, (float*)nullptr,
                         (float*)nullptr, lpn_ws, status, (const int*)bad);
  } else {
    hipLaunchKernelGGL(ctc_alpha_kernel, dim3(N), dim3(CTC_THREADS), lds, (hipStream_t)stream, logits, in_lens, targets,
                       tgt_offsets, tgt_lens, nll, (float*)workspace, T, N, V, S_max, blank, lpi);
  }
  MS_LAUNCH_CHECK();
  zero_infinity &= 1;
  if (reduction != 0 || zero_infinity) {
    hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, tgt_lens, reduced, N,
                       reduction, zero_infinity);
    MS_LAUNCH_CHECK();
  }
  return MS_OK;
}

namespace {
// y[o, a, i] = x[o, a, i] - logsumexp_a x[o, :, i] on a contiguous [outer, axis, inner] view: one thread per (o, i),
// consecutive threads = consecutive i (coalesced for every a)
__global__ void log_softmax_axis_kernel(const float* __restrict__ x, float* __restrict__ y, int outer, int axis, int inner) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)outer * inner) return;
  const int o = (int)(id / inner), i = (int)(id % inner);
  const float* xp = x + (size_t)o * axis * inner + i;
  float* yp = y + (size_t)o * axis * inner + i;
  float m = -INFINITY;
  for (int a = 0; a < axis; ++a) m = fmaxf(m, xp[(size_t)a * inner]);
  float sum = 0.f;
  for (int a = 0; a < axis; ++a) sum += expf(xp[(size_t)a * inner] - m);
  const float lz = logf(sum) + m;
  for (int a = 0; a < axis; ++a) yp[(size_t)a * inner] = xp[(size_t)a * inner] - lz;
}

// backward of the pass above: gx[o, a, i] = g[o, a, i] - exp(y[o, a, i]) * sum_a g[o, :, i]   (y = the log-probabilities)
__global__ void log_softmax_axis_backward_kernel(const float* __restrict__ y, const float* g, float* gx, int outer, int axis,
                                                 int inner) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)outer * inner) return;
  const int o = (int)(id / inner), i = (int)(id % inner);
  const size_t base = (size_t)o * axis * inner + i;
  float sum = 0.f;
  for (int a = 0; a < axis; ++a) sum += g[base + (size_t)a * inner];
  for (int a = 0; a < axis; ++a) {
    const size_t j = base + (size_t)a * inner;
    gx[j] = g[j] - expf(y[j]) * sum;
  }
}
}  // namespace

extern "C" int ms_log_softmax_axis_backward(const float* y, const float* g, float* gx, int outer, int axis, int inner,
                                            void* stream) {
  MS_REQUIRE(y && g && gx, "null pointer");
  MS_REQUIRE(outer > 0 && axis > 0 && inner > 0, "bad shape");
  const long n = (long)outer * inner;
  hipLaunchKernelGGL(log_softmax_axis_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, g,
                     gx, outer, axis, inner);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_log_softmax_axis(const float* x, float* y, int outer, int axis, int inner, void* stream) {
  MS_REQUIRE(x && y, "null pointer");
  MS_REQUIRE(outer > 0 && axis > 0 && inner > 0, "bad shape");
  const long n = (long)outer * inner;
  hipLaunchKernelGGL(log_softmax_axis_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, outer,
                     axis, inner);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_ctc_greedy_decode(const float* x, const int32_t* lens, int32_t* out_idx, int32_t* out_len, int T,
                                    int N, int V, int blank, void* stream) {
  ms::ProfScope prof_span(MS_PROF_GREEDY, (hipStream_t)stream);
  MS_REQUIRE(x && lens && out_idx && out_len, "null pointer");
  MS_REQUIRE(T > 0 && N > 0 && V > 0, "bad shape");
  if (V > 64) {
    const long pairs = (long)T * N;
    MS_REQUIRE((pairs + 3) / 4 <= 0x7fffffffL, "T * N exceeds the grid limit");
    hipLaunchKernelGGL(ctc_argmax_wave_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, lens, out_idx,
                       T, N, V);
    MS_LAUNCH_CHECK();
    hipLaunchKernelGGL(ctc_greedy_kernel<true>, dim3(N), dim3(256), 0, (hipStream_t)stream, x, lens, out_idx, out_len, T, N, V,
                       blank);
  } else {
    hipLaunchKernelGGL(ctc_greedy_kernel<false>, dim3(N), dim3(256), 0, (hipStream_t)stream, x, lens, out_idx, out_len, T, N, V,
                       blank);
  }
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" size_t ms_ctc_loss_backward_workspace_bytes(int T, int N, int V, int S_max) {
  if (T <= 0 || N <= 0 || S_max < 1 || V <= 0) return 0;
  // normalisers [T][N] + alpha rows [N][T][S_max] (ctc_grad_kernel); + beta rows, the normalised log-probabilities [N][T][V] and
  // the log-likelihoods [N] of the pipeline path
  return CTC_STATUS_BYTES + ms::align_up((size_t)T * N * sizeof(float), 256) + (V > 64 ? 3 : 2) * ms::align_up((size_t)T * N * S_max * sizeof(float), 256) +
         ms::align_up((size_t)T * N * V * sizeof(float), 256) + ms::align_up((size_t)N * sizeof(float), 256);
}

extern "C" int ms_ctc_loss_backward(const float* logits, const int32_t* in_lens, const int32_t* targets,
                                    const int32_t* tgt_offsets, const int32_t* tgt_lens, const float* grad_nll,
                                    float* grad_logits, int T, int N, int V, int S_max, int blank, int zero_infinity,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  MS_REQUIRE(logits && in_lens && targets && tgt_offsets && tgt_lens && grad_nll && grad_logits && workspace, "null pointer");
  MS_REQUIRE(T > 0 && N > 0 && V > 0 && S_max >= 1, "bad shape");
  MS_REQUIRE(blank >= 0 && blank < V, "blank out of range");
  MS_REQUIRE(zero_infinity >= 0 && zero_infinity <= (1 | MS_CTC_LOG_PROBS_IN), "zero_infinity takes 0 / 1, optionally OR-ed with MS_CTC_LOG_PROBS_IN");
  if (workspace_bytes < ms_ctc_loss_backward_workspace_bytes(T, N, V, S_max)) {
    ms::set_error("ms_ctc_loss_backward: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  unsigned* status = (unsigned*)workspace;
  workspace = (char*)workspace + CTC_STATUS_BYTES;
  {
    // the pipeline path: alpha rows and beta rows (the same kernel on the reversed utterance) in one launch, then one wave per
    // frame for the gradient rows.  MS_CTC_WAVE=0 (read per call) keeps ctc_grad_kernel.
    const char* we = getenv("MS_CTC_WAVE");
    const size_t rows_lds = ((size_t)5 * S_max + V + 1) * sizeof(float);
    if (!(we && we[0] == '0') && S_max <= 1024 && alpha_wave_lds(T) <= 160 * 1024 && rows_lds <= 64 * 1024 &&
        (size_t)T * std::max(V, S_max) * sizeof(float) < (1ull << 31)) {
      static ms::DeviceOnce once;
      if (once.need()) {
        MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_wave_kernel<1, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_wave_kernel<2, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MS_HIP(hipFuncSetAttribute((const void*)ctc_alpha_wave_kernel<4, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MS_HIP(hipFuncSetAttribute((const void*)ctc_grad_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        once.done();
      }
      char* wsb = (char*)workspace;
      const size_t rows_bytes = ms::align_up((size_t)T * N * S_max * sizeof(float), 256);
      float* alpha_rows = (float*)(wsb + ms::align_up((size_t)T * N * sizeof(float), 256));
      float* beta_rows = (float*)((char*)alpha_rows + rows_bytes);
      float* lpn = (float*)((char*)beta_rows + rows_bytes);
      float* ll2 = (float*)((char*)lpn + ms::align_up((size_t)T * N * V * sizeof(float), 256));
      float* lps = (float*)((char*)ll2 + ms::align_up((size_t)N * sizeof(float), 256));     // wide alphabets only
      const int lpi = (zero_infinity & MS_CTC_LOG_PROBS_IN) ? 1 : 0;
      const size_t wl = alpha_wave_lds(T);
      hipStream_t st = (hipStream_t)stream;
      // wide alphabets: phase 1 as a chip-wide launch (see ms_ctc_loss_forward); here it also writes every symbol's value
      int mode = 0;
      int* bad = nullptr;
      const char* pe = getenv("MS_CTC_WIDE_PHASE1");
      if (V > 64 && !(pe && pe[0] == '0')) {
        bad = (int*)wsb;
        MS_HIP(hipMemsetAsync(bad, 0, (size_t)N * sizeof(int), st));
        const long pairs = (long)T * N;
        hipLaunchKernelGGL((ctc_normalise_wide_kernel<true>), dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, st, logits, in_lens,
                           targets, tgt_offsets, tgt_lens, lpn, lps, bad, T, N, V, S_max, blank, lpi);
        MS_LAUNCH_CHECK();
        mode = 2;
      }
      // alpha and beta (= the reversed recursion) side by side: grid (N, 2)
      if (S_max <= 256)
        hipLaunchKernelGGL((ctc_alpha_wave_kernel<1, 16, true>), dim3(N, 2), dim3(CTC_THREADS), wl, st, logits, in_lens, targets,
                           tgt_offsets, tgt_lens, (float*)nullptr, lpn, T, N, V, blank, lpi, alpha_rows, S_max, mode, ll2, beta_rows, lps, status, (const int*)bad);
      else if (S_max <= 512)
        hipLaunchKernelGGL((ctc_alpha_wave_kernel<2, 16, true>), dim3(N, 2), dim3(CTC_THREADS), wl, st, logits, in_lens, targets,
                           tgt_offsets, tgt_lens, (float*)nullptr, lpn, T, N, V, blank, lpi, alpha_rows, S_max, mode, ll2, beta_rows, lps, status, (const int*)bad);
      else
        hipLaunchKernelGGL((ctc_alpha_wave_kernel<4, 8, true>), dim3(N, 2), dim3(CTC_THREADS), wl, st, logits, in_lens, targets,
                           tgt_offsets, tgt_lens, (float*)nullptr, lpn, T, N, V, blank, lpi, alpha_rows, S_max, mode, ll2, beta_rows, lps, status, (const int*)bad);
      MS_LAUNCH_CHECK();
      hipLaunchKernelGGL(ctc_grad_rows_kernel, dim3(ms::cdiv(T, 4 * GR_FRAMES), N), dim3(CTC_THREADS), rows_lds, st, lpn, alpha_rows,
                         beta_rows, ll2, in_lens, targets, tgt_offsets, tgt_lens, grad_nll, grad_logits, T, N, V, S_max, blank,
                         zero_infinity & 1);
      MS_LAUNCH_CHECK();
      return MS_OK;
    }
  }
  const size_t lds = (size_t)4 * S_max * sizeof(float);
  constexpr size_t GRAD_LDS_MAX = 160 * 1024 - 256;   // the kernel also has a static word
  MS_REQUIRE(lds <= GRAD_LDS_MAX, "target too long for the LDS-resident alpha / beta rows");
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)ctc_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRAD_LDS_MAX));
    attr_once.done();
  }
  float* logz = (float*)workspace;
  float* alpha = (float*)((char*)workspace + ms::align_up((size_t)T * N * sizeof(float), 256));
  hipLaunchKernelGGL(ctc_grad_kernel, dim3(N), dim3(CTC_THREADS), lds, (hipStream_t)stream, logits, in_lens, targets,
                     tgt_offsets, tgt_lens, grad_nll, grad_logits, logz, alpha, T, N, V, S_max, blank, zero_infinity & 1,
                     (zero_infinity & MS_CTC_LOG_PROBS_IN) ? 1 : 0);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
