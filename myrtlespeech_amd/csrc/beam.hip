// CTC prefix beam search (post_process/ctc_beam_decoder.py:175-258), bit-exact beam.
//
// The reference walks, per frame, every beam prefix l (in beam order) and every symbol c
// (in index order) and accumulates into two dict-of-Counter tables Pb[t], Pnb[t] in linear
// float32.  Observations that make it data-parallel without changing a single bit:
//   * every (prefix, table) entry receives at most two addends per frame, and fl(a+b) is
//     commutative, so values do not depend on visiting order -- only the *insertion order*
//     of keys does (it breaks ties in the stable sort and orders Counter.__add__);
//   * insertion order is a pure function of the task index (w*V + c) and the position of
//     the update inside the loop body, so every candidate carries an explicit order key;
//   * candidates are exactly: the beam entries themselves (slot w) and their one-symbol
//     extensions (slot W + w*V + c), an extension that is itself in the beam folds into
//     that beam entry's slot.
// Prefix identity is a trie (node = parent + symbol).  Round 5: the part of it the search READS lives in LDS -- for every
// prefix in the beam a row of its V children {node id, the child's child-table slot, where last frame's tables hold the
// child's Pb / Pnb} (232 entries at W = 8, V = 29), double-buffered across frames and carried over by the frame end for the
// survivors -- so a frame has no dependent global load (it had two per candidate: child table -> node stamp, 4.6 us per
// frame).  The workspace keeps what a back-trace needs (parent, symbol, length) and, for every node that has ever been in
// the beam, a copy of its child row (written through, read again only when a prefix that had left the beam comes back):
// "l + c" resolves to the same node no matter when it is re-derived (`l_plus in A_prev`, ctc_beam_decoder.py:232-241).
// Nodes are made lazily: only an extension that ENTERS the beam gets one (it needs an identity from then on); a candidate
// that is merely present in a frame's tables is found again through its parent's row (`lc_tidx` = its slot in those tables).
//
// One workgroup (256 threads) per utterance, frames sequential.  Compiled with
// -ffp-contract=off: the reference rounds after every multiply and every add.
#include <algorithm>

#include "common.h"

namespace {

constexpr int HDR_INTS = 16;
enum { F_PRESENT = 1, F_KEPT = 2 };
constexpr size_t BEAM_LDS_LIMIT = 150 * 1024;

// bytes of the per-utterance working arrays (LDS; the workspace's scratch region when they do not fit)
size_t beam_lds_bytes(int V, int W) {
  const size_t M = (size_t)W * (V + 1), WV = (size_t)W * V;
  return ((size_t)V + 10 * M + 6 * WV + 17 * (size_t)W + 2 * (size_t)((W + 7) / 8 * 8) + 8) * 4;      // (bm_node: padded to eights)
}

struct BeamLayout {
  size_t per_utt;  // bytes
  size_t hdr, beam_node, beam_pb, beam_pnb, tbl_pb, tbl_pnb, lc_node, lc_cslot, lc_tidx, node_parent, node_sym, node_len, node_nw,
      node_cslot, childtab, scratch;
  int NN, CS, M, big;
};

BeamLayout beam_layout(int T, int V, int W) {
  BeamLayout L;
  L.M = W * (V + 1);
  L.NN = 2 + T * W;          // a frame makes at most W nodes (the extensions that enter the beam)
  L.CS = 2 + T * W + W;
  L.big = beam_lds_bytes(V, W) > BEAM_LDS_LIMIT ? 1 : 0;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += ms::align_up(bytes, 64); return r; };
  L.hdr = take(HDR_INTS * 4);
  L.beam_node = take((size_t)W * 4);
  L.beam_pb = take((size_t)W * 4);
  L.beam_pnb = take((size_t)W * 4);
  L.tbl_pb = take((size_t)L.M * 4);              // the last processed frame's tables (a call may end between frames)
  L.tbl_pnb = take((size_t)L.M * 4);
  L.lc_node = take((size_t)W * V * 4);           // ... and the live beam's child rows
  L.lc_cslot = take((size_t)W * V * 4);
  L.lc_tidx = take((size_t)W * V * 4);
  L.node_parent = take((size_t)L.NN * 4);
  L.node_sym = take((size_t)L.NN * 4);
  L.node_len = take((size_t)L.NN * 4);
  L.node_nw = take((size_t)L.NN * 4);
  L.node_cslot = take((size_t)L.NN * 4);
  L.childtab = take((size_t)L.CS * V * 4);
  L.scratch = take(L.big ? beam_lds_bytes(V, W) : 0);
  L.per_utt = ms::align_up(o, 256);
  return L;
}

struct BeamP {
  const float* probs;
  const int32_t* lens;
  int32_t* out_idx;
  int32_t* out_len;
  const float* word_factor;
  const float* lm_factor;
  int32_t* beam_len_out;
  int32_t* beam_idx_out;
  int32_t* beam_plen_out;
  char* ws;
  BeamLayout L;
  int T, N, V, W, blank, sep, t_begin, t_end, finish, stamps;
  float thr;
};

// MS_PIN(x): an empty asm that "modifies" x -- the load that produced x is issued where it is written.  hipcc otherwise sinks an
// LDS load behind the branch that uses its value, which turns a phase's independent loads into a chain of round trips on a
// workgroup that has one wave per SIMD (S4a: 0.79 -> 0.55 us per frame with nothing else changed)
#define MS_PIN(x) asm volatile("" : "+v"(x))

// beam index of `node` (-1: not in the beam): the beam's node ids in batches of eight independent LDS reads.  The array is
// padded to a multiple of eight entries and every entry that is not a beam member holds -2 (never a node id, never the "no
// child" -1), so the scan needs neither a bound per element nor the beam's size
__device__ __forceinline__ int find_in_beam(const int* bmn, int W8, int node) {
  int at = -1;
  for (int k0 = 0; k0 < W8; k0 += 8) {
    int v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = bmn[k0 + m];
#pragma unroll
    for (int m = 0; m < 8; ++m) MS_PIN(v[m]);
#pragma unroll
    for (int m = 0; m < 8; ++m) at = v[m] == node ? k0 + m : at;
  }
  return at;
}

// NT threads: 256.  ONE wave (NT = 64, MS_BEAM_THREADS=64: a lane takes four candidates at W = 8, V = 29, no cross-wave barrier)
// was measured and is SLOWER -- 3.63 against 2.31 ms for 32 x 501 frames: a frame is ~70 exposed LDS round trips in seven
// short phases (hipcc waits before every dependent use), and four candidates per lane lengthen every phase more than the
// barriers cost.  The instantiation stays as the A/B switch.
// BIG: beam_width * (alphabet + 1) too large for the LDS (~2 300 candidates; 84 prefixes at V = 29) -- the same code
// with its working arrays in the workspace.  Slow (every phase goes through L2) but any width up to 256 decodes.
// MS_BEAM_STAMPS=1 (read per call): thread 0 of utterance 0 accumulates the 100 MHz wall clock per barrier-separated phase of
// the frame loop into hdr[4 .. 4 + 8] (ticks summed over the call's frames; tools/beam_stamps.py prints them)
#define MS_BEAM_STAMP_BEGIN() do { if (stamping) st_prev = wall_clock64(); } while (0)
#define MS_BEAM_STAMP(k) do { if (stamping) { const unsigned long long now_ = wall_clock64(); st_acc[k] += (unsigned)(now_ - st_prev); st_prev = now_; } } while (0)

// VC / WC (round 6): the alphabet size and the beam width as COMPILE-TIME constants (0 = taken from the arguments).  Every
// working array is at an offset from the LDS base that depends on V and W only; with both known the ~40 array pointers the
// frame loop keeps live (106 SGPRs, part of them spilled to VGPR lanes and fetched with v_readlane: DESIGN 9.3 of round 5)
// become immediate offsets of the LDS instructions, and the divisions by V constant multiplications.  Instantiated for the
// reference's decode size -- 29 symbols (configs/deep_speech_2_en.config), width 8; the arithmetic is untouched.
template <bool BIG, int NT, int VC = 0, int WC = 0>
__global__ __launch_bounds__(NT) void beam_kernel(BeamP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, n = blockIdx.x;
  const int V = VC ? VC : p.V, W = WC ? WC : p.W, M = (VC && WC) ? WC * (VC + 1) : p.L.M, WV = W * V;
  char* u = p.ws + (size_t)n * p.L.per_utt;
  // ---- working arrays
  char* arr = BIG ? (u + p.L.scratch) : smem_raw;
  float* prow = reinterpret_cast<float*>(arr);                // [V]
  float* c_pb = prow + V;                                     // [2][M] candidate tables Pb / Pnb of this and the previous frame
  float* c_pnb = c_pb + 2 * M;                                //        (ctc_beam_decoder.py:182-192), indexed by candidate slot
  float* c_s = c_pnb + 2 * M;                                 // [M]
  float* k_score = c_s + M;                                   // [M] the kept candidates' (score, order key) pairs, compacted
  int* c_key = reinterpret_cast<int*>(k_score + M);           // [M]
  int* c_flags = c_key + M;
  int* k_idx = c_flags + M;
  int* k_key = k_idx + M;
  int* lc_node = k_key + M;                                   // [2][W][V] child rows of the live beam: node id (-1: none yet)
  int* lc_cslot = lc_node + 2 * WV;                           //           the child's child-table slot (-1: never in the beam)
  int* lc_tidx = lc_cslot + 2 * WV;                           //           its slot in the previous frame's tables (-1: absent)
  const int W8 = (W + 7) / 8 * 8;
  int* bm_node = lc_tidx + 2 * WV;                            // [2][W8]: the beam's nodes, this frame's and the next one's (-2 = no entry)
  int* bm_last = bm_node + 2 * W8;                            // [2][W] each
  int* bm_len = bm_last + 2 * W;
  int* bm_nw = bm_len + 2 * W;
  int* bm_cslot = bm_nw + 2 * W;
  float* bm_pb = reinterpret_cast<float*>(bm_cslot + 2 * W);
  float* bm_pnb = bm_pb + 2 * W;
  int* par_present = reinterpret_cast<int*>(bm_pnb + 2 * W);  // [W] each
  float* par_val = reinterpret_cast<float*>(par_present + W);
  int* par_rank = reinterpret_cast<int*>(par_val + W);
  int* newbeam = par_rank + W;
  int* ent_fresh = newbeam + W;
  int* sh = ent_fresh + W;  // [0]=B, [1]=k_count, [2]=n_nodes, [3]=n_cslots

  // ---- global state of this utterance
  int* hdr = reinterpret_cast<int*>(u + p.L.hdr);
  int* g_beam_node = reinterpret_cast<int*>(u + p.L.beam_node);
  float* g_beam_pb = reinterpret_cast<float*>(u + p.L.beam_pb);
  float* g_beam_pnb = reinterpret_cast<float*>(u + p.L.beam_pnb);
  float* tbl_pb = reinterpret_cast<float*>(u + p.L.tbl_pb);
  float* tbl_pnb = reinterpret_cast<float*>(u + p.L.tbl_pnb);
  int* g_lc_node = reinterpret_cast<int*>(u + p.L.lc_node);
  int* g_lc_cslot = reinterpret_cast<int*>(u + p.L.lc_cslot);
  int* g_lc_tidx = reinterpret_cast<int*>(u + p.L.lc_tidx);
  int* node_parent = reinterpret_cast<int*>(u + p.L.node_parent);
  int* node_sym = reinterpret_cast<int*>(u + p.L.node_sym);
  int* node_len = reinterpret_cast<int*>(u + p.L.node_len);
  int* node_nw = reinterpret_cast<int*>(u + p.L.node_nw);
  int* node_cslot = reinterpret_cast<int*>(u + p.L.node_cslot);
  int* childtab = reinterpret_cast<int*>(u + p.L.childtab);

  // a thread's first (usually only) candidate / row entry is (tid / V, tid % V) in every frame: the integer division -- ~40
  // instructions each in S1 and S6 -- is done once
  const int w_first = tid / V, c_first = tid - w_first * V;
  const bool stamping = p.stamps != 0 && tid == 0 && n == 0;
  unsigned long long st_prev = 0;
  unsigned st_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = tid; k < 2 * W8; k += NT) bm_node[k] = -2;     // (members are written below, behind a barrier)
  __syncthreads();
  const int b0 = p.t_begin & 1;      // buffer of the beam / child rows a frame reads = the frame's parity
  if (p.t_begin == 0) {
    // Pb[-1][()] = 1, Pnb[-1][()] = 0, A_prev = [()]   (ctc_beam_decoder.py:182-192)
    if (tid == 0) {
      node_parent[0] = -1; node_sym[0] = -1; node_len[0] = 0; node_nw[0] = 0; node_cslot[0] = 0;
      bm_node[b0 * W8] = 0; bm_pb[b0 * W] = 1.0f; bm_pnb[b0 * W] = 0.0f;
      bm_last[b0 * W] = -1; bm_len[b0 * W] = 0; bm_nw[b0 * W] = 0; bm_cslot[b0 * W] = 0;
      sh[0] = 1; sh[2] = 1; sh[3] = 1;
    }
    for (int v = tid; v < V; v += NT) {
      childtab[v] = -1;
      lc_node[b0 * WV + v] = -1; lc_cslot[b0 * WV + v] = -1; lc_tidx[b0 * WV + v] = -1;
    }
  } else {
    if (tid == 0) { sh[0] = hdr[2]; sh[2] = hdr[0]; sh[3] = hdr[1]; }
    const int B_in = hdr[2];
    for (int w = tid; w < W; w += NT)
      if (w < B_in) {
        const int nd = g_beam_node[w];
        bm_node[b0 * W8 + w] = nd; bm_pb[b0 * W + w] = g_beam_pb[w]; bm_pnb[b0 * W + w] = g_beam_pnb[w];
        bm_last[b0 * W + w] = node_sym[nd]; bm_len[b0 * W + w] = node_len[nd]; bm_nw[b0 * W + w] = node_nw[nd];
        bm_cslot[b0 * W + w] = node_cslot[nd];
      }
    for (int i = tid; i < M; i += NT) { c_pb[(b0 ^ 1) * M + i] = tbl_pb[i]; c_pnb[(b0 ^ 1) * M + i] = tbl_pnb[i]; }
    for (int i = tid; i < WV; i += NT) {
      lc_node[b0 * WV + i] = g_lc_node[i]; lc_cslot[b0 * WV + i] = g_lc_cslot[i]; lc_tidx[b0 * WV + i] = g_lc_tidx[i];
    }
  }
  __syncthreads();

  const int len = min(max(p.lens[n], 0), p.T);
  const int t_stop = min(p.t_end, len);
  // the frame's probabilities do not depend on the search: thread v holds p[t + 1][v] a frame ahead (alphabets beyond 256
  // symbols fetch the rest in the frame itself)
  float p_next = 0.f;
  if (p.t_begin < t_stop && tid < V) p_next = p.probs[((size_t)p.t_begin * p.N + n) * V + tid];
  int t = p.t_begin;
  for (; t < t_stop; ++t) {
    MS_BEAM_STAMP_BEGIN();
    const int B = sh[0];
    if (B == 0) break;  // an empty beam stays empty (ctc_beam_decoder.py:258)
    const int cp = t & 1, pp = cp ^ 1;          // tables: this frame's / the previous frame's
    const int cb = cp, nb = pp;                 // beam + child rows: read / built for the next frame
    const int* bmn = bm_node + cb * W8;
    const int* bml = bm_last + cb * W;
    const int* bmlen = bm_len + cb * W;
    const int* bmnw = bm_nw + cb * W;
    const int* bmcs = bm_cslot + cb * W;
    const float* bmpb = bm_pb + cb * W;
    const float* bmpnb = bm_pnb + cb * W;
    int* lcn = lc_node + cb * WV;
    int* lcc = lc_cslot + cb * WV;
    const int* lct = lc_tidx + cb * WV;
    const float* row = p.probs + ((size_t)t * p.N + n) * V;
    if (tid < V) prow[tid] = p_next;
    for (int v = tid + NT; v < V; v += NT) prow[v] = row[v];
    if (t + 1 < t_stop && tid < V) p_next = p.probs[((size_t)(t + 1) * p.N + n) * V + tid];
    for (int w = tid; w < W; w += NT) par_present[w] = 0;
    if (tid == 0) sh[1] = 0;
    __syncthreads();
    MS_BEAM_STAMP(0);
    // (the blank's probability is the same word for every lane: taken into an SGPR, so that `p_blank <= thr` is a scalar
    // compare and not a lane mask evaluated under one loop's EXEC and read under another's -- tools/isa_lanemask_audit.py)
    const float p_blank = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(prow[p.blank])));

    // (S2's words that do not depend on S1 are requested here, ahead of S1 and its barrier)
    const int s2_w = tid;                                   // W <= NT is not assumed: the S2 loop below re-reads for w >= NT
    float s2_pb = 0.f, s2_pnb = 0.f, s2_plast = 0.f;
    int s2_last = 0, s2_len = 0;
    if (s2_w < B) {
      s2_pb = bmpb[s2_w]; s2_pnb = bmpnb[s2_w]; s2_last = bml[s2_w]; s2_len = bmlen[s2_w];
      s2_plast = prow[max(s2_last, 0)];
    }

    // ---- S1: extensions l + c.  Every LDS word a candidate may need is requested before the first is looked at (the phase is
    // a chain of LDS round trips on one wave per SIMD: with the loads behind the branches that need them hipcc waited five
    // times where two suffice); the addresses are valid for every candidate, the values are used under the same conditions.
    for (int i = tid; i < B * V; i += NT) {
      const int w = i == tid ? w_first : i / V, c = i == tid ? c_first : i - (i / V) * V;
      const int slot = W + i;
      float pc = prow[c];
      int child = lcn[i];
      int ti = lct[i];
      int len_w = bmlen[w], last_w = bml[w];
      MS_PIN(pc); MS_PIN(child); MS_PIN(ti); MS_PIN(len_w); MS_PIN(last_w);
      float pb_w = bmpb[w], pnb_w = bmpnb[w];
      MS_PIN(pb_w); MS_PIN(pnb_w);
      const int w2 = find_in_beam(bmn, W8, child);         // -1 for child == -1 (node ids are >= 0, non-members hold -2)
      const int tis = ti >= 0 ? ti : 0;
      float pb_t = c_pb[pp * M + tis], pnb_t = c_pnb[pp * M + tis];
      MS_PIN(pb_t); MS_PIN(pnb_t);
      int flags = 0;
      if (c != p.blank && !(pc <= p.thr)) {  // `if ctc[t][c] <= prune_threshold: continue`
        const bool repeat = len_w > 0 && c == last_w;
        float a = repeat ? pc * pb_w : pc * (pb_w + pnb_w);
        if (!repeat && p.lm_factor != nullptr && c == p.sep) a = a * p.lm_factor[(size_t)n * W + w];
        if (w2 >= 0) {  // l_plus in A_prev: only Pnb[t][l_plus] += a
          par_val[w2] = a; par_rank[w2] = i * 4; par_present[w2] = 1;
        } else {
          // l_plus was a candidate of the previous frame: its Pb / Pnb (ctc_beam_decoder.py:232-241)
          const float pb_c = ti >= 0 ? pb_t : 0.f, pnb_c = ti >= 0 ? pnb_t : 0.f;
          const float bterm = pc * pnb_c;
          const float pnb_new = a + bterm;
          const float pb_new = p_blank * (pb_c + pnb_c);
          const float s = pb_new + pnb_new;
          c_pb[cp * M + slot] = pb_new; c_pnb[cp * M + slot] = pnb_new; c_s[slot] = s; c_key[slot] = i * 4 + 2;
          flags = F_PRESENT | (s > 0.f ? F_KEPT : 0);
        }
      }
      c_flags[slot] = flags;
    }
    __syncthreads();
    MS_BEAM_STAMP(1);

    // ---- S2: the beam entries themselves
    for (int w = tid; w < W; w += NT) {
      int flags = 0;
      if (w < B) {
        const bool first = w == s2_w;
        const float pb_w = first ? s2_pb : bmpb[w], pnb_w = first ? s2_pnb : bmpnb[w];
        const int last = first ? s2_last : bml[w], len_w = first ? s2_len : bmlen[w];
        const float p_last = first ? s2_plast : prow[max(last, 0)];
        int parp = par_present[w];
        float parv = par_val[w];
        int parr = par_rank[w];
        MS_PIN(parp); MS_PIN(parv); MS_PIN(parr);
        const bool in_pb = !(p_blank <= p.thr);
        const float pb_new = in_pb ? p_blank * (pb_w + pnb_w) : 0.f;
        const bool own = len_w > 0 && !(p_last <= p.thr);
        const float ownv = own ? p_last * pnb_w : 0.f;
        const bool par = parp != 0;
        const bool in_pnb = own || par;
        float pnb_new = 0.f;
        if (own && par) pnb_new = ownv + parv;
        else if (own) pnb_new = ownv;
        else if (par) pnb_new = parv;
        float s = 0.f;
        int key = 0;
        if (in_pb) {
          s = in_pnb ? pb_new + pnb_new : pb_new;
          key = (w * V + p.blank) * 4;
        } else if (in_pnb) {
          s = pnb_new;
          int r = 0x3fffffff;
          if (own) r = min(r, (w * V + last) * 4 + 1);
          if (par) r = min(r, parr);
          key = 0x40000000 | r;  // keys only in Pnb[t] follow every key of Pb[t] (Counter.__add__)
        }
        c_pb[cp * M + w] = pb_new; c_pnb[cp * M + w] = pnb_new; c_s[w] = s; c_key[w] = key;
        if (in_pb || in_pnb) flags = F_PRESENT | (s > 0.f ? F_KEPT : 0);
      }
      c_flags[w] = flags;
    }
    __syncthreads();
    MS_BEAM_STAMP(2);

    // ---- S3: sort key (+ word-count scaling) and compaction of A_next: a wave's kept candidates take consecutive places
    // behind ONE LDS atomic per wave (their order among the kept does not matter: S4 ranks by (score, key))
    const int nslots = W + B * V;
    for (int i0 = 0; i0 < nslots; i0 += NT) {
      const int i = i0 + tid;
      const bool in = i < nslots;
      int fl = in ? c_flags[i] : 0;
      float score = in ? c_s[i] : 0.f;
      int key = in ? c_key[i] : 0;
      MS_PIN(fl); MS_PIN(score); MS_PIN(key);
      if (in) c_s[i] = 0.f;                           // (S4's rank counters live here: all-zero bits)
      const bool kept = (fl & F_KEPT) != 0;
      if (kept && p.sep >= 0) {
        int nw;
        if (i < W) nw = bmnw[i];
        else {
          const int w = (i - W) / V, c = (i - W) - w * V;
          nw = bmnw[w] + ((c == p.sep && bml[w] != p.sep) ? 1 : 0);
        }
        score = score * p.word_factor[nw];
      }
      const unsigned long long km = __ballot(kept);
      int base = 0;
      if ((tid & 63) == 0 && km != 0ull) base = atomicAdd(&sh[1], (int)__popcll(km));
      base = __builtin_amdgcn_readfirstlane(base);
      if (kept) {
        const int j = base + (int)__popcll(km & ((1ull << (tid & 63)) - 1ull));
        k_idx[j] = i; k_score[j] = score; k_key[j] = key;
      }
    }
    __syncthreads();
    MS_BEAM_STAMP(3);
    const int K = sh[1];

    // ---- S4: stable descending order, keep beam_width.  Rank of candidate j = candidates that sort before it.  Every wave
    // counts over its share of the candidates (wave q: m = q, q + NT / 64, ...), the partial counts meet in LDS counters (c_s is
    // free after S3, which left zeros in it): with one wave ranking its lanes' candidates against all K through v_readlane the
    // phase was 3 400 of the frame's 12 000 cycles
    {
      int* pos_acc = reinterpret_cast<int*>(c_s);
      constexpr int NW = NT / 64;
      const int lane = tid & 63, q = tid >> 6;
      for (int j0 = 0; j0 < K; j0 += 64) {
        const int j = j0 + lane;
        const bool have = j < K;
        const float sj = have ? k_score[j] : 0.f;
        const int kj = have ? k_key[j] : 0;
        int pos = 0;
        // four (score, key) pairs per batch, every word of the batch requested before the first comparison: left to itself
        // hipcc loads a key only behind the branch on its score (two dependent LDS round trips per candidate: the phase was
        // 0.79 us for ~6 candidates per wave); the empty asm pins the loads where they are written
        for (int m0 = q; m0 < K; m0 += 4 * NW) {
          float sm[4];
          int km[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int m = min(m0 + u * NW, K - 1);
            sm[u] = k_score[m];
            km[u] = k_key[m];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(sm[u]), "+v"(km[u]));
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int before = (int)(sm[u] > sj) | ((int)(sm[u] == sj) & (int)(km[u] < kj));
            pos += (m0 + u * NW < K) ? before : 0;
          }
        }
        if (have && pos) atomicAdd(&pos_acc[j], pos);
      }
      __syncthreads();
    MS_BEAM_STAMP(4);
      for (int j = tid; j < K; j += NT) {
        const int pos = pos_acc[j];
        if (pos < W) newbeam[pos] = k_idx[j];
      }
    }
    __syncthreads();
    MS_BEAM_STAMP(5);

    // ---- S5: an extension that enters the beam becomes a trie node (if it is not one yet) with a child-table slot (if it has
    // never been in the beam before); both are recorded in its parent's row, here and in the workspace copy of that row
    const int Bn = min(K, W);
    for (int j = tid; j < Bn; j += NT) {
      const int i = newbeam[j];
      int fresh = 0;
      if (i >= W) {
        const int e = i - W, w = e / V, c = e - w * V;
        int nd = lcn[e], cs = lcc[e];
        if (nd < 0) {
          nd = atomicAdd(&sh[2], 1);
          node_parent[nd] = bmn[w]; node_sym[nd] = c; node_len[nd] = bmlen[w] + 1;
          node_nw[nd] = bmnw[w] + ((c == p.sep && bml[w] != p.sep) ? 1 : 0);
          childtab[(size_t)bmcs[w] * V + c] = nd;
          lcn[e] = nd;
        }
        if (cs < 0) {
          cs = atomicAdd(&sh[3], 1);
          node_cslot[nd] = cs;
          lcc[e] = cs;
          fresh = 1;
        }
      }
      ent_fresh[j] = fresh;
    }
    __syncthreads();
    MS_BEAM_STAMP(6);

    // ---- S6: A_prev <- best beam_width candidates; every new beam entry's child row
    for (int idx = tid; idx < Bn * V; idx += NT) {
      const int j = idx == tid ? w_first : idx / V, c = idx == tid ? c_first : idx - (idx / V) * V;
      const int i = newbeam[j];
      const bool stay = i < W;
      // (both cases' words are requested together; the addresses are valid either way)
      const int e = stay ? i * V + c : 0;
      int child_s = lcn[e], cs_s = lcc[e], fl_s = c_flags[W + e];
      int mycs = lcc[stay ? 0 : i - W];
      int fresh = ent_fresh[j];
      MS_PIN(child_s); MS_PIN(cs_s); MS_PIN(fl_s); MS_PIN(mycs); MS_PIN(fresh);
      int child, cs, ti = -1;
      if (stay) {                      // a beam entry that stays: its row moves along; a child that was a candidate of this
        child = child_s; cs = cs_s;    // frame is found at its slot in this frame's tables
        if (fl_s & F_PRESENT) ti = W + e;
      } else if (fresh) {              // never in the beam before: no children yet
        child = -1; cs = -1;
        childtab[(size_t)mycs * V + c] = -1;
      } else {                         // it was in the beam once, left and comes back: its row from the workspace
        child = childtab[(size_t)mycs * V + c];
        cs = child >= 0 ? node_cslot[child] : -1;
      }
      // a child that is itself in the (old) beam is a candidate of this frame under its own beam slot
      if (ti < 0 && child >= 0) {
        const int k = find_in_beam(bmn, W8, child);
        if (k >= 0 && (c_flags[k] & F_PRESENT)) ti = k;
      }
      lc_node[nb * WV + idx] = child; lc_cslot[nb * WV + idx] = cs; lc_tidx[nb * WV + idx] = ti;
    }
    for (int j = tid; j < Bn; j += NT) {
      const int i = newbeam[j];
      int nd, last, ln, nw, cs;
      if (i < W) {
        nd = bmn[i]; last = bml[i]; ln = bmlen[i]; nw = bmnw[i]; cs = bmcs[i];
      } else {
        const int e = i - W, w = e / V, c = e - w * V;
        nd = lcn[e]; cs = lcc[e];
        last = c; ln = bmlen[w] + 1;
        nw = bmnw[w] + ((c == p.sep && bml[w] != p.sep) ? 1 : 0);
      }
      bm_node[nb * W8 + j] = nd; bm_last[nb * W + j] = last; bm_len[nb * W + j] = ln; bm_nw[nb * W + j] = nw;
      bm_cslot[nb * W + j] = cs; bm_pb[nb * W + j] = c_pb[cp * M + i]; bm_pnb[nb * W + j] = c_pnb[cp * M + i];
    }
    for (int j = Bn + tid; j < W8; j += NT) bm_node[nb * W8 + j] = -2;
    if (tid == 0) sh[0] = Bn;
    __syncthreads();
    MS_BEAM_STAMP(7);
  }

  // ---- persist state, emit results (the beam and its rows sit in the buffers of the frame the loop stopped at)
  const int eb = t & 1;
  const int B = sh[0];
  if (tid == 0) { hdr[0] = sh[2]; hdr[1] = sh[3]; hdr[2] = B; }
  if (stamping)
    for (int k = 0; k < 9; ++k) hdr[4 + k] = (int)st_acc[k];
  for (int i = tid; i < M; i += NT) { tbl_pb[i] = c_pb[(eb ^ 1) * M + i]; tbl_pnb[i] = c_pnb[(eb ^ 1) * M + i]; }
  for (int i = tid; i < WV; i += NT) {
    g_lc_node[i] = lc_node[eb * WV + i]; g_lc_cslot[i] = lc_cslot[eb * WV + i]; g_lc_tidx[i] = lc_tidx[eb * WV + i];
  }
  for (int w = tid; w < W; w += NT)
    if (w < B) { g_beam_node[w] = bm_node[eb * W8 + w]; g_beam_pb[w] = bm_pb[eb * W + w]; g_beam_pnb[w] = bm_pnb[eb * W + w]; }
  if (p.finish && tid == 0) {
    int L = 0;
    if (B > 0) {
      int nd = bm_node[eb * W8];
      L = node_len[nd];
      for (int i = L - 1; i >= 0; --i) { p.out_idx[(size_t)n * p.T + i] = node_sym[nd]; nd = node_parent[nd]; }
    }
    p.out_len[n] = L;
  }
  if (p.beam_idx_out != nullptr) {
    if (tid == 0) p.beam_len_out[n] = B;
    for (int w = tid; w < B; w += NT) {
      int nd = bm_node[eb * W8 + w];
      const int L = node_len[nd];
      p.beam_plen_out[(size_t)n * W + w] = L;
      for (int i = L - 1; i >= 0; --i) { p.beam_idx_out[((size_t)n * W + w) * p.T + i] = node_sym[nd]; nd = node_parent[nd]; }
    }
  }
}

}  // namespace

extern "C" size_t ms_ctc_beam_workspace_bytes(int T, int N, int V, int beam_width) {
  if (T <= 0 || N <= 0 || V <= 0 || beam_width <= 0) return 0;
  return beam_layout(T, V, beam_width).per_utt * (size_t)N;
}

extern "C" int ms_ctc_beam_decode(const float* probs, const int32_t* lens, int32_t* out_idx, int32_t* out_len, int T,
                                  int N, int V, int blank, int beam_width, float prune_threshold, int separator,
                                  const float* word_factor, int t_begin, int t_end, const float* lm_factor, int finish,
                                  int32_t* beam_len, int32_t* beam_idx, int32_t* beam_plen, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  MS_REQUIRE(probs && lens && workspace, "null pointer");
  MS_REQUIRE(T > 0 && N > 0 && V > 0 && beam_width > 0, "bad shape");
  MS_REQUIRE(blank >= 0 && blank < V, "blank out of range");
  MS_REQUIRE(separator < V, "separator out of range");
  MS_REQUIRE(separator < 0 || word_factor, "word_factor required with a separator");
  MS_REQUIRE(0 <= t_begin && t_begin <= t_end && t_end <= T, "bad frame range");
  MS_REQUIRE(!finish || (out_idx && out_len), "outputs required when finishing");
  MS_REQUIRE((beam_idx == nullptr) == (beam_len == nullptr) && (beam_idx == nullptr) == (beam_plen == nullptr),
             "beam_len/beam_idx/beam_plen go together");
  const BeamLayout L = beam_layout(T, V, beam_width);
  if (workspace_bytes < L.per_utt * (size_t)N) {
    ms::set_error("ms_ctc_beam_decode: workspace too small");
    return MS_ERR_WORKSPACE;
  }
  MS_REQUIRE(beam_width <= 256, "beam_width must not exceed 256");
  const size_t lds = L.big ? 0 : beam_lds_bytes(V, beam_width);
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)beam_kernel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MS_HIP(hipFuncSetAttribute((const void*)beam_kernel<false, 256, 29, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  BeamP p;
  p.probs = probs; p.lens = lens; p.out_idx = out_idx; p.out_len = out_len; p.word_factor = word_factor;
  p.lm_factor = lm_factor; p.beam_len_out = beam_len; p.beam_idx_out = beam_idx; p.beam_plen_out = beam_plen;
  p.ws = (char*)workspace; p.L = L; p.T = T; p.N = N; p.V = V; p.W = beam_width; p.blank = blank;
  p.sep = separator < 0 ? -1 : separator; p.t_begin = t_begin; p.t_end = t_end; p.finish = finish;
  p.thr = prune_threshold;
  {
    const char* e = getenv("MS_BEAM_STAMPS");
    p.stamps = (e && e[0] == '1') ? 1 : 0;
  }
  static const int nt_env = getenv("MS_BEAM_THREADS") ? atoi(getenv("MS_BEAM_THREADS")) : 0;      // A/B switch: 64 or 256
  const bool one_wave = nt_env == 64;
  if (L.big) hipLaunchKernelGGL((beam_kernel<true, 256>), dim3(N), dim3(256), 0, (hipStream_t)stream, p);
  else if (one_wave && lds <= 64 * 1024) hipLaunchKernelGGL((beam_kernel<false, 64>), dim3(N), dim3(64), lds, (hipStream_t)stream, p);
  else if (V == 29 && beam_width == 8 && !(getenv("MS_BEAM_CONST") && getenv("MS_BEAM_CONST")[0] == '0'))
    hipLaunchKernelGGL((beam_kernel<false, 256, 29, 8>), dim3(N), dim3(256), lds, (hipStream_t)stream, p);
  // (the reference's alphabet at the other usual widths: the generic instantiation costs 4.4 us per frame at width 4 and
  // 6.3 at width 16 where 29 x 8 costs 3.5 -- tools/ctc_sweep.py)
  else if (V == 29 && beam_width == 4 && !(getenv("MS_BEAM_CONST") && getenv("MS_BEAM_CONST")[0] == '0'))
    hipLaunchKernelGGL((beam_kernel<false, 256, 29, 4>), dim3(N), dim3(256), lds, (hipStream_t)stream, p);
  else if (V == 29 && beam_width == 16 && !(getenv("MS_BEAM_CONST") && getenv("MS_BEAM_CONST")[0] == '0'))
    hipLaunchKernelGGL((beam_kernel<false, 256, 29, 16>), dim3(N), dim3(256), lds, (hipStream_t)stream, p);
  else if (V == 29 && beam_width == 32 && !(getenv("MS_BEAM_CONST") && getenv("MS_BEAM_CONST")[0] == '0'))
    hipLaunchKernelGGL((beam_kernel<false, 256, 29, 32>), dim3(N), dim3(256), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((beam_kernel<false, 256>), dim3(N), dim3(256), lds, (hipStream_t)stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
