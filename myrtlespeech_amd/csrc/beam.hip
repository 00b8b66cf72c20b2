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
// Prefix identity is a trie in the workspace (node = parent + symbol) with a child table
// for every node that has ever been in the beam, so "l + c" resolves to the same node no
// matter when it is re-derived (needed for `l_plus in A_prev` and the Pb[t-1][l_plus]
// look-ups of ctc_beam_decoder.py:232-241).
//
// One workgroup (256 threads) per utterance, frames sequential.  Compiled with
// -ffp-contract=off: the reference rounds after every multiply and every add.
#include <algorithm>

#include "common.h"

namespace {

constexpr int HDR_INTS = 16;
enum { F_PRESENT = 1, F_KEPT = 2 };

struct BeamLayout {
  size_t per_utt;  // bytes
  size_t hdr, beam_node, beam_pb, beam_pnb, tbl_pb, tbl_pnb, node_parent, node_sym, node_len, node_nw, node_cslot,
      node_tidx, node_tstamp, childtab;
  int NN, CS, M;
};

BeamLayout beam_layout(int T, int V, int W) {
  BeamLayout L;
  L.M = W * (V + 1);
  L.NN = 2 + T * W * V;
  L.CS = 2 + T * W + W;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += ms::align_up(bytes, 64); return r; };
  L.hdr = take(HDR_INTS * 4);
  L.beam_node = take((size_t)W * 4);
  L.beam_pb = take((size_t)W * 4);
  L.beam_pnb = take((size_t)W * 4);
  L.tbl_pb = take((size_t)2 * L.M * 4);
  L.tbl_pnb = take((size_t)2 * L.M * 4);
  L.node_parent = take((size_t)L.NN * 4);
  L.node_sym = take((size_t)L.NN * 4);
  L.node_len = take((size_t)L.NN * 4);
  L.node_nw = take((size_t)L.NN * 4);
  L.node_cslot = take((size_t)L.NN * 4);
  L.node_tidx = take((size_t)2 * L.NN * 8);     // {frame stamp, table index} pairs, [2][NN] (one 8-byte load per look-up)
  L.node_tstamp = L.node_tidx;
  L.childtab = take((size_t)L.CS * V * 4);
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
  int T, N, V, W, blank, sep, t_begin, t_end, finish;
  float thr;
};

__global__ __launch_bounds__(256) void beam_kernel(BeamP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, n = blockIdx.x;
  const int V = p.V, W = p.W, M = p.L.M;
  // ---- LDS carve
  float* prow = reinterpret_cast<float*>(smem_raw);          // [V]
  float* c_pb = prow + V;                                     // [M]
  float* c_pnb = c_pb + M;
  float* c_s = c_pnb + M;
  float* c_score = c_s + M;
  int* c_key = reinterpret_cast<int*>(c_score + M);           // [M]
  int* c_flags = c_key + M;
  int* c_child = c_flags + M;
  int* k_idx = c_child + M;
  int* bm_node = k_idx + M;                                   // [W] each
  int* bm_last = bm_node + W;
  int* bm_len = bm_last + W;
  int* bm_nw = bm_len + W;
  int* bm_cslot = bm_nw + W;
  float* bm_pb = reinterpret_cast<float*>(bm_cslot + W);
  float* bm_pnb = bm_pb + W;
  int* par_present = reinterpret_cast<int*>(bm_pnb + W);
  float* par_val = reinterpret_cast<float*>(par_present + W);
  int* par_rank = reinterpret_cast<int*>(par_val + W);
  int* newbeam = par_rank + W;
  int* sh = newbeam + W;  // [0]=B, [1]=k_count, [2]=n_nodes, [3]=n_cslots
  // this frame's and the previous frame's candidate tables Pb / Pnb (ctc_beam_decoder.py:182-192) live in LDS (round 4: they
  // were a dependent global load at the end of every look-up chain); they go back to the workspace when the call ends
  float* l_tbl_pb = reinterpret_cast<float*>(sh + 8);       // [2][M]
  float* l_tbl_pnb = l_tbl_pb + 2 * M;                      // [2][M]
  // the kept candidates' (score, order key) pairs, compacted: the ranking loop of S4 reads them in sequence
  float* k_score = l_tbl_pnb + 2 * M;                       // [M]
  int* k_key = reinterpret_cast<int*>(k_score + M);         // [M]
  int* c_cslot = k_key + M;                                 // [M] child-table slot of an extension's node (-1: none yet)

  // ---- global state of this utterance
  char* u = p.ws + (size_t)n * p.L.per_utt;
  int* hdr = reinterpret_cast<int*>(u + p.L.hdr);
  int* g_beam_node = reinterpret_cast<int*>(u + p.L.beam_node);
  float* g_beam_pb = reinterpret_cast<float*>(u + p.L.beam_pb);
  float* g_beam_pnb = reinterpret_cast<float*>(u + p.L.beam_pnb);
  float* tbl_pb = reinterpret_cast<float*>(u + p.L.tbl_pb);
  float* tbl_pnb = reinterpret_cast<float*>(u + p.L.tbl_pnb);
  int* node_parent = reinterpret_cast<int*>(u + p.L.node_parent);
  int* node_sym = reinterpret_cast<int*>(u + p.L.node_sym);
  int* node_len = reinterpret_cast<int*>(u + p.L.node_len);
  int* node_nw = reinterpret_cast<int*>(u + p.L.node_nw);
  int* node_cslot = reinterpret_cast<int*>(u + p.L.node_cslot);
  int2* node_tt = reinterpret_cast<int2*>(u + p.L.node_tidx);   // [2][NN] {stamp, index}
  int* childtab = reinterpret_cast<int*>(u + p.L.childtab);
  const int NN = p.L.NN;

  if (p.t_begin == 0) {
    // Pb[-1][()] = 1, Pnb[-1][()] = 0, A_prev = [()]   (ctc_beam_decoder.py:182-192)
    if (tid == 0) {
      node_parent[0] = -1; node_sym[0] = -1; node_len[0] = 0; node_nw[0] = 0; node_cslot[0] = 0;
      node_tt[0] = make_int2(-1, 0); node_tt[NN] = make_int2(-1, 0);
      bm_node[0] = 0; bm_pb[0] = 1.0f; bm_pnb[0] = 0.0f;
      sh[0] = 1; sh[2] = 1; sh[3] = 1;
    }
    for (int v = tid; v < V; v += 256) childtab[v] = -1;
  } else {
    if (tid == 0) { sh[0] = hdr[2]; sh[2] = hdr[0]; sh[3] = hdr[1]; }
    for (int w = tid; w < W; w += 256) { bm_node[w] = g_beam_node[w]; bm_pb[w] = g_beam_pb[w]; bm_pnb[w] = g_beam_pnb[w]; }
    for (int i = tid; i < 2 * M; i += 256) { l_tbl_pb[i] = tbl_pb[i]; l_tbl_pnb[i] = tbl_pnb[i]; }
  }
  __syncthreads();
  // the beam entries' last symbol / length / word count / child-table slot ride along in LDS from frame to frame (S6 knows
  // them when it builds the next beam); only a call's first frame fetches them from the trie
  for (int w = tid; w < W; w += 256)
    if (w < sh[0]) {
      const int nd = bm_node[w];
      bm_last[w] = node_sym[nd]; bm_len[w] = node_len[nd]; bm_nw[w] = node_nw[nd]; bm_cslot[w] = node_cslot[nd];
    }
  __syncthreads();

  const int len = min(max(p.lens[n], 0), p.T);
  const int t_stop = min(p.t_end, len);
  // the frame's probabilities do not depend on the search: thread v holds p[t + 1][v] a frame ahead (alphabets beyond 256
  // symbols fetch the rest in the frame itself)
  float p_next = 0.f;
  if (p.t_begin < t_stop && tid < V) p_next = p.probs[((size_t)p.t_begin * p.N + n) * V + tid];
  for (int t = p.t_begin; t < t_stop; ++t) {
    const int B = sh[0];
    if (B == 0) break;  // an empty beam stays empty (ctc_beam_decoder.py:258)
    const int cp = t & 1, pp = cp ^ 1;
    const float* row = p.probs + ((size_t)t * p.N + n) * V;
    if (tid < V) prow[tid] = p_next;
    for (int v = tid + 256; v < V; v += 256) prow[v] = row[v];
    if (t + 1 < t_stop && tid < V) p_next = p.probs[((size_t)(t + 1) * p.N + n) * V + tid];
    for (int w = tid; w < W; w += 256) par_present[w] = 0;
    if (tid == 0) sh[1] = 0;
    __syncthreads();
    const float p_blank = prow[p.blank];

    // ---- S1: extensions l + c
    for (int i = tid; i < B * V; i += 256) {
      const int w = i / V, c = i - w * V;
      const int slot = W + i;
      int flags = 0;
      if (c != p.blank) {
        const float pc = prow[c];
        if (!(pc <= p.thr)) {  // `if ctc[t][c] <= prune_threshold: continue`
          const int child = childtab[(size_t)bm_cslot[w] * V + c];
          int w2 = -1;
          if (child >= 0)
            for (int k = 0; k < B; ++k)
              if (bm_node[k] == child) w2 = k;
          const bool repeat = bm_len[w] > 0 && c == bm_last[w];
          float a = repeat ? pc * bm_pb[w] : pc * (bm_pb[w] + bm_pnb[w]);
          if (!repeat && p.lm_factor != nullptr && c == p.sep) a = a * p.lm_factor[(size_t)n * W + w];
          if (w2 >= 0) {  // l_plus in A_prev: only Pnb[t][l_plus] += a
            par_val[w2] = a; par_rank[w2] = i * 4; par_present[w2] = 1;
          } else {
            float pb_c = 0.f, pnb_c = 0.f;
            int cslot_c = -1;
            if (child >= 0) {
              const int2 tt = node_tt[pp * NN + child];
              cslot_c = node_cslot[child];          // beside the stamp, not behind it: S6 needs it if this candidate survives
              if (tt.x == t) { pb_c = l_tbl_pb[pp * M + tt.y]; pnb_c = l_tbl_pnb[pp * M + tt.y]; }
            }
            c_cslot[slot] = cslot_c;
            const float bterm = pc * pnb_c;
            const float pnb_new = a + bterm;
            const float pb_new = p_blank * (pb_c + pnb_c);
            const float s = pb_new + pnb_new;
            c_pb[slot] = pb_new; c_pnb[slot] = pnb_new; c_s[slot] = s; c_key[slot] = i * 4 + 2; c_child[slot] = child;
            flags = F_PRESENT | (s > 0.f ? F_KEPT : 0);
          }
        }
      }
      c_flags[slot] = flags;
    }
    __syncthreads();

    // ---- S2: the beam entries themselves
    for (int w = tid; w < W; w += 256) {
      int flags = 0;
      if (w < B) {
        const bool in_pb = !(p_blank <= p.thr);
        const float pb_new = in_pb ? p_blank * (bm_pb[w] + bm_pnb[w]) : 0.f;
        const int last = bm_last[w];
        const bool own = bm_len[w] > 0 && !(prow[last] <= p.thr);
        const float ownv = own ? prow[last] * bm_pnb[w] : 0.f;
        const bool par = par_present[w] != 0;
        const bool in_pnb = own || par;
        float pnb_new = 0.f;
        if (own && par) pnb_new = ownv + par_val[w];
        else if (own) pnb_new = ownv;
        else if (par) pnb_new = par_val[w];
        float s = 0.f;
        int key = 0;
        if (in_pb) {
          s = in_pnb ? pb_new + pnb_new : pb_new;
          key = (w * V + p.blank) * 4;
        } else if (in_pnb) {
          s = pnb_new;
          int r = 0x3fffffff;
          if (own) r = min(r, (w * V + last) * 4 + 1);
          if (par) r = min(r, par_rank[w]);
          key = 0x40000000 | r;  // keys only in Pnb[t] follow every key of Pb[t] (Counter.__add__)
        }
        c_pb[w] = pb_new; c_pnb[w] = pnb_new; c_s[w] = s; c_key[w] = key; c_child[w] = bm_node[w];
        if (in_pb || in_pnb) flags = F_PRESENT | (s > 0.f ? F_KEPT : 0);
      }
      c_flags[w] = flags;
    }
    __syncthreads();

    // ---- S3: sort key (+ word-count scaling) and compaction of A_next
    const int nslots = W + B * V;
    for (int i = tid; i < nslots; i += 256) {
      if (c_flags[i] & F_KEPT) {
        float score = c_s[i];
        if (p.sep >= 0) {
          int nw;
          if (i < W) nw = bm_nw[i];
          else {
            const int w = (i - W) / V, c = (i - W) - w * V;
            nw = bm_nw[w] + ((c == p.sep && bm_last[w] != p.sep) ? 1 : 0);
          }
          score = score * p.word_factor[nw];
        }
        c_score[i] = score;
        const int j = atomicAdd(&sh[1], 1);
        k_idx[j] = i; k_score[j] = score; k_key[j] = c_key[i];
      }
    }
    __syncthreads();
    const int K = sh[1];

    // ---- S4: stable descending order, keep beam_width.  Rank of candidate j = candidates that sort before it; the pairs are
    // read in sequence (round 4: the loop went through k_idx[m] -> c_score / c_key, two dependent LDS round trips per
    // comparison, and was 3.6 of the frame's 7 us)
    if (K <= 64) {
      // the usual case (22 kept candidates on average at V = 29, W = 8): one wave, candidate j in lane j, the others'
      // pairs broadcast from registers (v_readlane) -- no LDS traffic in the loop
      if (tid < 64) {
        const bool have = tid < K;
        const float sj = have ? k_score[tid] : 0.f;
        const int kj = have ? k_key[tid] : 0;
        int pos = 0;
        for (int m = 0; m < K; ++m) {
          const float sm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sj), m));   // m is wave-uniform: v_readlane
          const int km = __builtin_amdgcn_readlane(kj, m);
          pos += (sm > sj || (sm == sj && km < kj)) ? 1 : 0;
        }
        if (have && pos < W) newbeam[pos] = k_idx[tid];
      }
    } else {
      for (int j = tid; j < K; j += 256) {
        const float sj = k_score[j];
        const int kj = k_key[j];
        int pos = 0;
#pragma unroll 8
        for (int m = 0; m < K; ++m) {
          const float sm = k_score[m];
          pos += (sm > sj || (sm == sj && k_key[m] < kj)) ? 1 : 0;
        }
        if (pos < W) newbeam[pos] = k_idx[j];
      }
    }

    // ---- S5: trie nodes + this frame's tables for every present candidate
    for (int i = tid; i < nslots; i += 256) {
      if (!(c_flags[i] & F_PRESENT)) continue;
      int nd = c_child[i];
      if (nd < 0) {
        const int w = (i - W) / V, c = (i - W) - w * V;
        nd = atomicAdd(&sh[2], 1);
        node_parent[nd] = bm_node[w]; node_sym[nd] = c; node_len[nd] = bm_len[w] + 1;
        node_nw[nd] = bm_nw[w] + ((c == p.sep && bm_last[w] != p.sep) ? 1 : 0);
        node_cslot[nd] = -1; node_tt[nd] = make_int2(-1, 0); node_tt[NN + nd] = make_int2(-1, 0);
        childtab[(size_t)bm_cslot[w] * V + c] = nd;
        c_child[i] = nd;
      }
      l_tbl_pb[cp * M + i] = c_pb[i]; l_tbl_pnb[cp * M + i] = c_pnb[i];
      node_tt[cp * NN + nd] = make_int2(t + 1, i);
    }
    __syncthreads();

    // ---- S6: A_prev <- best beam_width candidates
    const int Bn = min(K, W);
    int nb_node = -1, nb_last = -1, nb_len = 0, nb_nw = 0, nb_cslot = 0;
    float nb_pb = 0.f, nb_pnb = 0.f;
    if (tid < Bn) {
      const int i = newbeam[tid];
      nb_node = c_child[i]; nb_pb = c_pb[i]; nb_pnb = c_pnb[i];
      if (i < W) {                     // a beam entry that stays: it has had its child-table slot since it entered the beam
        nb_last = bm_last[i]; nb_len = bm_len[i]; nb_nw = bm_nw[i]; nb_cslot = bm_cslot[i];
      } else {                         // an extension l + c (its node exists since S5 at the latest)
        const int w = (i - W) / V, c = (i - W) - w * V;
        nb_last = c; nb_len = bm_len[w] + 1;
        nb_nw = bm_nw[w] + ((c == p.sep && bm_last[w] != p.sep) ? 1 : 0);
        nb_cslot = c_cslot[i];               // -1: the node was made this frame or has never been in the beam
        if (nb_cslot < 0) {
          nb_cslot = atomicAdd(&sh[3], 1);
          node_cslot[nb_node] = nb_cslot;
          for (int v = 0; v < V; ++v) childtab[(size_t)nb_cslot * V + v] = -1;
        }
      }
    }
    __syncthreads();
    if (tid < Bn) {
      bm_node[tid] = nb_node; bm_pb[tid] = nb_pb; bm_pnb[tid] = nb_pnb;
      bm_last[tid] = nb_last; bm_len[tid] = nb_len; bm_nw[tid] = nb_nw; bm_cslot[tid] = nb_cslot;
    }
    if (tid == 0) sh[0] = Bn;
    __syncthreads();
  }

  // ---- persist state, emit results
  const int B = sh[0];
  if (tid == 0) { hdr[0] = sh[2]; hdr[1] = sh[3]; hdr[2] = B; }
  for (int i = tid; i < 2 * M; i += 256) { tbl_pb[i] = l_tbl_pb[i]; tbl_pnb[i] = l_tbl_pnb[i]; }
  for (int w = tid; w < W; w += 256)
    if (w < B) { g_beam_node[w] = bm_node[w]; g_beam_pb[w] = bm_pb[w]; g_beam_pnb[w] = bm_pnb[w]; }
  if (p.finish && tid == 0) {
    int L = 0;
    if (B > 0) {
      int nd = bm_node[0];
      L = node_len[nd];
      for (int i = L - 1; i >= 0; --i) { p.out_idx[(size_t)n * p.T + i] = node_sym[nd]; nd = node_parent[nd]; }
    }
    p.out_len[n] = L;
  }
  if (p.beam_idx_out != nullptr) {
    if (tid == 0) p.beam_len_out[n] = B;
    if (tid < B) {
      int nd = bm_node[tid];
      const int L = node_len[nd];
      p.beam_plen_out[(size_t)n * W + tid] = L;
      for (int i = L - 1; i >= 0; --i) { p.beam_idx_out[((size_t)n * W + tid) * p.T + i] = node_sym[nd]; nd = node_parent[nd]; }
    }
  }
}

size_t beam_lds_bytes(int V, int W) {
  const size_t M = (size_t)W * (V + 1);
  return (size_t)V * 4 + M * 8 * 4 + (size_t)W * 11 * 4 + 8 * 4 + 4 * M * 4 + 3 * M * 4;
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
  const size_t lds = beam_lds_bytes(V, beam_width);
  if (lds > 150 * 1024 || beam_width > 256) {
    ms::set_error("ms_ctc_beam_decode: beam_width * (alphabet + 1) too large for the LDS candidate tables");
    return MS_ERR_UNSUPPORTED;
  }
  static ms::DeviceOnce attr_once;
  if (attr_once.need()) {
    MS_HIP(hipFuncSetAttribute((const void*)beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  BeamP p;
  p.probs = probs; p.lens = lens; p.out_idx = out_idx; p.out_len = out_len; p.word_factor = word_factor;
  p.lm_factor = lm_factor; p.beam_len_out = beam_len; p.beam_idx_out = beam_idx; p.beam_plen_out = beam_plen;
  p.ws = (char*)workspace; p.L = L; p.T = T; p.N = N; p.V = V; p.W = beam_width; p.blank = blank;
  p.sep = separator < 0 ? -1 : separator; p.t_begin = t_begin; p.t_end = t_end; p.finish = finish;
  p.thr = prune_threshold;
  hipLaunchKernelGGL(beam_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, p);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
