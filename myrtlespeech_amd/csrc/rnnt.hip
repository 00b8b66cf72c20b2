// RNN-T (transducer) decode-step kernels.  NOT reference-derived: the reference snapshot has
// no transducer; the specification is this repository's own
// (myrtlespeech_amd/model/rnnt.py, Graves 2012) and parity is against oracle/rnnt_oracle.py only.
//
//   ms_embedding_forward   : out[r,:] = table[idx[r],:]                       (gather)
//   ms_rnnt_joint_forward  : logp[r,:] = log_softmax(W_o . tanh(enc_p[enc_row[r]] + pred_p[r]) + b_o)
//                            one workgroup per hypothesis row, tanh vector staged in LDS,
//                            each wave owns output symbols v = wave, wave+4, ... (lanes split J)
//   ms_rnnt_topk           : per utterance, the k best of C candidate scores, descending,
//                            ties -> lowest candidate index (wave shuffles, no sort)
#include <math.h>

#include "common.h"

namespace {

__global__ void embedding_kernel(const float* __restrict__ table, const int32_t* __restrict__ idx,
                                 float* __restrict__ out, int D, int V1) {
  const int r = blockIdx.x;
  int i = idx[r];
  i = min(max(i, 0), V1 - 1);
  for (int k = threadIdx.x; k < D; k += blockDim.x) out[(size_t)r * D + k] = table[(size_t)i * D + k];
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void rnnt_joint_kernel(const float* __restrict__ enc_p, const int32_t* __restrict__ enc_row,
                                                         const float* __restrict__ pred_p, const float* __restrict__ w_out,
                                                         const float* __restrict__ b_out, float* __restrict__ logp, int J,
                                                         int V1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* z = smem;        // [J]
  float* lg = smem + J;   // [V1]
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* e = enc_p + (size_t)enc_row[r] * J;
  const float* p = pred_p + (size_t)r * J;
  for (int k = tid; k < J; k += 256) z[k] = tanhf(e[k] + p[k]);
  __syncthreads();
  for (int v = wave; v < V1; v += 4) {
    const float* w = w_out + (size_t)v * J;
    float acc = 0.f;
    for (int k = lane; k < J; k += 64) acc += w[k] * z[k];
    acc = wave_sum_f(acc);
    if (lane == 0) lg[v] = acc + (b_out ? b_out[v] : 0.f);
  }
  __syncthreads();
  if (wave == 0) {
    float m = -INFINITY;
    for (int v = lane; v < V1; v += 64) m = fmaxf(m, lg[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int v = lane; v < V1; v += 64) s += expf(lg[v] - m);
    s = wave_sum_f(s);
    const float lz = logf(s) + m;
    for (int v = lane; v < V1; v += 64) logp[(size_t)r * V1 + v] = lg[v] - lz;
  }
}

// k rounds of (max value, lowest index) selection over one row of C scores.
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ scores, int32_t* __restrict__ out_idx,
                                                   float* __restrict__ out_val, int C, int k) {
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ int chosen;
  extern __shared__ __attribute__((aligned(16))) float row[];  // [C] working copy
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 256) row[c] = scores[(size_t)b * C + c];
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = tid; c < C; c += 256) {
      const float v = row[c];
      if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float fv = wv[0];
      int fi = wi[0];
      for (int w = 1; w < 4; ++w)
        if (wv[w] > fv || (wv[w] == fv && wi[w] < fi)) { fv = wv[w]; fi = wi[w]; }
      const bool none = (fi == 0x7fffffff);
      out_idx[(size_t)b * k + j] = none ? -1 : fi;
      out_val[(size_t)b * k + j] = none ? -INFINITY : fv;
      chosen = none ? -1 : fi;
    }
    __syncthreads();
    if (tid == 0 && chosen >= 0) row[chosen] = -INFINITY;  // -inf entries are never selected again (idx -1)
    __syncthreads();
  }
}

}  // namespace

extern "C" int ms_embedding_forward(const float* table, const int32_t* idx, float* out, int R, int D, int V1,
                                    void* stream) {
  MS_REQUIRE(table && idx && out, "null pointer");
  MS_REQUIRE(R > 0 && D > 0 && V1 > 0, "bad shape");
  hipLaunchKernelGGL(embedding_kernel, dim3(R), dim3(128), 0, (hipStream_t)stream, table, idx, out, D, V1);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_rnnt_joint_forward(const float* enc_p, const int32_t* enc_row, const float* pred_p, const float* w_out,
                                     const float* b_out, float* logp, int R, int J, int V1, void* stream) {
  MS_REQUIRE(enc_p && enc_row && pred_p && w_out && logp, "null pointer");
  MS_REQUIRE(R > 0 && J > 0 && V1 > 0, "bad shape");
  const size_t lds = (size_t)(J + V1) * sizeof(float);
  MS_REQUIRE(lds <= 64 * 1024, "joint width too large");
  hipLaunchKernelGGL(rnnt_joint_kernel, dim3(R), dim3(256), lds, (hipStream_t)stream, enc_p, enc_row, pred_p, w_out, b_out,
                     logp, J, V1);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_rnnt_topk(const float* scores, int32_t* out_idx, float* out_val, int B, int C, int k, void* stream) {
  MS_REQUIRE(scores && out_idx && out_val, "null pointer");
  MS_REQUIRE(B > 0 && C > 0 && k > 0, "bad shape");
  MS_REQUIRE((size_t)C * sizeof(float) <= 60 * 1024, "too many candidates per row");
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(256), (size_t)C * sizeof(float), (hipStream_t)stream, scores, out_idx,
                     out_val, C, k);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
