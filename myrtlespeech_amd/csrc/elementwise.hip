// HBM-bound helpers: in-place time mask, (N,CF,T)->(T,N,CF) layout change, clamp.
#include "common.h"

namespace {

// x [N, inner, T]: zero t >= lens[n].  One workgroup per (utterance, 64 rows): it only touches the padded tail, so a
// batch of full-length utterances costs 0.6 k empty workgroups instead of one thread per element (21 -> 3 us at the
// config-2 conv2 input).
constexpr int MASK_ROWS = 64;
__global__ void mask_time_kernel(float* __restrict__ x, const int32_t* __restrict__ lens, int inner, int T) {
  const int n = blockIdx.y;
  const int len = max(lens[n], 0);
  if (len >= T) return;
  const int row0 = blockIdx.x * MASK_ROWS, row1 = min(row0 + MASK_ROWS, inner);
  for (int row = row0; row < row1; ++row) {
    float* xr = x + ((size_t)n * inner + row) * T;
    for (int t = len + threadIdx.x; t < T; t += blockDim.x) xr[t] = 0.0f;
  }
}

// 32x32 LDS tile transpose per batch element: in[n][cf][t] -> out[t][n][cf].
__global__ void nct_to_tnc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int CF, int T) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < CF && t < T) ? x[((size_t)n * CF + c) * T + t] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < CF) y[((size_t)t * N + n) * CF + c] = tile[tx][i];
  }
}

__global__ void clamp_kernel(const float* x, float* y, size_t n, float lo, float hi) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = fminf(fmaxf(x[i], lo), hi);
}

}  // namespace

extern "C" int ms_mask_time_(float* x, const int32_t* lens, int N, int inner, int T, void* stream) {
  ms::ProfScope prof_span(MS_PROF_OTHER, (hipStream_t)stream);
  MS_REQUIRE(x && lens, "null pointer");
  MS_REQUIRE(N > 0 && inner > 0 && T > 0, "bad shape");
  MS_REQUIRE(N <= 65535, "N exceeds grid limits");
  dim3 grid(ms::cdiv(inner, MASK_ROWS), N);
  hipLaunchKernelGGL(mask_time_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, lens, inner, T);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_nct_to_tnc(const float* x, float* y, int N, int CF, int T, void* stream) {
  ms::ProfScope prof_span(MS_PROF_LAYOUT, (hipStream_t)stream);
  MS_REQUIRE(x && y, "null pointer");
  MS_REQUIRE(N > 0 && CF > 0 && T > 0, "bad shape");
  MS_REQUIRE(N <= 65535 && ms::cdiv(CF, 32) <= 65535, "N/CF exceed grid limits");
  dim3 grid(ms::cdiv(T, 32), ms::cdiv(CF, 32), N);
  hipLaunchKernelGGL(nct_to_tnc_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, x, y, N, CF, T);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_clamp(const float* x, float* y, size_t n, float lo, float hi, void* stream) {
  ms::ProfScope prof_span(MS_PROF_OTHER, (hipStream_t)stream);
  MS_REQUIRE((x && y) || n == 0, "null pointer");
  if (n == 0) return MS_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, lo, hi);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
