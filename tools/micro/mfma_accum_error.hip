// How accurately does v_mfma_f32_*_{f16,bf16} ACCUMULATE?  The products of 16-bit operands are exact in float32; what is
// left is the summation inside the instruction (32 / 16 products per lane-group and call) and into the accumulator.
// One wave, one 16x16 (32x32) output tile, K = 64 .. 4096, operands = random values exactly representable in the plane
// format; reference in float64 from the same operands.  Reported: max and rms of |got - exact| / sum_k |a_k b_k| over the
// tile, and the mean signed error (a bias shows truncation rather than round-to-nearest), next to a float32 fmaf chain in
// k order on the same operands.  FLUSH = n: the accumulator is added into a float32 master (v_add_f32, RN) and cleared
// every n MFMAs.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_accum_error.hip -o tools/micro/mfma_accum_error
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A [16][K], B [16][K] as floats (exactly representable in the format); out [16][16]: D[i][j] = sum_k A[i][k] B[j][k]
template <bool HALF>
__global__ void k16(const float* A, const float* B, float* out, int K, int flush) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f4 acc = {0, 0, 0, 0}, master = {0, 0, 0, 0};
  int n = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    if (HALF) {
      half8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[r * K + k0 + 8 * q + e]; b[e] = (_Float16)B[r * K + k0 + 8 * q + e]; }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    } else {
      bf8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[r * K + k0 + 8 * q + e]; b[e] = (__bf16)B[r * K + k0 + 8 * q + e]; }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    if (flush > 0 && ++n == flush) { master += acc; acc = f4{0, 0, 0, 0}; n = 0; }
  }
  master += acc;
  // D layout of 16x16: lane (r = column j of B... ) -- row i = 4 q + e, column j = r   [A rows = i ... verified below by the host]
  for (int e = 0; e < 4; ++e) out[(4 * q + e) * 16 + r] = master[e];
}

int main() {
  const int Ks[] = {64, 256, 1024, 2048, 4096};
  float *dA, *dB, *dO;
  hipMalloc(&dA, 16 * 4096 * 4); hipMalloc(&dB, 16 * 4096 * 4); hipMalloc(&dO, 1024);
  for (int half = 0; half < 2; ++half)
    for (int K : Ks)
      for (int flush : {0, 8, 2, 1}) {
        std::vector<float> A(16 * K), B(16 * K);
        srand(1234 + K);
        auto rnd = [&]() {
          float v = (float)rand() / RAND_MAX * 2.f - 1.f;
          if (half) return (float)(_Float16)v;
          return (float)(__bf16)v;
        };
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = fabsf(rnd());   // positive B: sums of mixed sign through A
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        if (half) hipLaunchKernelGGL(k16<true>, dim3(1), dim3(64), 0, 0, dA, dB, dO, K, flush);
        else hipLaunchKernelGGL(k16<false>, dim3(1), dim3(64), 0, 0, dA, dB, dO, K, flush);
        float O[256];
        hipMemcpy(O, dO, 1024, hipMemcpyDeviceToHost);
        // the tile may be D[i][j] or its transpose: take the assignment with the smaller error
        double best_max = 1e30, best_rms = 0, best_bias = 0, f32_max = 0;
        for (int tr = 0; tr < 2; ++tr) {
          double mx = 0, sq = 0, bias = 0, fm = 0;
          for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
              double ex = 0, ab = 0; float chain = 0.f;
              for (int k = 0; k < K; ++k) {
                ex += (double)A[i * K + k] * B[j * K + k]; ab += fabs((double)A[i * K + k] * B[j * K + k]);
                chain = fmaf(A[i * K + k], B[j * K + k], chain);
              }
              const double got = tr ? O[j * 16 + i] : O[i * 16 + j];
              const double e = (got - ex) / ab;
              mx = fmax(mx, fabs(e)); sq += e * e; bias += e;
              fm = fmax(fm, fabs((chain - ex) / ab));
            }
          if (mx < best_max) { best_max = mx; best_rms = sqrt(sq / 256); best_bias = bias / 256; f32_max = fm; }
        }
        printf("%-5s K %4d flush %d: max %.3e  rms %.3e  mean signed %+.3e   (f32 fmaf chain: max %.3e)   [units of sum|ab|; 2^-24 = 5.96e-8]\n",
               half ? "f16" : "bf16", K, flush, best_max, best_rms, best_bias, f32_max);
      }
  return 0;
}
