// Does v_mfma_f32_16x16x32_f16 on gfx950 honour fp16 subnormal INPUTS (gradual underflow), or flush them to zero?
// The "f16x3" operand split (x = hi + lo in fp16) puts the lo plane of small operands (|x| < 2^-3) in the subnormal range.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f16_denorm.hip -o gpurun_out/mfma_f16_denorm && gpurun_out/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));

__global__ void k(float a_val, float b_val, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    float4_ c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}

int main() {
    float* d; hipMalloc(&d, 64);
    // (a, b): a subnormal in fp16 (< 2^-14 = 6.1e-5), b large so that the product is a normal float32
    const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 1024.f}, {ldexpf(3.f, -24), 2.f},
                              {1024.f, ldexpf(1.f, -20)}, {ldexpf(1.f, -13), 1.f}, {ldexpf(1.f, -20), ldexpf(1.f, -20)}};
    int bad = 0;
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        double want = 32.0 * (double)h[1] * (double)h[2];
        printf("a = %.6e (fp16 %.6e)  b = %.6e  ->  mfma 32-term sum %.9e   exact %.9e   %s\n", c[0], h[1], c[1], h[0], want,
               fabs(h[0] - want) <= 1e-6 * fabs(want) ? "kept" : "FLUSHED / wrong");
        bad += !(fabs(h[0] - want) <= 1e-6 * fabs(want));
    }
    printf("%s\n", bad ? "RESULT: fp16 subnormal inputs are NOT honoured" : "RESULT: fp16 subnormal inputs are honoured (gradual underflow)");
    return 0;
}
