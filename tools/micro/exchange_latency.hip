// Microbenchmark of the LSTM hand-off pattern alone: G groups of P producer workgroups; every
// step each workgroup publishes SLICE bytes (tagged 16-byte granules, sc1 stores) and then each of
// its 4 waves loads a quarter of the group's whole vector (P*SLICE bytes) until every granule
// shows the step's tag.  No arithmetic.  Reports us per step.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/exchange_latency.hip -o tools/micro/exchange_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct P {
  unsigned* buf;      // [G][2 parity][P*SLICE bytes]
  unsigned long long* out;
  unsigned* fail;
  int G, PW, slice_granules, steps, loads_per_wave, work_sleeps;
};

template <int LPW>
__global__ __launch_bounds__(256, 1) void exchange(P p) {
  extern __shared__ char pad[];
  const int g = blockIdx.x / p.PW, j = blockIdx.x % p.PW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vec_granules = p.PW * p.slice_granules;  // 16-byte granules in one group's vector
  char* base = (char*)p.buf + (size_t)g * 2 * vec_granules * 16;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 2 * vec_granules * 16, 0x00020000);
  // publish step 0 (tag 1) into parity 0
  if (tid < p.slice_granules) {
    u32x4 v = {1u, 1u, 1u, 1u};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (j * p.slice_granules + tid) * 16, 0, 16);
  }
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < p.steps; ++s) {
    const unsigned tag = (unsigned)(s + 1);
    const int par = s & 1;
    const int q_granules = vec_granules / 4;  // per wave
    const unsigned long long tw = wall_clock64();
    bool ok = true;
    // the wave walks its quarter in chunks of LPW x 64 granules (all LPW loads in flight together),
    // re-polling a chunk until fresh
    for (int c0 = 0; c0 < q_granules; c0 += LPW * 64) {
      for (;;) {
        unsigned bad = 0;
        u32x4 v[LPW];
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
          const int gi = wave * q_granules + c0 + l * 64 + lane;
          v[l] = __builtin_amdgcn_raw_buffer_load_b128(r, (par * vec_granules + gi) * 16, 0, (int)(0x80000000u | 17u));
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) bad |= (v[l][0] ^ tag) | (v[l][3] ^ tag);
        if (!__any(bad != 0)) break;
        if (wall_clock64() - tw > 2000000ull) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) break;
    }
    if (!ok) { if (lane == 0) atomicAdd(p.fail, 1u); return; }
    __syncthreads();
    for (int z = 0; z < p.work_sleeps; ++z) __builtin_amdgcn_s_sleep(8);  // emulated per-step work (~0.21 us each)
    if (tid < p.slice_granules) {
      const unsigned nt = tag + 1;
      u32x4 v = {nt, nt, nt, nt};
      __builtin_amdgcn_raw_buffer_store_b128(v, r, ((par ^ 1) * vec_granules + j * p.slice_granules + tid) * 16, 0, 16);
    }
  }
  if (tid == 0 && blockIdx.x == 0) p.out[0] = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 2;
  const int PW = argc > 2 ? atoi(argv[2]) : 128;
  const int slice = argc > 3 ? atoi(argv[3]) : 64;       // granules per producer per step (64 = 1 KB)
  const int lpw = argc > 4 ? atoi(argv[4]) : 16;         // loads per wave per poll
  const int steps = 1000;
  P p;
  const size_t bytes = (size_t)G * 2 * PW * slice * 16;
  (void)hipMalloc(&p.buf, bytes); (void)hipMalloc(&p.out, 8); (void)hipMalloc(&p.fail, 4);
  (void)hipMemset(p.buf, 0, bytes); (void)hipMemset(p.out, 0, 8); (void)hipMemset(p.fail, 0, 4);
  p.G = G; p.PW = PW; p.slice_granules = slice; p.steps = steps; p.loads_per_wave = lpw;
  if ((PW * slice / 4) % (lpw * 64) != 0) { printf("bad chunking\n"); return 1; }
  p.work_sleeps = argc > 5 ? atoi(argv[5]) : 0;
#define LAUNCH(L) { (void)hipFuncSetAttribute((const void*)exchange<L>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
                    hipLaunchKernelGGL(exchange<L>, dim3(G * PW), dim3(256), 100 * 1024, 0, p); }
  switch (lpw) { case 1: LAUNCH(1) break; case 2: LAUNCH(2) break; case 4: LAUNCH(4) break; case 8: LAUNCH(8) break;
                 case 16: LAUNCH(16) break; default: printf("lpw must be 1,2,4,8,16\n"); return 1; }
  (void)hipDeviceSynchronize();
  unsigned long long t; unsigned f;
  (void)hipMemcpy(&t, p.out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, p.fail, 4, hipMemcpyDeviceToHost);
  printf("G=%d P=%d slice=%d B vector=%d KB loads/wave/poll=%d work=%d : %.3f us per step%s\n", G, PW, slice * 16,
         PW * slice * 16 / 1024, lpw, p.work_sleeps, t * 10.0 / steps / 1000.0, f ? "  (TIMEOUTS!)" : "");
  return 0;
}
