// Microbenchmark: round-trip latency of a 16-byte tagged hand-off between two workgroups on the
// same XCD vs different XCDs, for store/load cache-policy variants.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/handoff_latency.hip -o /tmp/handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ST, int LD>  // ST: 0 plain, 16 sc1; LD aux: 16 sc1, 17 sc0 sc1
__global__ __launch_bounds__(64) void pingpong(unsigned* buf, int a, int b, int iters, unsigned long long* out, int* xcc) {
  extern __shared__ char pad[];  // force 1 WG per CU
  const int me = blockIdx.x;
  if (me != a && me != b) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[me == a ? 0 : 1] = (int)(id & 0xf);
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
  const int mine = (me == a) ? 0 : 1024, theirs = (me == a) ? 1024 : 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (me == a) {
      u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
      if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, ST);
    }
    // wait for the partner's tag i (bounded: 5 ms per hop, then give up and report)
    const unsigned long long tw = wall_clock64();
    bool ok = true;
    for (;;) {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, (int)(0x80000000u | LD));
      if (__builtin_amdgcn_readfirstlane(v[0]) == (unsigned)i && __builtin_amdgcn_readfirstlane(v[3]) == (unsigned)i) break;
      if (wall_clock64() - tw > 500000ull) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) {
      if (me == a && threadIdx.x == 0) out[0] = ~0ull;
      return;
    }
    if (me == b) {
      u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
      if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, ST);
    }
  }
  if (me == a && threadIdx.x == 0) out[0] = wall_clock64() - t0;
}

template <int ST, int LD>
void run(const char* name, unsigned* buf, unsigned long long* out, int* xcc, int a, int b) {
  const int iters = 2000;
  (void)hipMemset(buf, 0, 4096);
  (void)hipMemset(out, 0, 8);
  hipLaunchKernelGGL((pingpong<ST, LD>), dim3(256), dim3(64), 100 * 1024, 0, buf, a, b, iters, out, xcc);
  hipDeviceSynchronize();
  unsigned long long t; int x[2];
  hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
  hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
  if (t == ~0ull) printf("%-28s WG %3d (xcc %d) <-> WG %3d (xcc %d): NEVER VISIBLE (gave up)\n", name, a, x[0], b, x[1]);
  else printf("%-28s WG %3d (xcc %d) <-> WG %3d (xcc %d): %.3f us per one-way hop\n", name, a, x[0], b, x[1],
              t * 10.0 / iters / 2 / 1000.0);
  fflush(stdout);
}

int main() {
  unsigned* buf; unsigned long long* out; int* xcc;
  hipMalloc(&buf, 4096); hipMalloc(&out, 8); hipMalloc(&xcc, 8);
  hipFuncSetAttribute((const void*)pingpong<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)pingpong<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)pingpong<16, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)pingpong<0, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int b : {8, 16, 1, 3}) {
    run<16, 16>("store sc1 / load sc1", buf, out, xcc, 0, b);
    run<16, 17>("store sc1 / load sc0sc1", buf, out, xcc, 0, b);
    run<0, 16>("store plain / load sc1", buf, out, xcc, 0, b);
    run<0, 17>("store plain / load sc0sc1", buf, out, xcc, 0, b);
  }
  return 0;
}
