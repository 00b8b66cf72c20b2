// Microbenchmark of the h exchange of lstm_persistent_wide2_kernel (csrc/rnn.hip) with NO arithmetic: what does a
// stream-step cost when the MFMAs and the cell are replaced by sleeps of their measured length?  Per direction 64
// workgroups of 8 waves (16 hidden units each); per stream-step (16 batch rows) a workgroup
//   pulls the whole h_{t-1} of its direction and stream: 2 planes x 32 KB, every wave its K-eighth (4 KB of each plane),
//   [emulated MFMA time, all waves], barrier,
//   [emulated cell time, waves 0-3], publishes its 16 units of h_t: 2 planes x 2 blocks x 256 B (16-byte granules, tagged).
// NS row streams are interleaved (A, B, A, B ...).  Every 16-byte granule carries the step's tag; a load repeats until it
// is fresh.  Knobs: the scope bits of the loads and of the stores, the point at which a stream-step's request is issued
// (after the cell of the step before it, as shipped, or before its barrier), the number of workgroups (one or two batch
// groups).  Reports us per stream-step and the repeated requests per wave and stream-step.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/exchange_wide.hip -o tools/micro/exchange_wide
//   exchange_wide <NS> <mfma_sleeps> <cell_sleeps> <load_aux> <store_aux> <early 0|1|2> <groups 1|2> <ring slots> <push> <mem>   (one s_sleep(8) ~ 0.21 us; aux: 1 sc0, 16 sc1, 17 both, 2 nt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct P {
  char* hbuf;    // [group][dir][stream][slot][plane][32 KB]
  unsigned long long* out;
  unsigned* fail;
  unsigned long long* spins;
  int NS, steps, mfma_sleeps, cell_sleeps, early, ring, push;   // ring: slots per stream (a power of two >= 2); push: what follows the publish
  char* dummy;
};

constexpr int PLANE = 32 * 1024;
constexpr unsigned long long LIMIT = 20000000ull;   // 0.2 s of the 100 MHz clock

template <int LAUX>
__device__ __forceinline__ u32x4 ld(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, (int)(0x80000000u | (unsigned)LAUX));
}
template <int SAUX>
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, int off, unsigned tag) {
  u32x4 v = {tag, tag, tag, tag};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, SAUX);
}
// store aux 99: the granule as two 64-bit atomic exchanges at agent scope (executed at the memory side, no return value)
__device__ __forceinline__ void st_atomic(char* base, int off, unsigned tag) {
  const unsigned long long v = ((unsigned long long)tag << 32) | tag;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(base + off);
  (void)__hip_atomic_exchange(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)__hip_atomic_exchange(q + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int LAUX, int SAUX>
__global__ __launch_bounds__(512, 2) void exchange_wide(P p) {
  const int NS = p.NS;
  const int grp = blockIdx.x / 128, d = (blockIdx.x % 128) / 64, jj = blockIdx.x % 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int RING = p.ring;
  const size_t dir_bytes = (size_t)NS * RING * 2 * PLANE;
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(p.hbuf + ((size_t)grp * 2 + d) * dir_bytes, 0, (int)dir_bytes, 0x00020000);
  auto slot_off = [&](int s, int par) { return (s * RING + par) * 2 * PLANE; };   // par: slot index, step & (RING - 1)
  // publish: 64 lanes of wave 0, plane = lane >> 5, block = (lane >> 4) & 1, row granule = lane & 15
  char* hbase = p.hbuf + ((size_t)grp * 2 + d) * dir_bytes;
  auto publish = [&](int s, int par, unsigned tag) {
    const int off = slot_off(s, par) + (tid >> 5) * PLANE + (2 * jj + ((tid >> 4) & 1)) * 256 + (tid & 15) * 16;
    if (tid < 64) {
      if (SAUX == 99) st_atomic(hbase, off, tag);
      else st<SAUX>(rh, off, tag);
    }
  };
  for (int s = 0; s < NS; ++s) publish(s, 0, 1u);
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  unsigned long long spins = 0;
  u32x4 v[8], vn[8];
  auto request = [&](int s, int par) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      v[l] = ld<LAUX>(rh, slot_off(s, par) + wave * 4096 + l * 1024 + lane * 16);
      v[4 + l] = ld<LAUX>(rh, slot_off(s, par) + PLANE + wave * 4096 + l * 1024 + lane * 16);
    }
  };
  auto request_next = [&](int s, int par) {   // early == 2: a second register set, asked for a whole stream-step ahead
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      vn[l] = ld<LAUX>(rh, slot_off(s, par) + wave * 4096 + l * 1024 + lane * 16);
      vn[4 + l] = ld<LAUX>(rh, slot_off(s, par) + PLANE + wave * 4096 + l * 1024 + lane * 16);
    }
  };
  if (p.early) request(0, 0);
  for (int t = 0; t < p.steps && ok; ++t) {
    const unsigned tag = (unsigned)(t + 1);
    const int par = t & (RING - 1);
    for (int s = 0; s < NS && ok; ++s) {
      const unsigned long long tw = wall_clock64();
      if (!p.early) request(s, par);
      for (;;) {
        unsigned bad = 0;
#pragma unroll
        for (int l = 0; l < 8; ++l) bad |= (v[l][0] ^ tag) | (v[l][3] ^ tag);
        if (!__any(bad != 0)) break;
        ++spins;
        if (wall_clock64() - tw > LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
        request(s, par);
      }
      if (p.early == 2) {   // the tags of this stream-step passed: ask for the next one now, into the other register set
        const int sn = (s + 1) % NS, tn = s + 1 < NS ? t : t + 1;
        request_next(sn, tn & (RING - 1));
      }
      for (int z = 0; z < p.mfma_sleeps; ++z) __builtin_amdgcn_s_sleep(8);
      if (p.early == 1) {   // the next stream-step's request goes out before the barrier and the cell of this one
        const int sn = (s + 1) % NS, tn = s + 1 < NS ? t : t + 1;
        request(sn, tn & (RING - 1));
      }
      __syncthreads();
      if (wave < 4) {
        for (int z = 0; z < p.cell_sleeps; ++z) __builtin_amdgcn_s_sleep(8);
        publish(s, (t + 1) & (RING - 1), tag + 1);
        if (p.push == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // buffer_wbl2 sc1 + s_waitcnt vmcnt(0)
        if (p.push == 2) {                                                              // two plain stores elsewhere (the kernel's output stores)
          float* o = reinterpret_cast<float*>(p.dummy) + ((size_t)blockIdx.x * 512 + tid) * 2;
          o[0] = (float)t; o[1] = (float)s;
        }
        if (p.push == 3) __builtin_amdgcn_s_waitcnt(0x0f70);                            // s_waitcnt vmcnt(0) alone
      }
      if (p.early == 2) {
#pragma unroll
        for (int l = 0; l < 8; ++l) v[l] = vn[l];
      }
    }
  }
  if (!ok && lane == 0) atomicAdd(p.fail, 1u);
  if (lane == 0) atomicAdd(p.spins, spins);
  if (tid == 0 && blockIdx.x == 0) p.out[0] = wall_clock64() - t0;
}

template <int LAUX>
static void launch(int saux, int wgs, const P& p) {
  if (saux == 16) hipLaunchKernelGGL((exchange_wide<LAUX, 16>), dim3(wgs), dim3(512), 0, 0, p);
  else if (saux == 17) hipLaunchKernelGGL((exchange_wide<LAUX, 17>), dim3(wgs), dim3(512), 0, 0, p);
  else if (saux == 1) hipLaunchKernelGGL((exchange_wide<LAUX, 1>), dim3(wgs), dim3(512), 0, 0, p);
  else if (saux == 99) hipLaunchKernelGGL((exchange_wide<LAUX, 99>), dim3(wgs), dim3(512), 0, 0, p);
  else hipLaunchKernelGGL((exchange_wide<LAUX, 18>), dim3(wgs), dim3(512), 0, 0, p);
}

int main(int argc, char** argv) {
  P p;
  p.NS = argc > 1 ? atoi(argv[1]) : 2;
  p.mfma_sleeps = argc > 2 ? atoi(argv[2]) : 0;
  p.cell_sleeps = argc > 3 ? atoi(argv[3]) : 0;
  const int laux = argc > 4 ? atoi(argv[4]) : 16, saux = argc > 5 ? atoi(argv[5]) : 16;
  p.early = argc > 6 ? atoi(argv[6]) : 0;
  const int groups = argc > 7 ? atoi(argv[7]) : 1;
  p.ring = argc > 8 ? atoi(argv[8]) : 2;
  p.push = argc > 9 ? atoi(argv[9]) : 0;
  (void)hipMalloc(&p.dummy, (size_t)256 * 512 * 8);
  if (p.ring < 2 || (p.ring & (p.ring - 1)) || p.ring > 256) { printf("ring: a power of two in 2..256\n"); return 1; }
  p.steps = 1000;
  if (p.NS < 1 || p.NS > 8 || groups < 1 || groups > 2) { printf("NS in 1..8, groups in 1..2\n"); return 1; }
  const size_t hbytes = (size_t)groups * 2 * p.NS * p.ring * 2 * PLANE;
  // <mem>: 0 = hipMalloc (coarse-grained, what the library's callers hand over), 1 = fine-grained device memory, 2 = uncached
  const int mem = argc > 10 ? atoi(argv[10]) : 0;
  if (mem == 1) { if (hipExtMallocWithFlags((void**)&p.hbuf, hbytes, hipDeviceMallocFinegrained) != hipSuccess) { printf("fine-grained allocation failed\n"); return 1; } }
  else if (mem == 2) { if (hipExtMallocWithFlags((void**)&p.hbuf, hbytes, hipDeviceMallocUncached) != hipSuccess) { printf("uncached allocation failed\n"); return 1; } }
  else (void)hipMalloc(&p.hbuf, hbytes);
  (void)hipMalloc(&p.out, 8); (void)hipMalloc(&p.fail, 4); (void)hipMalloc(&p.spins, 8);
  (void)hipMemset(p.hbuf, 0, hbytes); (void)hipMemset(p.out, 0, 8); (void)hipMemset(p.fail, 0, 4); (void)hipMemset(p.spins, 0, 8);
  const int wgs = groups * 128;
  if (laux == 16) launch<16>(saux, wgs, p);
  else if (laux == 17) launch<17>(saux, wgs, p);
  else if (laux == 1) launch<1>(saux, wgs, p);
  else launch<18>(saux, wgs, p);
  (void)hipDeviceSynchronize();
  unsigned long long t, sp; unsigned f;
  (void)hipMemcpy(&t, p.out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, p.fail, 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&sp, p.spins, 8, hipMemcpyDeviceToHost);
  const double us = t * 10.0 / p.steps / 1000.0;
  printf("wide exchange: ring %d, %d streams, %d workgroups, mfma=%d cell=%d sleeps, load aux %d, store aux %d, request %s: %.3f us per stream-step, "
         "%.2f repeated requests per wave and stream-step%s\n", p.ring, p.NS, wgs, p.mfma_sleeps, p.cell_sleeps, laux, saux,
         p.early == 2 ? "a stream-step ahead (second register set)" : p.early ? "before the barrier" : "after the cell", us / p.NS, (double)sp / ((double)wgs * 8 * p.steps * p.NS), f ? "  (TIMEOUTS!)" : "");
  if (mem) printf("   (exchange buffer in %s device memory)\n", mem == 1 ? "fine-grained" : "uncached");
  if (p.push) printf("   (after the publish: %s)\n", p.push == 1 ? "release fence at agent scope" : p.push == 2 ? "two plain stores elsewhere" : "s_waitcnt vmcnt(0)");
  return 0;
}
