// Microbenchmark of the exchange pattern of a 2-D decomposition of the LSTM recurrence (VERDICT r2 item 3 / DESIGN 7.2b), no
// arithmetic: what would a time step cost if a workgroup owned 8 S hidden units x a 1/S slice of K instead of 8 units x all
// of K?  Per direction 128 workgroups = (128 / S) unit groups x S K-peers.  Per stream-step (16 batch rows) a workgroup
//   hop 1: pulls its K-slice of h_{t-1}: 16 rows x (1024 / S) x 4 B (64 KB / S), published 512 B at a time by the 128
//          workgroups of the direction (every workgroup finalises 8 units);
//          [emulated MFMA time]
//   hop 2: publishes its partial gate sums for its S - 1 peers (16 rows x 32 gate columns x 4 B = 2 KB each) and pulls the
//          S - 1 blocks meant for it;
//          [emulated cell time], publishes its 8 units of h_t (512 B).
// NS independent row streams are interleaved hop by hop (A(0) .. A(NS-1), B(0) .. B(NS-1)), so that one stream's hop
// latency hides behind the others' work.  Every 16-byte granule carries the step's tag; loads are sc0 sc1 and repeat until
// fresh, stores are sc1 -- the protocol of csrc/rnn.hip.  Reports us per step of all NS streams.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/exchange_2d.hip -o tools/micro/exchange_2d
//   exchange_2d <S> <NS> <mfma_sleeps> <cell_sleeps>      (one s_sleep(8) ~ 0.21 us)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct P {
  char* hbuf;    // [dir][stream][parity][64 KB]            unit-major: unit u -> 64 B (16 rows x 4 B)
  char* pbuf;    // [dir][stream][parity][unit group][dst peer][src peer][2 KB]
  unsigned long long* out;
  unsigned* fail;
  int S, NS, steps, mfma_sleeps, cell_sleeps;
};

constexpr int HVEC = 64 * 1024, PBLK = 2048;
constexpr unsigned long long LIMIT = 20000000ull;   // 0.2 s of the 100 MHz clock

__device__ __forceinline__ u32x4 ld(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, (int)(0x80000000u | 17u));
}
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, int off, unsigned tag) {
  u32x4 v = {tag, tag, tag, tag};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}

__global__ __launch_bounds__(256, 1) void exchange2d(P p) {
  const int S = p.S, NS = p.NS, UG = 128 / S;
  const int d = blockIdx.x / 128, w = blockIdx.x % 128, ug = w / S, ks = w % S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t hdir = (size_t)NS * 2 * HVEC, pdir = (size_t)NS * 2 * UG * S * S * PBLK;
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(p.hbuf + d * hdir, 0, (int)hdir, 0x00020000);
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(p.pbuf + d * pdir, 0, (int)pdir, 0x00020000);
  const int my_units = (ug * S + ks) * 8 * 64;            // byte offset of this workgroup's 8 units inside a 64 KB vector
  const int slice = HVEC / S;                              // bytes of h this workgroup pulls per stream-step
  auto hoff = [&](int s, int par) { return (s * 2 + par) * HVEC; };
  auto poff = [&](int s, int par, int dst, int src) { return (((s * 2 + par) * UG + ug) * S * S + dst * S + src) * PBLK; };
  for (int s = 0; s < NS; ++s)                             // h_0 (tag 1) into parity 0
    if (tid < 32) st(rh, hoff(s, 0) + my_units + tid * 16, 1u);
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  for (int t = 0; t < p.steps && ok; ++t) {
    const unsigned tag = (unsigned)(t + 1);
    const int par = t & 1;
    // ---- A(s): pull the K-slice of h, "MFMA", publish partial sums for the peers
    for (int s = 0; s < NS && ok; ++s) {
      const int per_wave = slice / 4;                     // bytes per wave; 1 KB per load instruction
      const unsigned long long tw = wall_clock64();
      for (;;) {
        unsigned bad = 0;
        for (int l = 0; l < per_wave / 1024; ++l) {
          const u32x4 v = ld(rh, hoff(s, par) + ks * slice + wave * per_wave + l * 1024 + lane * 16);
          bad |= (v[0] ^ tag) | (v[3] ^ tag);
        }
        if (!__any(bad != 0)) break;
        if (wall_clock64() - tw > LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      for (int z = 0; z < p.mfma_sleeps; ++z) __builtin_amdgcn_s_sleep(8);
      for (int g = tid; g < (S - 1) * (PBLK / 16); g += 256) {     // 128 granules per peer
        const int peer = g / (PBLK / 16), dst = peer + (peer >= ks ? 1 : 0);
        st(rp, poff(s, par, dst, ks) + (g % (PBLK / 16)) * 16, tag);
      }
    }
    // ---- B(s): pull the peers' partial sums, "cell", publish 8 units of h_t
    for (int s = 0; s < NS && ok; ++s) {
      const unsigned long long tw = wall_clock64();
      for (;;) {
        unsigned bad = 0;
        for (int l = wave; l < (S - 1) * 2; l += 4) {              // 2 loads of 1 KB per peer block
          const int peer = l / 2, src = peer + (peer >= ks ? 1 : 0);
          const u32x4 v = ld(rp, poff(s, par, ks, src) + (l & 1) * 1024 + lane * 16);
          bad |= (v[0] ^ tag) | (v[3] ^ tag);
        }
        if (!__any(bad != 0)) break;
        if (wall_clock64() - tw > LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      for (int z = 0; z < p.cell_sleeps; ++z) __builtin_amdgcn_s_sleep(8);
      if (tid < 32) st(rh, hoff(s, par ^ 1) + my_units + tid * 16, tag + 1);
    }
  }
  if (!ok && lane == 0) atomicAdd(p.fail, 1u);
  if (tid == 0 && blockIdx.x == 0) p.out[0] = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  P p;
  p.S = argc > 1 ? atoi(argv[1]) : 4;
  p.NS = argc > 2 ? atoi(argv[2]) : 4;
  p.mfma_sleeps = argc > 3 ? atoi(argv[3]) : 0;
  p.cell_sleeps = argc > 4 ? atoi(argv[4]) : 0;
  p.steps = 1000;
  if (!(p.S == 2 || p.S == 4 || p.S == 8) || p.NS < 1 || p.NS > 8) { printf("S in {2,4,8}, NS in 1..8\n"); return 1; }
  const size_t hbytes = (size_t)2 * p.NS * 2 * HVEC, pbytes = (size_t)2 * p.NS * 2 * (128 / p.S) * p.S * p.S * PBLK;
  (void)hipMalloc(&p.hbuf, hbytes); (void)hipMalloc(&p.pbuf, pbytes); (void)hipMalloc(&p.out, 8); (void)hipMalloc(&p.fail, 4);
  (void)hipMemset(p.hbuf, 0, hbytes); (void)hipMemset(p.pbuf, 0, pbytes); (void)hipMemset(p.out, 0, 8); (void)hipMemset(p.fail, 0, 4);
  hipLaunchKernelGGL(exchange2d, dim3(256), dim3(256), 0, 0, p);
  (void)hipDeviceSynchronize();
  unsigned long long t; unsigned f;
  (void)hipMemcpy(&t, p.out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&f, p.fail, 4, hipMemcpyDeviceToHost);
  const double us = t * 10.0 / p.steps / 1000.0;
  printf("2-D exchange S=%d streams=%d (16 rows each) mfma=%d cell=%d: pulled per workgroup and stream-step %d KB h + %d KB partial sums; "
         "%.3f us per step of all streams = %.3f us per 32 rows%s\n", p.S, p.NS, p.mfma_sleeps, p.cell_sleeps, 64 / p.S, 2 * (p.S - 1), us,
         us * 2.0 / p.NS, f ? "  (TIMEOUTS!)" : "");
  return 0;
}
