// Error string + device query shared by every entry point.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <algorithm>
#include "common.h"

namespace ms {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int num_cus() {
  static std::atomic<int> cus[64];  // per device ordinal; 0 = not queried yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int v = cus[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    v = p.multiProcessorCount;
    cus[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}
int precision_mode() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_PRECISION");
    v = PREC_F16X3;
    if (e && strcmp(e, "f32") == 0) v = PREC_F32;
    if (e && strcmp(e, "fp16") == 0) v = PREC_F16;
    if (e && strcmp(e, "bf16x3") == 0) v = PREC_BF16X3;
  }
  return v;
}

namespace {
__global__ void absmax_bits_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(w[i]) & 0x7FFFFFFFu;
    if (b < 0x7F800000u) m = max(m, b);          // finite values only (|x| as bits is monotone)
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void scale_word_kernel(float* __restrict__ word, int half_planes) {
  const unsigned bits = __float_as_uint(word[0]);
  int s = 0;
  if (half_planes && bits != 0) {
    const int e = (int)((bits >> 23) & 255) - 127;      // floor(log2(max |w|)) (a subnormal maximum counts as 2^-127)
    s = min(max(12 - e, -100), 100);
  }
  word[0] = ldexpf(1.0f, s);
  word[1] = ldexpf(1.0f, -s);
}
}  // namespace

int weight_scale_launch(const float* w, size_t n, float* scale_word, int prec, hipStream_t stream) {
  MS_HIP(hipMemsetAsync(scale_word, 0, 2 * sizeof(float), stream));
  const bool half_planes = prec == PREC_F16X3 || prec == PREC_F16;
  if (half_planes && n > 0) {
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(blocks), dim3(256), 0, stream, w, n, (unsigned*)scale_word);
  }
  hipLaunchKernelGGL(scale_word_kernel, dim3(1), dim3(1), 0, stream, scale_word, half_planes ? 1 : 0);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

namespace {
struct ProfSpan { hipEvent_t a, b; int kind; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfSpan> g_spans;      // recorded, not yet read
std::vector<ProfSpan> g_free;       // event pairs ready for re-use
}  // namespace

ProfScope::ProfScope(int kind_, hipStream_t st) : kind(kind_), stream(st), on(g_prof_on.load(std::memory_order_relaxed)) {
  if (!on) return;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_free.empty()) { a = g_free.back().a; b = g_free.back().b; g_free.pop_back(); }
  }
  if (!a && (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)) { on = false; return; }
  (void)hipEventRecord(a, stream);
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(b, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_spans.push_back({a, b, kind});
}
}  // namespace ms

extern "C" int ms_prof_enable(int on) {
  ms::g_prof_on.store(on != 0, std::memory_order_relaxed);
  return MS_OK;
}

extern "C" int ms_prof_read(float* out_ms, int* out_n) {
  MS_REQUIRE(out_ms && out_n, "null pointer");
  for (int k = 0; k < MS_PROF_KINDS; ++k) { out_ms[k] = 0.f; out_n[k] = 0; }
  std::lock_guard<std::mutex> lk(ms::g_prof_mu);
  for (auto& s : ms::g_spans) {
    MS_HIP(hipEventSynchronize(s.b));
    float t = 0.f;
    MS_HIP(hipEventElapsedTime(&t, s.a, s.b));
    if (s.kind >= 0 && s.kind < MS_PROF_KINDS) { out_ms[s.kind] += t; out_n[s.kind] += 1; }
    ms::g_free.push_back(s);
  }
  ms::g_spans.clear();
  return MS_OK;
}

namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int samples, unsigned long long spacing_ticks) {
  if (threadIdx.x != 0) return;
  unsigned long long next = wall_clock64();
  for (int i = 0; i < samples; ++i) {
    unsigned long long now;
    while ((now = wall_clock64()) < next) __builtin_amdgcn_s_sleep(8);     // bounded: `next` is at most 50 ms away in total
    out[2 * i] = now;
    out[2 * i + 1] = clock64();
    next = now + spacing_ticks;
  }
}
}  // namespace

extern "C" int ms_clock_probe(unsigned long long* out_dev, int samples, int spacing_us, void* stream) {
  MS_REQUIRE(out_dev && samples > 0 && spacing_us > 0, "bad arguments");
  MS_REQUIRE((long)samples * spacing_us <= 50000, "at most 50 ms of sampling");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev, samples,
                     (unsigned long long)spacing_us * 100ull);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

namespace {
// `steps` dependent rounds of {LDS write, s_barrier, LDS read of the neighbour's word} in ONE workgroup: the cost of one
// barrier-separated step of the scan kernels (CTC alpha / beta rows, the beam search's phases), which is what their time
// is made of -- bench.py prices their serial chains with it (floor = barriers x this).
__global__ __launch_bounds__(1024) void barrier_chain_kernel(unsigned long long* out, int steps) {
  __shared__ float cell[1024];
  const int tid = threadIdx.x, nb = (tid + 1) % blockDim.x;
  float v = (float)tid;
  cell[tid] = v;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    cell[tid] = v;
    __syncthreads();
    v = cell[nb] * 0.5f + 1.0f;
    __syncthreads();          // the scan kernels' double buffers save this second barrier; counted as two steps below
  }
  const unsigned long long t1 = wall_clock64();
  if (tid == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)v; }
}
}  // namespace

extern "C" int ms_barrier_chain_probe(unsigned long long* out_dev, int steps, int threads, void* stream) {
  MS_REQUIRE(out_dev && steps > 0 && steps <= (1 << 22), "bad arguments");
  MS_REQUIRE(threads >= 64 && threads <= 1024 && threads % 64 == 0, "threads must be a multiple of 64 in [64, 1024]");
  hipLaunchKernelGGL(barrier_chain_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, out_dev, steps);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_abi_version(void) { return MS_ABI_VERSION; }
extern "C" const char* ms_last_error(void) {
  static thread_local std::string copy;
  copy = ms::g_err;
  return copy.c_str();
}
