// Feature front-end on device: MFCC (torchaudio 0.4.0 pipeline restated: centred reflect-padded
// STFT -> power -> HTK mel filterbank -> dB with top_db clamp -> DCT-II), Standardize, AddContextFrames and the
// SpecAugment band zeroing, all batched over ragged utterances.  The three contractions of the MFCC (DFT, mel, DCT)
// are exact-f32 MFMA GEMMs (ms::linear_launch); everything else is HBM-bound elementwise / reduction work.
#include <limits.h>

#include "common.h"

namespace ms {
int linear_launch(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, float lo,
                  float hi, hipStream_t stream);
}

namespace {

__device__ __forceinline__ int frames_of(int samples, int hop) { return 1 + samples / hop; }

// Monotonic float <-> int key so a float maximum can be an integer atomicMax.
__device__ __forceinline__ int float_key(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// frames[(n*T + t), j] = window[j] * wave[n, reflect(t*hop + j - n_fft/2)]; rows t >= frames(n) are zero.
// One workgroup per frame: consecutive lanes read consecutive samples (coalesced apart from the mirrored edges).
__global__ void stft_frames_kernel(const float* __restrict__ wave, const int32_t* __restrict__ wave_lens,
                                   const float* __restrict__ window, float* __restrict__ frames, int max_samples, int T,
                                   int n_fft, int hop) {
  const int n = blockIdx.y, t = blockIdx.x;
  const int len = wave_lens[n];
  float* dst = frames + ((size_t)n * T + t) * n_fft;
  const bool live = len > 0 && t < frames_of(len, hop);
  const float* src = wave + (size_t)n * max_samples;
  const int s0 = t * hop - n_fft / 2;
  for (int j = threadIdx.x; j < n_fft; j += blockDim.x) {
    float v = 0.0f;
    if (live) {
      int s = s0 + j;
      if (s < 0) s = -s;
      if (s >= len) s = 2 * (len - 1) - s;
      s = min(max(s, 0), len - 1);   // only reachable when len <= n_fft/2, which the host rejects
      v = window[j] * src[s];
    }
    dst[j] = v;
  }
}

// spec [M, 2*nf] (real bins then imaginary bins) -> pw [M, nf] = re^2 + im^2.
__global__ void power_kernel(const float* __restrict__ spec, float* __restrict__ pw, size_t M, int nf) {
  const size_t total = M * (size_t)nf;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / nf;
    const int k = (int)(i - m * nf);
    const float re = spec[m * 2 * nf + k], im = spec[m * 2 * nf + nf + k];
    pw[i] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
  }
}

// mel [N*T, n_mels] -> dB in place; per-utterance maximum over the valid frames into maxkey[n].
__global__ void to_db_kernel(float* __restrict__ mel, const int32_t* __restrict__ wave_lens, int* __restrict__ maxkey,
                             int T, int n_mels, int hop, float multiplier, float amin, float db_offset) {
  __shared__ int red[4];
  const int n = blockIdx.y;
  const int valid = frames_of(wave_lens[n], hop) * n_mels;   // valid prefix of this utterance's [T, n_mels] block
  float* base = mel + (size_t)n * T * n_mels;
  int best = INT_MIN;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T * n_mels; i += gridDim.x * blockDim.x) {
    const float v = multiplier * log10f(fmaxf(base[i], amin)) - db_offset;
    base[i] = v;
    if (i < valid) best = max(best, float_key(v));
  }
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = max(best, red[w]);
    if (best != INT_MIN) atomicMax(maxkey + n, best);
  }
}

__global__ void db_floor_kernel(float* __restrict__ mel, const int* __restrict__ maxkey, int T, int n_mels,
                                float top_db) {
  const int n = blockIdx.y;
  const float floor_db = key_float(maxkey[n]) - top_db;
  float* base = mel + (size_t)n * T * n_mels;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T * n_mels; i += gridDim.x * blockDim.x)
    base[i] = fmaxf(base[i], floor_db);
}

// cep [N*T, C] -> out [N, C, T], zero for t >= frames(n).  32x32 LDS tile transpose.
__global__ void cep_to_nct_kernel(const float* __restrict__ cep, const int32_t* __restrict__ wave_lens,
                                  float* __restrict__ out, int C, int T, int hop) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int nfr = frames_of(wave_lens[n], hop);
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (t < nfr && t < T && c < C) ? cep[((size_t)n * T + t) * C + c] : 0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (c < C && t < T) out[((size_t)n * C + c) * T + t] = tile[tx][i];
  }
}

// ---- Standardize ------------------------------------------------------------------------------------------------

// pass 0: stats[n][0] += sum x;  pass 1: stats[n][1] += sum (x - mean)^2   over the valid region t < len.
template <int PASS>
__global__ void moments_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens, double* __restrict__ stats,
                               int inner, int T) {
  __shared__ double red[4];
  const int n = blockIdx.y;
  const int len = lens ? min(lens[n], T) : T;
  const float* base = x + (size_t)n * inner * T;
  const size_t total = (size_t)inner * T;
  double mean = 0.0;
  if (PASS == 1) mean = stats[2 * n] / ((double)inner * len);
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    if (t < len) {
      const double v = (double)base[i];
      acc += PASS == 0 ? v : (v - mean) * (v - mean);
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) acc += red[w];
    atomicAdd(stats + 2 * n + PASS, acc);
  }
}

__global__ void standardize_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                   const double* __restrict__ stats, float* __restrict__ y, int inner, int T) {
  const int n = blockIdx.y;
  const int len = lens ? min(lens[n], T) : T;
  const double count = (double)inner * len;
  const float mean = (float)(stats[2 * n] / count);
  const float sd = (float)sqrt(stats[2 * n + 1] / (count - 1.0));   // unbiased, like torch.Tensor.std()
  const size_t total = (size_t)inner * T;
  const float* src = x + (size_t)n * total;
  float* dst = y + (size_t)n * total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    dst[i] = t < len ? (src[i] - mean) / sd : 0.0f;
  }
}

// ---- AddContextFrames -------------------------------------------------------------------------------------------

// x [N, F, T] -> y [N, 2c+1, F, T]: y[n,w,f,t] = x[n,f,t+w-c] inside the utterance, 0 outside.
__global__ void context_frames_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                      float* __restrict__ y, int F, int T, int n_context) {
  const int n = blockIdx.z;
  const int len = lens ? min(lens[n], T) : T;
  const int wf = blockIdx.y;             // w * F + f
  const int w = wf / F, f = wf - w * F;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int s = t + w - n_context;
  const int W = 2 * n_context + 1;
  float v = 0.0f;
  if (t < len && s >= 0 && s < len) v = x[((size_t)n * F + f) * T + s];
  y[(((size_t)n * W + w) * F + f) * T + t] = v;
}

// ---- SpecAugment ------------------------------------------------------------------------------------------------

// x [N, C, F, T]: zero feature rows [start, start+width) of f_bands[n][i] and frames of t_bands[n][i].
__global__ void zero_bands_kernel(float* __restrict__ x, const int32_t* __restrict__ f_bands,
                                  const int32_t* __restrict__ t_bands, int C, int F, int T, int n_f, int n_t) {
  const int n = blockIdx.z;
  const int cf = blockIdx.y;
  const int f = cf % F;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  bool hit = false;
  for (int i = 0; i < n_f; ++i) {
    const int s = f_bands[((size_t)n * n_f + i) * 2], wd = f_bands[((size_t)n * n_f + i) * 2 + 1];
    hit |= (f >= s && f < s + wd);
  }
  for (int i = 0; i < n_t; ++i) {
    const int s = t_bands[((size_t)n * n_t + i) * 2], wd = t_bands[((size_t)n * n_t + i) * 2 + 1];
    hit |= (t >= s && t < s + wd);
  }
  if (hit) x[((size_t)n * C * F + cf) * T + t] = 0.0f;
}

// ---- MFCCLegacy (python_speech_features 0.6 pipeline, float64) ------------------------------------------------------

__host__ __device__ inline int legacy_frames(int samples, int frame_len, int frame_step) {
  return samples <= frame_len ? 1 : 1 + (samples - frame_len + frame_step - 1) / frame_step;
}

// One workgroup per frame.  wave -> int16 range -> pre-emphasis -> rectangular frame -> |DFT_nfft|^2 / nfft ->
// filterbank energies -> log -> DCT -> lifter, c0 := log(frame energy).  All in float64; out float32 [N, numcep, T].
__global__ void __launch_bounds__(256) mfcc_legacy_kernel(
    const float* __restrict__ wave, const int32_t* __restrict__ wave_lens, const double* __restrict__ twiddle,
    const double* __restrict__ fbank, const double* __restrict__ dct, const double* __restrict__ lifter,
    float* __restrict__ out, int max_samples, int T, int frame_len, int frame_step, int nfft, int nfilt, int numcep,
    double preemph) {
  extern __shared__ double lds[];
  const int nbins = nfft / 2 + 1;
  double* frame = lds;                    // [nfft]
  double* tw = frame + nfft;              // [nfft][2]
  double* pspec = tw + 2 * nfft;          // [nbins]
  double* feat = pspec + nbins;           // [nfilt]
  double* red = feat + nfilt;             // [4]
  const int n = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int slen = wave_lens[n];
  if (t >= legacy_frames(slen, frame_len, frame_step)) {
    for (int c = tid; c < numcep; c += blockDim.x) out[((size_t)n * numcep + c) * T + t] = 0.0f;
    return;
  }
  const float* src = wave + (size_t)n * max_samples;
  const int used = min(frame_len, nfft);  // numpy.fft.rfft(frames, nfft) truncates longer frames
  for (int j = tid; j < nfft; j += blockDim.x) {
    double v = 0.0;
    const int s = t * frame_step + j;
    if (j < used && s < slen) {
      const double cur = (double)(int16_t)(int)(src[s] * 32768.0f);
      v = s == 0 ? cur : cur - preemph * (double)(int16_t)(int)(src[s - 1] * 32768.0f);
    }
    frame[j] = v;
    tw[2 * j] = twiddle[2 * j];
    tw[2 * j + 1] = twiddle[2 * j + 1];
  }
  __syncthreads();
  double part = 0.0;
  for (int k = tid; k < nbins; k += blockDim.x) {
    double re = 0.0, im = 0.0;
    int idx = 0;                          // (k * j) mod nfft
    for (int j = 0; j < used; ++j) {
      re += frame[j] * tw[2 * idx];
      im -= frame[j] * tw[2 * idx + 1];
      idx += k;
      if (idx >= nfft) idx -= nfft;
    }
    const double p = (re * re + im * im) / (double)nfft;
    pspec[k] = p;
    part += p;
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  double energy = red[0] + red[1] + red[2] + red[3];
  if (energy == 0.0) energy = 2.220446049250313e-16;
  for (int m = tid; m < nfilt; m += blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < nbins; ++k) acc += pspec[k] * fbank[(size_t)m * nbins + k];
    if (acc == 0.0) acc = 2.220446049250313e-16;
    feat[m] = log(acc);
  }
  __syncthreads();
  for (int c = tid; c < numcep; c += blockDim.x) {
    double acc = 0.0;
    for (int m = 0; m < nfilt; ++m) acc += feat[m] * dct[(size_t)c * nfilt + m];
    acc *= lifter[c];
    if (c == 0) acc = log(energy);
    out[((size_t)n * numcep + c) * T + t] = (float)acc;
  }
}

struct MfccLayout {
  size_t frames, spec, pw, mel, cep, maxkey, total;
  MfccLayout(int N, int T, int n_fft, int n_mels, int n_mfcc) {
    const size_t M = (size_t)N * T;
    const int nf = n_fft / 2 + 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += ms::align_up(bytes, 256); return at; };
    frames = take(M * n_fft * 4);
    spec = take(M * 2 * nf * 4);
    pw = take(M * nf * 4);
    mel = take(M * n_mels * 4);
    cep = take(M * n_mfcc * 4);
    maxkey = take((size_t)N * 4);
    total = o;
  }
};

unsigned stream_blocks(size_t elems) {
  size_t b = (elems + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t ms_mfcc_workspace_bytes(int N, int T, int n_fft, int n_mels, int n_mfcc) {
  if (N <= 0 || T <= 0 || n_fft <= 0 || n_mels <= 0 || n_mfcc <= 0) return 0;
  return MfccLayout(N, T, n_fft, n_mels, n_mfcc).total;
}

extern "C" int ms_mfcc_forward(const float* wave, const int32_t* wave_lens, const float* window, const float* dft,
                               const float* mel_fb, const float* dct, float* out, int N, int max_samples, int T,
                               int n_fft, int hop, int n_mels, int n_mfcc, float top_db, void* workspace,
                               size_t workspace_bytes, void* stream) {
  MS_REQUIRE(wave && wave_lens && window && dft && mel_fb && dct && out && workspace, "null pointer");
  MS_REQUIRE(N > 0 && max_samples > 0 && n_fft > 1 && hop > 0 && n_mels > 0 && n_mfcc > 0, "bad shape");
  MS_REQUIRE(n_mfcc <= n_mels, "n_mfcc must not exceed n_mels");
  MS_REQUIRE(T == 1 + max_samples / hop, "T must be 1 + max_samples / hop");
  MS_REQUIRE(N <= 65535 && T <= 2147483647 / N, "N/T exceed grid limits");
  const MfccLayout L(N, T, n_fft, n_mels, n_mfcc);
  MS_REQUIRE(workspace_bytes >= L.total, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* frames = (float*)(ws + L.frames);
  float* spec = (float*)(ws + L.spec);
  float* pw = (float*)(ws + L.pw);
  float* mel = (float*)(ws + L.mel);
  float* cep = (float*)(ws + L.cep);
  int* maxkey = (int*)(ws + L.maxkey);
  const int nf = n_fft / 2 + 1;
  const int M = N * T;

  hipLaunchKernelGGL(stft_frames_kernel, dim3(T, N), dim3(128), 0, s, wave, wave_lens, window, frames, max_samples, T,
                     n_fft, hop);
  MS_LAUNCH_CHECK();
  int rc = ms::linear_launch(frames, dft, nullptr, spec, M, n_fft, 2 * nf, MS_ACT_NONE, 0.f, 0.f, s);
  if (rc != MS_OK) return rc;
  hipLaunchKernelGGL(power_kernel, dim3(stream_blocks((size_t)M * nf)), dim3(256), 0, s, spec, pw, (size_t)M, nf);
  MS_LAUNCH_CHECK();
  rc = ms::linear_launch(pw, mel_fb, nullptr, mel, M, nf, n_mels, MS_ACT_NONE, 0.f, 0.f, s);
  if (rc != MS_OK) return rc;
  MS_HIP(hipMemsetAsync(maxkey, 0x80, (size_t)N * 4, s));
  const unsigned bx = stream_blocks((size_t)T * n_mels);
  hipLaunchKernelGGL(to_db_kernel, dim3(bx, N), dim3(256), 0, s, mel, wave_lens, maxkey, T, n_mels, hop, 10.0f, 1e-10f,
                     0.0f);
  MS_LAUNCH_CHECK();
  if (top_db >= 0.0f) {
    hipLaunchKernelGGL(db_floor_kernel, dim3(bx, N), dim3(256), 0, s, mel, maxkey, T, n_mels, top_db);
    MS_LAUNCH_CHECK();
  }
  rc = ms::linear_launch(mel, dct, nullptr, cep, M, n_mels, n_mfcc, MS_ACT_NONE, 0.f, 0.f, s);
  if (rc != MS_OK) return rc;
  hipLaunchKernelGGL(cep_to_nct_kernel, dim3(ms::cdiv(T, 32), ms::cdiv(n_mfcc, 32), N), dim3(32, 8), 0, s, cep,
                     wave_lens, out, n_mfcc, T, hop);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" size_t ms_standardize_workspace_bytes(int N) { return N > 0 ? (size_t)N * 2 * sizeof(double) : 0; }

extern "C" int ms_standardize_forward(const float* x, const int32_t* lens, float* y, int N, int inner, int T,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  MS_REQUIRE(x && y && workspace, "null pointer");
  MS_REQUIRE(N > 0 && inner > 0 && T > 0, "bad shape");
  MS_REQUIRE(N <= 65535, "N exceeds grid limits");
  MS_REQUIRE(workspace_bytes >= (size_t)N * 16, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  double* stats = (double*)workspace;
  MS_HIP(hipMemsetAsync(stats, 0, (size_t)N * 16, s));
  size_t per = (size_t)inner * T;
  unsigned bx = (unsigned)((per + 256 * 16 - 1) / (256 * 16));
  if (bx < 1) bx = 1;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(moments_kernel<0>, dim3(bx, N), dim3(256), 0, s, x, lens, stats, inner, T);
  MS_LAUNCH_CHECK();
  hipLaunchKernelGGL(moments_kernel<1>, dim3(bx, N), dim3(256), 0, s, x, lens, stats, inner, T);
  MS_LAUNCH_CHECK();
  hipLaunchKernelGGL(standardize_kernel, dim3(bx, N), dim3(256), 0, s, x, lens, stats, y, inner, T);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_context_frames_forward(const float* x, const int32_t* lens, float* y, int N, int F, int T,
                                         int n_context, void* stream) {
  MS_REQUIRE(x && y, "null pointer");
  MS_REQUIRE(N > 0 && F > 0 && T > 0 && n_context >= 0, "bad shape");
  MS_REQUIRE(N <= 65535 && (long long)F * (2 * n_context + 1) <= 65535, "N / F*(2c+1) exceed grid limits");
  hipLaunchKernelGGL(context_frames_kernel, dim3(ms::cdiv(T, 256), F * (2 * n_context + 1), N), dim3(256), 0,
                     (hipStream_t)stream, x, lens, y, F, T, n_context);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_spec_augment_(float* x, const int32_t* f_bands, const int32_t* t_bands, int N, int C, int F, int T,
                                int n_f, int n_t, void* stream) {
  MS_REQUIRE(x, "null pointer");
  MS_REQUIRE(N > 0 && C > 0 && F > 0 && T > 0 && n_f >= 0 && n_t >= 0, "bad shape");
  MS_REQUIRE((n_f == 0 || f_bands) && (n_t == 0 || t_bands), "null band table");
  MS_REQUIRE(N <= 65535 && (long long)C * F <= 65535, "N / C*F exceed grid limits");
  if (n_f == 0 && n_t == 0) return MS_OK;
  hipLaunchKernelGGL(zero_bands_kernel, dim3(ms::cdiv(T, 256), C * F, N), dim3(256), 0, (hipStream_t)stream, x,
                     f_bands, t_bands, C, F, T, n_f, n_t);
  MS_LAUNCH_CHECK();
  return MS_OK;
}

extern "C" int ms_mfcc_legacy_forward(const float* wave, const int32_t* wave_lens, const double* twiddle,
                                      const double* fbank, const double* dct, const double* lifter, float* out, int N,
                                      int max_samples, int T, int frame_len, int frame_step, int nfft, int nfilt,
                                      int numcep, double preemph, void* stream) {
  MS_REQUIRE(wave && wave_lens && twiddle && fbank && dct && lifter && out, "null pointer");
  MS_REQUIRE(N > 0 && max_samples > 0 && frame_len > 0 && frame_step > 0 && nfft > 1 && nfilt > 0 && numcep > 0,
             "bad shape");
  MS_REQUIRE(numcep <= nfilt, "numcep must not exceed nfilt");
  MS_REQUIRE(T == legacy_frames(max_samples, frame_len, frame_step), "T must be the frame count of max_samples");
  MS_REQUIRE(N <= 65535, "N exceeds grid limits");
  const size_t lds = ((size_t)3 * nfft + nfft / 2 + 1 + nfilt + 4) * sizeof(double);
  MS_REQUIRE(lds <= 64 * 1024, "nfft too large for the frame kernel");
  hipLaunchKernelGGL(mfcc_legacy_kernel, dim3(T, N), dim3(256), lds, (hipStream_t)stream, wave, wave_lens, twiddle,
                     fbank, dct, lifter, out, max_samples, T, frame_len, frame_step, nfft, nfilt, numcep, preemph);
  MS_LAUNCH_CHECK();
  return MS_OK;
}
