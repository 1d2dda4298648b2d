// Kaldi-compatible 80-bin log-mel filterbank (25 ms / 10 ms, povey window, 512-point FFT).
//
// Replaces `torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80, frame_length=25,
// frame_shift=10, dither=0.0, energy_floor=0.0, sample_frequency=16000)` at
// asr/wenet/cli/reverb.py:136-144 (torchaudio 2.2.2 defaults: snip_edges, remove_dc_offset,
// preemphasis 0.97, use_power, log of max(., eps)).
//
// HBM-bound and tiny (32 KB/s of audio in, 32 KB/s of features out): one wave64 per frame,
// four frames per workgroup; the 400-sample window is read coalesced, the radix-2 FFT runs
// in LDS (4 butterflies per lane per stage), the mel projection reads the power spectrum
// from LDS with each lane owning mel bins {lane, lane+64}.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/rvb.h"
#include "common.h"
#include "kernels.h"

namespace rvb {

static constexpr int WIN = 400, SHIFT = 160, NFFT = 512, NBIN = 257, NMEL = 80;

template <typename In>
__global__ __launch_bounds__(256) void fbank_kernel(const In* __restrict__ pcm, int64_t n_frames,
                                                    float* __restrict__ feats, FbankTables t) {
  __shared__ float s_re[4][NFFT];
  __shared__ float s_im[4][NFFT];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t frame = (int64_t)blockIdx.x * 4 + w;
  const bool live = frame < n_frames;
  float* re = s_re[w];
  float* im = s_im[w];

  // 1. load window, remove DC
  float x[7];
  float sum = 0.f;
  const In* src = pcm + frame * SHIFT;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int j = lane + 64 * i;
    x[i] = (live && j < WIN) ? (float)src[j] : 0.f;
    sum += x[i];
  }
  const float mean = wave_sum(sum) / (float)WIN;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int j = lane + 64 * i;
    if (j < WIN) im[j] = x[i] - mean;   // stage DC-removed samples in `im`
  }
  __syncthreads();
  // 2. pre-emphasis (x[-1] := x[0]) + povey window, written bit-reversed into re; im := 0
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    float v = 0.f;
    if (j < WIN) {
      const float cur = im[j];
      const float prev = im[j > 0 ? j - 1 : 0];
      v = (cur - 0.97f * prev) * t.window[j];
    }
    y[i] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    re[__brev((unsigned)j) >> 23] = y[i];
    im[j] = 0.f;
  }
  __syncthreads();
  // 3. 512-point radix-2 DIT FFT
#pragma unroll 1
  for (int s = 0; s < 9; ++s) {
    const int half = 1 << s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = lane + 64 * q;
      const int j = b & (half - 1);
      const int i0 = ((b >> s) << (s + 1)) + j;
      const int i1 = i0 + half;
      const int tw = j << (8 - s);
      const float wr = t.twiddle[2 * tw], wi = t.twiddle[2 * tw + 1];
      const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
      const float tr = wr * br - wi * bi;
      const float ti = wr * bi + wi * br;
      re[i0] = ar + tr; im[i0] = ai + ti;
      re[i1] = ar - tr; im[i1] = ai - ti;
    }
    __syncthreads();
  }
  // 4. power spectrum (bins 0..256) into re
  float pw[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = lane + 64 * i;
    pw[i] = (j < NBIN) ? re[j] * re[j] + im[j] * im[j] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = lane + 64 * i;
    if (j < NBIN) re[j] = pw[i];
  }
  __syncthreads();
  // 5. mel projection + log
  if (live) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = lane + 64 * i;
      if (m < NMEL) {
        const int lo = t.mel_lo[m], hi = t.mel_hi[m];
        const float* wrow = t.mel_w + (size_t)m * NBIN;
        float acc = 0.f;
        for (int b = lo; b < hi; ++b) acc += re[b] * wrow[b];
        feats[frame * NMEL + m] = logf(fmaxf(acc, 1.1920928955078125e-07f));
      }
    }
  }
}

int fbank(hipStream_t s, const int16_t* pcm, int64_t n_frames, float* feats, const FbankTables& t) {
  if (n_frames <= 0) return OK;
  hipLaunchKernelGGL(fbank_kernel<int16_t>, dim3(cdiv(n_frames, 4)), dim3(256), 0, s, pcm, n_frames, feats, t);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// same on a float waveform at int16 scale (the output of the resampler: reverb.py:128-134 resamples the
// `.to(float)` waveform and hands the un-rounded result to kaldi.fbank)
int fbank_f32(hipStream_t s, const float* wave, int64_t n_frames, float* feats, const FbankTables& t) {
  if (n_frames <= 0) return OK;
  hipLaunchKernelGGL(fbank_kernel<float>, dim3(cdiv(n_frames, 4)), dim3(256), 0, s, wave, n_frames, feats, t);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// torchaudio.transforms.Resample(orig, new) (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) as a
// polyphase FIR: out[j*new + p] = sum_k ker[p][k] * x[j*orig + k - width], x zero outside [0, n_in).
// One thread per output sample; the K-tap windows of neighbouring outputs overlap almost entirely (L1/L2).
template <typename In>
__global__ __launch_bounds__(256) void resample_kernel(const In* __restrict__ pcm, int64_t n_in, const float* __restrict__ ker,
                                                       int orig, int new_, int width, int K, float* __restrict__ out,
                                                       int64_t n_out) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  const int64_t j = n / new_;
  const int p = (int)(n - j * new_);
  const int64_t base = j * orig - width;
  const float* kr = ker + (size_t)p * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int64_t i = base + k;
    if (i >= 0 && i < n_in) acc = fmaf(kr[k], (float)pcm[i], acc);
  }
  out[n] = acc;
}
int resample(hipStream_t s, const int16_t* pcm, int64_t n_in, const float* ker, int orig, int new_, int width, int K, float* out,
             int64_t n_out) {
  if (n_out <= 0) return OK;
  hipLaunchKernelGGL(resample_kernel<int16_t>, dim3(cdiv(n_out, 256)), dim3(256), 0, s, pcm, n_in, ker, orig, new_, width, K, out, n_out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int resample_f32(hipStream_t s, const float* pcm, int64_t n_in, const float* ker, int orig, int new_, int width, int K, float* out,
             int64_t n_out) {
  if (n_out <= 0) return OK;
  hipLaunchKernelGGL(resample_kernel<float>, dim3(cdiv(n_out, 256)), dim3(256), 0, s, pcm, n_in, ker, orig, new_, width, K, out, n_out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

void resample_taps(int sample_rate, int target, std::vector<float>* ker, int* orig_out, int* new_out, int* width_out, int* K_out) {
  int a = sample_rate, b = target;
  while (b) { const int t = a % b; a = b; b = t; }
  const int orig = sample_rate / a, nw = target / a;
  const double lpw = 6.0, rolloff = 0.99;
  const double base_freq = std::min(orig, nw) * rolloff;
  const int width = (int)std::ceil(lpw * orig / base_freq);
  const int K = 2 * width + orig;
  ker->assign((size_t)nw * K, 0.f);
  const double PI = 3.14159265358979323846;
  for (int p = 0; p < nw; ++p)
    for (int k = 0; k < K; ++k) {
      double t = (-(double)p / nw + (double)(k - width) / orig) * base_freq;
      t = std::max(-lpw, std::min(lpw, t));
      const double c = std::cos(t * PI / lpw / 2.0);
      const double window = c * c;
      t *= PI;
      const double sinc = t == 0.0 ? 1.0 : std::sin(t) / t;
      (*ker)[(size_t)p * K + k] = (float)(sinc * window * (base_freq / orig));
    }
  *orig_out = orig; *new_out = nw; *width_out = width; *K_out = K;
}

// ------------------------------------------------------------------------------------------------
// compute_feats with OTHER front-end settings (cli/reverb.py:119-146 takes any num_mel_bins / frame_length / frame_shift; the
// model's own configuration is the kernel above).  Same algorithm with the three sizes as arguments: window of 257 .. 512
// samples (frame_length 16.1 .. 32 ms at 16 kHz: Kaldi pads to the next power of two, i.e. the same 512-point FFT), any shift,
// up to 128 mel bins.  Not on the hot path: host tables are built per call, the waveform comes from the host as float.
struct FbankAny { int win, shift, nmel; };
__global__ __launch_bounds__(256) void fbank_any_kernel(const float* __restrict__ wave, int64_t n_frames, float* __restrict__ feats,
                                                        FbankTables t, FbankAny g) {
  __shared__ float s_re[4][NFFT];
  __shared__ float s_im[4][NFFT];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t frame = (int64_t)blockIdx.x * 4 + w;
  const bool live = frame < n_frames;
  float* re = s_re[w];
  float* im = s_im[w];
  float x[8];
  float sum = 0.f;
  const float* src = wave + frame * g.shift;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    x[i] = (live && j < g.win) ? src[j] : 0.f;
    sum += x[i];
  }
  const float mean = wave_sum(sum) / (float)g.win;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    if (j < g.win) im[j] = x[i] - mean;
  }
  __syncthreads();
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    float v = 0.f;
    if (j < g.win) v = (im[j] - 0.97f * im[j > 0 ? j - 1 : 0]) * t.window[j];
    y[i] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    re[__brev((unsigned)j) >> 23] = y[i];
    im[j] = 0.f;
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < 9; ++s) {
    const int half = 1 << s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = lane + 64 * q;
      const int j = b & (half - 1);
      const int i0 = ((b >> s) << (s + 1)) + j;
      const int i1 = i0 + half;
      const int tw = j << (8 - s);
      const float wr = t.twiddle[2 * tw], wi = t.twiddle[2 * tw + 1];
      const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
      const float tr = wr * br - wi * bi;
      const float ti = wr * bi + wi * br;
      re[i0] = ar + tr; im[i0] = ai + ti;
      re[i1] = ar - tr; im[i1] = ai - ti;
    }
    __syncthreads();
  }
  float pw[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = lane + 64 * i;
    pw[i] = (j < NBIN) ? re[j] * re[j] + im[j] * im[j] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = lane + 64 * i;
    if (j < NBIN) re[j] = pw[i];
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = lane + 64 * i;
      if (m < g.nmel) {
        const int lo = t.mel_lo[m], hi = t.mel_hi[m];
        const float* wrow = t.mel_w + (size_t)m * NBIN;
        float acc = 0.f;
        for (int b = lo; b < hi; ++b) acc += re[b] * wrow[b];
        feats[frame * g.nmel + m] = logf(fmaxf(acc, 1.1920928955078125e-07f));
      }
    }
  }
}

// Kaldi mel banks / povey window / FFT twiddles for (win samples, nmel bins) at 16 kHz with a 512-point FFT
// (oracle/fbank_ref.py restates torchaudio's; engine.hip make_fbank_tables is the 400 / 80 instance)
void fbank_host_tables(int win, int nmel, std::vector<float>* window, std::vector<float>* tw, std::vector<float>* melw,
                       std::vector<int32_t>* lo, std::vector<int32_t>* hi) {
  const double PI = 3.14159265358979323846;
  window->resize(win); tw->resize(2 * 256); melw->assign((size_t)nmel * NBIN, 0.f);
  lo->assign(nmel, NBIN); hi->assign(nmel, 0);
  for (int i = 0; i < win; ++i) (*window)[i] = (float)std::pow(0.5 - 0.5 * std::cos(2.0 * PI * i / (win - 1)), 0.85);
  for (int k = 0; k < 256; ++k) { (*tw)[2 * k] = (float)std::cos(2.0 * PI * k / NFFT); (*tw)[2 * k + 1] = (float)(-std::sin(2.0 * PI * k / NFFT)); }
  auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
  const double mlo = mel(20.0), mhi = mel(8000.0), delta = (mhi - mlo) / (nmel + 1);
  for (int m = 0; m < nmel; ++m) {
    const double left = mlo + m * delta, center = left + delta, right = center + delta;
    for (int b = 0; b < NFFT / 2; ++b) {
      const double mf = mel(16000.0 / NFFT * b);
      const double up = (mf - left) / (center - left), down = (right - mf) / (right - center);
      const double wgt = std::max(0.0, std::min(up, down));
      if (wgt > 0.0) {
        (*melw)[(size_t)m * NBIN + b] = (float)wgt;
        (*lo)[m] = std::min((*lo)[m], b); (*hi)[m] = std::max((*hi)[m], b + 1);
      }
    }
    if ((*hi)[m] == 0) (*lo)[m] = 0;
  }
}

int fbank_any(hipStream_t s, const float* wave, int64_t n_frames, float* feats, const FbankTables& t, int win, int shift, int nmel) {
  if (n_frames <= 0) return OK;
  const FbankAny g{win, shift, nmel};
  hipLaunchKernelGGL(fbank_any_kernel, dim3(cdiv(n_frames, 4)), dim3(256), 0, s, wave, n_frames, feats, t, g);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

__global__ __launch_bounds__(256) void round_i16_kernel(const float* __restrict__ x, int64_t n, int16_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int16_t)__builtin_rintf(fminf(fmaxf(x[i], -32768.f), 32767.f));
}
int round_to_i16(hipStream_t s, const float* x, int64_t n, int16_t* out) {
  if (n <= 0) return OK;
  hipLaunchKernelGGL(round_i16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, n, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb

// ReverbASR.compute_feats with front-end settings other than the model's (cli/reverb.py:119-146): see include/rvb.h
extern "C" int rvb_compute_feats(int device, const float* wave, int64_t n_samples, int num_mel_bins, double frame_length_ms,
                                 double frame_shift_ms, float* feats_out, int64_t* n_frames) {
  using namespace rvb;
  if (!n_frames || (!wave && n_samples > 0) || n_samples < 0) { set_error("rvb_compute_feats: bad argument"); return E_ARG; }
  const int win = (int)(16000.0 * frame_length_ms * 0.001), shift = (int)(16000.0 * frame_shift_ms * 0.001);
  if (win <= 256 || win > 512 || shift < 1 || num_mel_bins < 1 || num_mel_bins > 128) {
    set_error("rvb_compute_feats: supported are frame_length 16.1 .. 32 ms (a 512-point FFT at 16 kHz), frame_shift >= 1 sample, 1 .. 128 mel bins");
    return E_UNSUPPORTED;
  }
  const int64_t nf = n_samples < win ? 0 : 1 + (n_samples - win) / shift;
  *n_frames = nf;
  if (!feats_out || nf == 0) return OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) { set_error("rvb_compute_feats: no HIP device (this library has no CPU fallback)"); return E_HIP; }
  RVB_HIP_CHECK(hipSetDevice(device));
  std::vector<float> window, tw, melw;
  std::vector<int32_t> lo, hi;
  fbank_host_tables(win, num_mel_bins, &window, &tw, &melw, &lo, &hi);
  void *dwave = nullptr, *dwin = nullptr, *dtw = nullptr, *dmel = nullptr, *dlo = nullptr, *dhi = nullptr, *dout = nullptr;
  auto up = [](void** d, const void* h, size_t bytes) { return hipMalloc(d, bytes) == hipSuccess && hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess; };
  int rc = OK;
  if (!up(&dwave, wave, (size_t)n_samples * 4) || !up(&dwin, window.data(), window.size() * 4) || !up(&dtw, tw.data(), tw.size() * 4) ||
      !up(&dmel, melw.data(), melw.size() * 4) || !up(&dlo, lo.data(), lo.size() * 4) || !up(&dhi, hi.data(), hi.size() * 4) ||
      hipMalloc(&dout, (size_t)nf * num_mel_bins * 4) != hipSuccess) {
    set_error("rvb_compute_feats: device allocation / upload failed"); rc = E_NOMEM;
  }
  if (rc == OK) {
    const FbankTables t{(const float*)dwin, (const float*)dtw, (const float*)dmel, (const int*)dlo, (const int*)dhi};
    rc = fbank_any(nullptr, (const float*)dwave, nf, (float*)dout, t, win, shift, num_mel_bins);
    if (rc == OK && hipDeviceSynchronize() != hipSuccess) { set_error("rvb_compute_feats: kernel failed"); rc = E_HIP; }
    if (rc == OK && hipMemcpy(feats_out, dout, (size_t)nf * num_mel_bins * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("rvb_compute_feats: download failed"); rc = E_HIP; }
  }
  for (void* q : {dwave, dwin, dtw, dmel, dlo, dhi, dout}) if (q) (void)hipFree(q);
  return rc;
}
