// Diarization segmentation kernels (PyanNet = SincNet + BiLSTM + linears, pyannote/segmentation-3.0 as
// fine-tuned by reverb-diarization-v1; call site /root/reference/diarization/infer_pyannote3.0.py:33-42).
//
// The pipeline slides a 10 s window every 1 s, so every sample is seen by 10 windows.  SincNet's first
// layer is instance-norm(waveform) -> 80 fixed band-pass filters (251 taps, stride 10): because the
// window hop (16 000 samples) is a multiple of the stride, the filter bank is evaluated ONCE on the raw
// waveform (sinc_conv) and each window applies its own normalisation as an affine correction
// (pool_norm, first block) -- 10x fewer FLOPs and no per-window copies of the audio.
#include "kernels.h"

namespace rvb {

// ------------------------------------------------------------------------------------ pcm -> float
__global__ void pcm_to_float_kernel(const int16_t* __restrict__ pcm, int64_t n, float* __restrict__ out, int64_t n_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) out[i] = i < n ? (float)pcm[i] * (1.0f / 32768.0f) : 0.0f;
}
int pcm_to_float(hipStream_t s, const int16_t* pcm, int64_t n, float* out, int64_t n_pad) {
  if (n_pad <= 0) return OK;
  hipLaunchKernelGGL(pcm_to_float_kernel, dim3(cdiv(n_pad, 256)), dim3(256), 0, s, pcm, n, out, n_pad);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ sinc filter bank
// 192 frames x 80 filters per block; filters (80 KB) and the 2161 samples they cover live in LDS.  Thread
// (filter f, slot) walks its 16 frames four at a time: per tap one conflict-free LDS read of the weight
// (row stride 251 words is odd) and four broadcast reads of samples feed four FMAs.
// 12 slots x 80 filters = 960 threads (15 waves) per workgroup: the 80 KB of filters allow one workgroup per CU, so the
// waves that hide the LDS latency have to come from inside it (320 threads: 16.5 ms per hour of audio)
static constexpr int SC_SLOTS = 24, SC_FT = 8 * SC_SLOTS, SC_NF = 80, SC_KS = 251, SC_THREADS = (SC_NF / 2) * SC_SLOTS;

template <typename O>
__global__ __launch_bounds__(SC_THREADS) void sinc_conv_kernel(const float* __restrict__ wave, const float* __restrict__ filt,
                                                               O* __restrict__ craw, int64_t n_frames, int nf,
                                                               int ksize, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sc_smem[];
  float* sf = sc_smem;                       // [nf][ksize]
  float* sx = sc_smem + SC_NF * SC_KS;       // [(FT-1)*stride + ksize]
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * SC_FT;
  const int nsamp = (SC_FT - 1) * stride + ksize;
  const int64_t last = (n_frames - 1) * stride + ksize;   // samples available
  for (int i = tid; i < nf * ksize; i += SC_THREADS) sf[i] = filt[i];
  for (int i = tid; i < nsamp; i += SC_THREADS) {
    const int64_t g = t0 * stride + i;
    sx[i] = g < last ? wave[g] : 0.0f;
  }
  __syncthreads();
  // thread = (filter pair f / f + 40, slot of 8 frames): per tap two weight reads and four broadcast sample reads feed
  // eight FMAs
  constexpr int HALF = SC_NF / 2;
  const int f = tid % HALF, slot = tid / HALF;
  if (f >= nf) return;
  const bool two = f + HALF < nf;
  const float* wf0 = sf + f * ksize;
  const float* wf1 = sf + (two ? f + HALF : f) * ksize;
#pragma unroll 1
  for (int g = 0; g < 2; ++g) {
    const int fr = slot * 8 + g * 4;
    const float* x0 = sx + fr * stride;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    for (int k = 0; k < ksize; ++k) {
      const float w0 = wf0[k], w1 = wf1[k];
      const float v0 = x0[k], v1 = x0[k + stride], v2 = x0[k + 2 * stride], v3 = x0[k + 3 * stride];
      a0 = fmaf(v0, w0, a0); a1 = fmaf(v1, w0, a1); a2 = fmaf(v2, w0, a2); a3 = fmaf(v3, w0, a3);
      b0 = fmaf(v0, w1, b0); b1 = fmaf(v1, w1, b1); b2 = fmaf(v2, w1, b2); b3 = fmaf(v3, w1, b3);
    }
    const int64_t t = t0 + fr;
    const float av[4] = {a0, a1, a2, a3}, bv[4] = {b0, b1, b2, b3};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (t + i < n_frames) {
        craw[(t + i) * nf + f] = Cvt<O>::from_f32(av[i]);
        if (two) craw[(t + i) * nf + f + HALF] = Cvt<O>::from_f32(bv[i]);
      }
  }
}

// The same filter bank on the fp32 matrix pipe (round 5): frames x taps x filters is a GEMM whose A operand is a Toeplitz view of the
// waveform -- v_mfma_f32_32x32x2_f32 takes ONE float per lane for A (frame l & 31, k-slot l >> 5) and one for B (filter l & 31, the
// same k-slot), is exact fp32 (an fmaf chain) and runs at the fp32 VECTOR peak, but without the operand traffic of the VALU form
// above (per 8 FMAs that one issues 6 LDS reads: it is LDS-bound at a third of the fp32 peak).  The two k-slots of an MFMA are
// taps j and j + 126, so that a lane walks CONSECUTIVE samples / weights: one 8-byte LDS read of each per two MFMAs.
// Workgroup = 8 waves, 1 024 frames; wave w owns frames [128 w, 128 w + 128) as four 32-frame tiles x three 32-filter tiles
// (filters 80 .. 95 are zero rows).  Accumulation order: (0, 126), (1, 127), ... -- not the tap order of the VALU form (fp32 both;
// the oracle bound of the segmentation tests is 2e-3 on the SincNet output).
static constexpr int SM_WAVES = 8, SM_TPW = 4, SM_FT = 32 * SM_TPW * SM_WAVES, SM_NFP = 96, SM_KH = 126, SM_KP = 2 * SM_KH;
typedef __attribute__((ext_vector_type(16))) float sm_f32x16;

template <typename O>
__global__ __launch_bounds__(64 * SM_WAVES) void sinc_mfma_kernel(const float* __restrict__ wave, const float* __restrict__ filt,
                                                                   O* __restrict__ craw, int64_t n_frames, int nf, int ksize, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sm_smem[];
  float* sw = sm_smem;                        // [96][252]: filter rows, taps 251 (and rows >= nf) zero
  float* sx = sm_smem + SM_NFP * SM_KP;       // [(SM_FT - 1) * stride + 252]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * SM_FT;
  const int nsamp = (SM_FT - 1) * stride + SM_KP;
  const int64_t last = (n_frames - 1) * stride + ksize;   // samples available
  for (int i = tid; i < SM_NFP * SM_KP; i += 64 * SM_WAVES) {
    const int f = i / SM_KP, k = i - f * SM_KP;
    sw[i] = (f < nf && k < ksize) ? filt[f * ksize + k] : 0.0f;
  }
  for (int i = tid; i < nsamp; i += 64 * SM_WAVES) {
    const int64_t g = t0 * stride + i;
    sx[i] = g < last ? wave[g] : 0.0f;
  }
  __syncthreads();
  const int r = lane & 31, kk = lane >> 5;
#pragma unroll 1
  for (int tt = 0; tt < SM_TPW; ++tt) {
    const int fr0 = (wv * SM_TPW + tt) * 32;
    if (t0 + fr0 >= n_frames) break;
    const float* xa = sx + (fr0 + r) * stride + kk * SM_KH;
    const float* wb = sw + r * SM_KP + kk * SM_KH;
    sm_f32x16 acc[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll 3
    for (int j = 0; j < SM_KH; j += 2) {
      const float2 a2 = *(const float2*)(xa + j);
      float2 b2[3];
#pragma unroll
      for (int n = 0; n < 3; ++n) b2[n] = *(const float2*)(wb + n * 32 * SM_KP + j);
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2[n].x, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2[n].y, acc[n], 0, 0, 0);
    }
    // D[frame][filter]: lane = filter l & 31 of the n-tile, frames 8 (e >> 2) + 4 (l >> 5) + (e & 3)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int f = n * 32 + r;
      if (f >= nf) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t t = t0 + fr0 + 8 * (e >> 2) + 4 * kk + (e & 3);
        if (t < n_frames) craw[t * nf + f] = Cvt<O>::from_f32(acc[n][e]);
      }
    }
  }
}

int sinc_conv(hipStream_t s, int dtype, const float* wave, const float* filt, void* craw, int64_t n_frames, int nf, int ksize,
              int stride) {
  if (nf > SC_NF || ksize > SC_KS || stride < 1 || stride > 16) { set_error("sinc_conv: unsupported filter bank shape"); return E_UNSUPPORTED; }
  if (n_frames <= 0) return OK;
  {
    const char* e = lab_env("RVD_SINC_MFMA");            // lab: 0 = the VALU form (until round 5)
    if (!(e && atoi(e) == 0) && nf <= SM_NFP && ksize <= SM_KP && (stride % 2) == 0) {
      const size_t lds = (size_t)(SM_NFP * SM_KP + (SM_FT - 1) * stride + SM_KP) * sizeof(float);
      static bool attr_m = false;
      if (!attr_m) {
        RVB_HIP_CHECK(hipFuncSetAttribute((const void*)sinc_mfma_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RVB_HIP_CHECK(hipFuncSetAttribute((const void*)sinc_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_m = true;
      }
      if (lds <= 160 * 1024) {
        const dim3 grid((unsigned)cdiv(n_frames, (int64_t)SM_FT)), block(64 * SM_WAVES);
        if (dtype == DT_BF16) hipLaunchKernelGGL(sinc_mfma_kernel<bf16_t>, grid, block, lds, s, wave, filt, (bf16_t*)craw, n_frames, nf, ksize, stride);
        else hipLaunchKernelGGL(sinc_mfma_kernel<float>, grid, block, lds, s, wave, filt, (float*)craw, n_frames, nf, ksize, stride);
        RVB_HIP_CHECK(hipGetLastError());
        return OK;
      }
    }
  }
  const size_t lds = (size_t)(SC_NF * SC_KS + (SC_FT - 1) * 16 + SC_KS) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)sinc_conv_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)sinc_conv_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(sinc_conv_kernel<bf16_t>, dim3(cdiv(n_frames, SC_FT)), dim3(SC_THREADS), lds, s, wave, filt, (bf16_t*)craw, n_frames, nf, ksize, stride);
  else
    hipLaunchKernelGGL(sinc_conv_kernel<float>, dim3(cdiv(n_frames, SC_FT)), dim3(SC_THREADS), lds, s, wave, filt, (float*)craw, n_frames, nf, ksize, stride);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ window statistics
__device__ inline double block_sum_d(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

__global__ __launch_bounds__(256) void window_stats_kernel(const float* __restrict__ wave, int64_t first, int64_t step, int len,
                                                           float eps, float* __restrict__ stats) {
  __shared__ double red[4];
  const float* x = wave + (first + blockIdx.x) * step;
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x * 4; i < len; i += 256 * 4) {
    if (i + 4 <= len) {
      const float4 v = *(const float4*)(x + i);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (int j = i; j < len; ++j) { s += x[j]; ss += (double)x[j] * x[j]; }
    }
  }
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  if (threadIdx.x == 0) {
    const double mean = s / len;
    double var = ss / len - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * blockIdx.x] = (float)mean;
    stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
int window_stats(hipStream_t s, const float* wave, int64_t first, int nwin, int64_t step, int len, float eps, float* stats) {
  if (nwin <= 0) return OK;
  if (step % 4) { set_error("window_stats: step must be a multiple of 4 samples"); return E_ARG; }
  hipLaunchKernelGGL(window_stats_kernel, dim3(nwin), dim3(256), 0, s, wave, first, step, len, eps, stats);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ pool + instance norm + leaky relu
// One block per window.  Thread (channel c, slot) strides over the pooled frames; pass 1 accumulates the
// per-channel mean / variance (fp64 partials), pass 2 recomputes the pooled value and writes the
// normalised activation (cheaper than a round trip of the pooled tensor through HBM).
template <typename T, bool FIRST>
__global__ __launch_bounds__(256) void pool_norm_kernel(PoolNormArgs p) {
  __shared__ double s_sum[256], s_sq[256];
  __shared__ float s_scale[128], s_shift[128];
  const int w = blockIdx.x;
  const int CP = p.ld_out;
  const int nslots = 256 / CP;
  const int c = threadIdx.x % CP, slot = threadIdx.x / CP;
  const bool active = slot < nslots && c < p.C;
  const int TP = p.frames_in / 3;

  float a = 0.f, off = 0.f;
  const T* cr = nullptr;
  const T* x = nullptr;
  if (FIRST) {
    const float mean = p.stats[2 * w], rstd = p.stats[2 * w + 1];
    a = p.wn_gamma * rstd;
    if (active) off = (p.wn_beta - a * mean) * p.fsum[c];
    cr = (const T*)p.craw + (p.craw_frame0 + (int64_t)w * p.craw_frames_per_step) * p.C + c;
  } else {
    x = (const T*)p.x + (size_t)w * p.rows_in * p.ld_in + c;
  }
  auto pooled = [&](int tp) -> float {
    if (FIRST) {
      const T* q = cr + (size_t)(3 * tp) * p.C;
      const float v0 = fabsf(fmaf(a, Cvt<T>::to_f32(q[0]), off)), v1 = fabsf(fmaf(a, Cvt<T>::to_f32(q[p.C]), off)),
                  v2 = fabsf(fmaf(a, Cvt<T>::to_f32(q[2 * p.C]), off));
      return fmaxf(v0, fmaxf(v1, v2));
    } else {
      const T* q = x + (size_t)(3 * tp) * p.ld_in;
      return fmaxf(Cvt<T>::to_f32(q[0]), fmaxf(Cvt<T>::to_f32(q[p.ld_in]), Cvt<T>::to_f32(q[2 * p.ld_in])));
    }
  };

  double s = 0.0, ss = 0.0;
  if (active)
    for (int tp = slot; tp < TP; tp += nslots) { const float v = pooled(tp); s += v; ss += (double)v * v; }
  s_sum[threadIdx.x] = s; s_sq[threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < CP) {
    if (threadIdx.x < p.C) {
      double ts = 0.0, tq = 0.0;
      for (int k = 0; k < nslots; ++k) { ts += s_sum[k * CP + threadIdx.x]; tq += s_sq[k * CP + threadIdx.x]; }
      const double mean = ts / TP;
      double var = tq / TP - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
      const float g = p.gamma[threadIdx.x] * rstd;
      s_scale[threadIdx.x] = g;
      s_shift[threadIdx.x] = p.beta[threadIdx.x] - (float)mean * g;
    } else {
      s_scale[threadIdx.x] = 0.f; s_shift[threadIdx.x] = 0.f;
    }
  }
  __syncthreads();
  if (slot >= nslots) return;
  T* out = (T*)p.out + (size_t)w * TP * CP + c;
  if (c >= p.C) {
    for (int tp = slot; tp < TP; tp += nslots) out[(size_t)tp * CP] = Cvt<T>::from_f32(0.f);
    return;
  }
  const float sc = s_scale[c], sh = s_shift[c];
  for (int tp = slot; tp < TP; tp += nslots) {
    float v = fmaf(pooled(tp), sc, sh);
    v = v > 0.f ? v : 0.01f * v;
    out[(size_t)tp * CP] = Cvt<T>::from_f32(v);
  }
}

// bf16, 16-byte vectors (round 6): thread = (group of 8 channels, slot); the scalar form above issues one 2-byte load per channel and
// pooled frame (three per pooled value, twice: statistics, then output) and is bound by load issue -- 5.2 ms per hour for the first
// layer, whose 18 GB of (cached) reads would take 3 ms at the HBM rate.  Same arithmetic per element; the fp64 partial sums of a
// channel are added over slots in a fixed order (another order than the scalar form's: 1e-16 relative on the statistics).
template <bool FIRST>
__global__ __launch_bounds__(256) void pool_norm_vec_kernel(PoolNormArgs p) {
  extern __shared__ __attribute__((aligned(16))) double pnv_red[];       // [2][nslots][CG * 8]
  __shared__ float s_scale[128], s_shift[128];
  const int w = blockIdx.x;
  const int ldi = FIRST ? p.C : p.ld_in;               // input row stride (elements)
  const int CG = p.ld_out / 8;                          // channel groups of 8 (ld_out covers C)
  const int nslots = 256 / CG;
  const int g = threadIdx.x % CG, slot = threadIdx.x / CG;
  const bool active = slot < nslots;
  const int TP = p.frames_in / 3;
  const int c0 = g * 8;

  float a = 0.f, off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) off[e] = 0.f;
  const bf16_t* src;
  if (FIRST) {
    const float mean = p.stats[2 * w], rstd = p.stats[2 * w + 1];
    a = p.wn_gamma * rstd;
#pragma unroll
    for (int e = 0; e < 8; ++e) off[e] = (c0 + e < p.C) ? (p.wn_beta - a * mean) * p.fsum[c0 + e] : 0.f;
    src = (const bf16_t*)p.craw + (p.craw_frame0 + (int64_t)w * p.craw_frames_per_step) * p.C + c0;
  } else {
    src = (const bf16_t*)p.x + (size_t)w * p.rows_in * p.ld_in + c0;
  }
  auto pooled = [&](int tp, float (&v)[8]) {
    const bf16_t* q = src + (size_t)(3 * tp) * ldi;
    const uint4 r0 = *(const uint4*)q, r1 = *(const uint4*)(q + ldi), r2 = *(const uint4*)(q + 2 * ldi);
    const bf16_t* b0 = (const bf16_t*)&r0;
    const bf16_t* b1 = (const bf16_t*)&r1;
    const bf16_t* b2 = (const bf16_t*)&r2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (FIRST) {
        const float v0 = fabsf(fmaf(a, bf16_to_f32(b0[e]), off[e])), v1 = fabsf(fmaf(a, bf16_to_f32(b1[e]), off[e])),
                    v2 = fabsf(fmaf(a, bf16_to_f32(b2[e]), off[e]));
        v[e] = fmaxf(v0, fmaxf(v1, v2));
      } else {
        v[e] = fmaxf(bf16_to_f32(b0[e]), fmaxf(bf16_to_f32(b1[e]), bf16_to_f32(b2[e])));
      }
    }
  };
  double s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.0; ss[e] = 0.0; }
  if (active)
    for (int tp = slot; tp < TP; tp += nslots) {
      float v[8];
      pooled(tp, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += v[e]; ss[e] += (double)v[e] * v[e]; }
    }
  double* r_sum = pnv_red;
  double* r_sq = pnv_red + nslots * CG * 8;
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { r_sum[slot * CG * 8 + c0 + e] = s[e]; r_sq[slot * CG * 8 + c0 + e] = ss[e]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < CG * 8) {
    const int c = threadIdx.x;
    if (c < p.C) {
      double ts = 0.0, tq = 0.0;
      for (int k = 0; k < nslots; ++k) { ts += r_sum[k * CG * 8 + c]; tq += r_sq[k * CG * 8 + c]; }
      const double mean = ts / TP;
      double var = tq / TP - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
      const float gm = p.gamma[c] * rstd;
      s_scale[c] = gm;
      s_shift[c] = p.beta[c] - (float)mean * gm;
    } else {
      s_scale[c] = 0.f; s_shift[c] = 0.f;
    }
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = s_scale[c0 + e]; sh[e] = s_shift[c0 + e]; }
  bf16_t* out = (bf16_t*)p.out + (size_t)w * TP * p.ld_out + c0;
  for (int tp = slot; tp < TP; tp += nslots) {
    float v[8];
    pooled(tp, v);
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float y = fmaf(v[e], sc[e], sh[e]);
      y = y > 0.f ? y : 0.01f * y;
      o[e] = (c0 + e < p.C) ? f32_to_bf16(y) : f32_to_bf16(0.f);
    }
    *(uint4*)(out + (size_t)tp * p.ld_out) = *(const uint4*)o;
  }
}

int pool_norm(hipStream_t s, int dtype, const PoolNormArgs& a) {
  if (a.W <= 0) return OK;
  if (a.ld_out > 128 || a.ld_out < a.C || a.C < 1) { set_error("pool_norm: channels must be <= ld_out <= 128"); return E_ARG; }
  const bool first = a.x == nullptr;
  {
    const int vec = lab_env("RVD_POOLNORM_VEC") ? atoi(lab_env("RVD_POOLNORM_VEC")) : 1;      // lab: 0 = the scalar form (read per call: the A/B test flips it)
    const int ldi = first ? a.C : a.ld_in;
    const uintptr_t base = first ? (uintptr_t)a.craw : (uintptr_t)a.x;
    if (vec && dtype == DT_BF16 && (ldi % 8) == 0 && (a.ld_out % 8) == 0 && a.ld_out >= 8 && (base % 16) == 0 && ((uintptr_t)a.out % 16) == 0 &&
        (!first || ((a.craw_frame0 * a.C) % 8 == 0 && ((int64_t)a.craw_frames_per_step * a.C) % 8 == 0)) &&
        (first || ((size_t)a.rows_in * a.ld_in) % 8 == 0)) {
      const int CG = a.ld_out / 8, nslots = 256 / CG;
      const size_t lds = (size_t)2 * nslots * CG * 8 * sizeof(double);
      if (first) hipLaunchKernelGGL((pool_norm_vec_kernel<true>), dim3(a.W), dim3(256), lds, s, a);
      else hipLaunchKernelGGL((pool_norm_vec_kernel<false>), dim3(a.W), dim3(256), lds, s, a);
      RVB_HIP_CHECK(hipGetLastError());
      return OK;
    }
  }
  if (dtype == DT_BF16) {
    if (first) hipLaunchKernelGGL((pool_norm_kernel<bf16_t, true>), dim3(a.W), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pool_norm_kernel<bf16_t, false>), dim3(a.W), dim3(256), 0, s, a);
  } else {
    if (first) hipLaunchKernelGGL((pool_norm_kernel<float, true>), dim3(a.W), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pool_norm_kernel<float, false>), dim3(a.W), dim3(256), 0, s, a);
  }
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ SincNet conv layers 2 and 3 (kernel 5, 60 filters)
// out[m][n] = bias[n] + sum_{k < 5 CIN} A[m * CIN + k] * W[n][k]: row m of the im2col matrix is frames m .. m + 4 of the activation
// tensor [rows][CIN], which are contiguous -- no im2col buffer.  As a GEMM this is [19 M x 400] x [400 x 64]: too thin for the 256 x 256
// tile (three quarters of its columns idle) and, at K = 400, not a multiple of its K step; on the generic 128 x 128 kernel it ran at
// 160 TFLOP/s, 6 ms per hour for layer 2.  Here: persistent workgroups (8 waves), the 64 x (5 CIN) weight matrix resident in LDS, 256
// frames per tile staged through registers into one of two LDS buffers (rows padded so that 16 rows hit 16 different bank groups) while
// the previous tile is multiplied; wave = 64 frames x 32 filters (v_mfma_f32_16x16x32_bf16, the weights as the A operand so that a lane
// ends up with four consecutive filters of one frame: 8-byte stores).  K = 400 is 12.5 MFMA steps: the lanes of the last half step feed zeros.
template <int CIN>
__global__ __launch_bounds__(512) void conv1d5_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ out, int64_t M) {
  constexpr int K = 5 * CIN, KS = (K + 31) / 32, KP = KS * 32;
  constexpr int WS = KP * 2 + 16;                 // weight row pitch (bytes): an odd multiple of 16 mod 256
  constexpr int RS = CIN * 2 + 16;                // frame row pitch (bytes): 176 / 144, odd multiples of 16
  constexpr int TM = 256, ROWS = TM + 4;
  constexpr int VPR = CIN / 8;                    // 16-byte vectors per frame
  constexpr int NV = (ROWS * VPR + 511) / 512;
  static_assert((WS / 16) % 2 == 1 && (RS / 16) % 2 == 1, "bank-conflict-free pitches");
  extern __shared__ __attribute__((aligned(16))) char c5_smem[];
  char* sW = c5_smem;                             // [64][WS]
  char* sA = c5_smem + 64 * WS;                   // [2][ROWS][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  // weights -> LDS (zero beyond K)
  for (int i = tid; i < 64 * (KP / 8); i += 512) {
    const int n = i / (KP / 8), v = i - n * (KP / 8);
    uint4 w = make_uint4(0, 0, 0, 0);
    if (v * 8 < K) w = *(const uint4*)(Wt + (size_t)n * K + v * 8);
    *(uint4*)(sW + n * WS + v * 16) = w;
  }
  const int64_t ntiles = (M + TM - 1) / TM;
  uint4 ra[NV];
  auto gload = [&](int64_t tile) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * 512;
      const int r = v / VPR, c = v - r * VPR;
      int64_t row = tile * TM + r;
      row = row < M + 3 ? row : M + 3;            // the buffer holds M + 8 rows; what lies past M feeds only the rows nobody reads
      ra[i] = make_uint4(0, 0, 0, 0);
      if (v < ROWS * VPR) ra[i] = *(const uint4*)(A + row * CIN + c * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * 512;
      if (v < ROWS * VPR) { const int r = v / VPR, c = v - r * VPR; *(uint4*)(sA + (size_t)buf * ROWS * RS + r * RS + c * 16) = ra[i]; }
    }
  };
  // this lane's byte offset inside its frame row for MFMA step ks (k = 32 ks + 8 lgrp: frame k / CIN, channel k % CIN)
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { const int k = 32 * ks + 8 * lgrp; koff[ks] = (k / CIN) * RS + (k % CIN) * 2; }
  const int wm = wave >> 1, wn = wave & 1;        // wave: frames [64 wm, +64), filters [32 wn, +32)
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = bias[wn * 32 + j * 16 + lgrp * 4 + r];
  int64_t tile = blockIdx.x;
  if (tile < ntiles) gload(tile);
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    lstore(buf);
    __syncthreads();
    if (tile + gridDim.x < ntiles) gload(tile + gridDim.x);
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){bv[j][0], bv[j][1], bv[j][2], bv[j][3]};
    const char* ab = sA + (size_t)buf * ROWS * RS + (wm * 64 + lrow) * RS;
    const char* wb = sW + (wn * 32 + lrow) * WS + lgrp * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 wf[2], af[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *(const uint4*)(wb + j * 16 * WS + ks * 64);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *(const uint4*)(ab + i * 16 * RS + koff[ks]);
        if (K % 32 != 0 && ks == KS - 1 && 32 * ks + 8 * lgrp >= K) af[i] = make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          union U { uint4 u; bf16x8_t v; } ua, ub;
          ua.u = wf[j]; ub.u = af[i];
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[i][j], 0, 0, 0);
        }
    }
    // D^T fragment (i, j): lane = frame 16 i + lrow, filters 16 j + 4 lgrp + r
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = tile * TM + wm * 64 + i * 16 + lrow;
      if (m < M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint2 pk;
          pk.x = pack2_bf16(acc[i][j][0], acc[i][j][1]);
          pk.y = pack2_bf16(acc[i][j][2], acc[i][j][3]);
          *(uint2*)(out + m * 64 + wn * 32 + j * 16 + lgrp * 4) = pk;
        }
      }
    }
  }
}

int conv1d5(hipStream_t s, int dtype, const void* A, int cin, const void* W, const float* bias, void* out, int64_t M) {
  if (dtype != DT_BF16 || (cin != 80 && cin != 64)) { set_error("conv1d5: built for bf16 and 80 or 64 input channels"); return E_UNSUPPORTED; }
  if (M <= 0) return OK;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    RVB_HIP_CHECK(hipGetDevice(&dev));
    RVB_HIP_CHECK(hipGetDeviceProperties(&pr, dev));
    ncu = pr.multiProcessorCount;
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv1d5_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv1d5_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const int64_t ntiles = (M + 255) / 256;
  const int grid = (int)std::min<int64_t>(ntiles, ncu);
  if (cin == 80) {
    const size_t lds = 64 * (13 * 64 + 16) + 2 * 260 * (160 + 16);
    hipLaunchKernelGGL(conv1d5_kernel<80>, dim3(grid), dim3(512), lds, s, (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)out, M);
  } else {
    const size_t lds = 64 * (10 * 64 + 16) + 2 * 260 * (128 + 16);
    hipLaunchKernelGGL(conv1d5_kernel<64>, dim3(grid), dim3(512), lds, s, (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)out, M);
  }
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ LSTM recurrence
// One block = 16 windows x one direction, 4 waves; wave v owns hidden units [32v, 32v+32) and therefore the
// four gate columns {q*128 + unit} of each: the i/f/g/o pre-activations of a (window, unit) pair land in
// the same lane of four accumulator fragments, so the cell update is lane-local.  The recurrent weights
// (4H x H) stay in registers as MFMA B fragments for all 589 steps (bf16: 128 VGPRs per lane); h_t goes
// through a double-buffered 4 KB LDS tile to become the next step's A operand; the input projection of
// step t+1 is prefetched while step t runs (its columns permuted at load time so that a lane reads one vector per window).
template <typename T> struct GateMath;
template <> struct GateMath<float> {
  __device__ static inline float sig(float x) { return 1.0f / (1.0f + expf(-x)); }
  __device__ static inline float tanh_(float x) { return tanhf(x); }
};
template <> struct GateMath<bf16_t> {
  __device__ static inline float sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
  __device__ static inline float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x)); }
};

template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void lstm_kernel(const T* __restrict__ xproj, const T* __restrict__ whh, T* __restrict__ out,
                                                   int W, int TT) {
  constexpr int H = 128, NB = 16;
  constexpr int KC = Mma16<T>::KC, VE = Mma16<T>::VE, NCH = H / KC;
  constexpr int HSB = H * (int)sizeof(T) + 16;          // padded LDS row stride (bytes)
  constexpr bool WREG = sizeof(T) == 2;
  __shared__ __attribute__((aligned(16))) char hs[2][NB * HSB];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int dir = blockIdx.y;
  const int w0 = blockIdx.x * NB;
  const int col = lane & 15, grp = lane >> 4;
  const T* whd = whh + (size_t)dir * 4 * H * H;

  auto wptr = [&](int q, int hf, int kc) { return (const uint4*)(whd + (size_t)(q * H + 32 * wv + 16 * hf + col) * H + kc * KC + grp * VE); };
  uint4 wreg[WREG ? 4 : 1][2][WREG ? NCH : 1];
  if (WREG) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) wreg[WREG ? q : 0][hf][WREG ? kc : 0] = *wptr(q, hf, kc);
  }

  size_t xrow[4];
  bool wok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int wi = w0 + 4 * grp + r;
    wok[r] = wi < W;
    xrow[r] = (size_t)(wok[r] ? wi : W - 1) * TT;
  }
  const int ucol = 32 * wv + col;     // + 16*hf
  float xpv[4][2][4];
  // the projection's columns come in the order this kernel reads them (diar_engine.hip: column 128 v + 8 c + 2 q + hf of a direction):
  // a lane's eight starting values of a window and step are one 16-byte vector (two for fp32)
  auto load_xp = [&](int t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const T* xp = xproj + (xrow[r] + t) * (size_t)(8 * H) + dir * 4 * H + 128 * wv + 8 * col;
      T v8[8];
      if constexpr (sizeof(T) == 2) { *(uint4*)v8 = *(const uint4*)xp; }
      else { *(uint4*)v8 = *(const uint4*)xp; *(uint4*)(v8 + 4) = *(const uint4*)(xp + 4); }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) xpv[q][hf][r] = Cvt<T>::to_f32(v8[2 * q + hf]);
    }
  };

  for (int i = threadIdx.x; i < NB * HSB / 4; i += 256) ((uint32_t*)hs[0])[i] = 0u;
  float cst[2][4];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cst[hf][r] = 0.f;
  load_xp(dir ? TT - 1 : 0);
  __syncthreads();

  for (int s = 0; s < TT; ++s) {
    const int t = dir ? TT - 1 - s : s;
    const int cur = s & 1;
    f32x4_t acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) acc[q][hf] = (f32x4_t){xpv[q][hf][0], xpv[q][hf][1], xpv[q][hf][2], xpv[q][hf][3]};
    if (s + 1 < TT) load_xp(dir ? t - 1 : t + 1);

    const char* ha = hs[cur] + col * HSB + grp * 16;
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const uint4 a = *(const uint4*)(ha + kc * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const uint4 b = WREG ? wreg[WREG ? q : 0][hf][WREG ? kc : 0] : *wptr(q, hf, kc);
          Mma16<T>::run(a, b, acc[q][hf]);
        }
    }
    char* hn = hs[cur ^ 1];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int unit = ucol + 16 * hf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = GateMath<T>::sig(acc[0][hf][r]);
        const float fg = GateMath<T>::sig(acc[1][hf][r]);
        const float gg = GateMath<T>::tanh_(acc[2][hf][r]);
        const float og = GateMath<T>::sig(acc[3][hf][r]);
        const float c = fg * cst[hf][r] + ig * gg;
        cst[hf][r] = c;
        const T hv = Cvt<T>::from_f32(og * GateMath<T>::tanh_(c));
        *(T*)(hn + (4 * grp + r) * HSB + unit * sizeof(T)) = hv;
        if (wok[r]) out[(xrow[r] + t) * (size_t)(2 * H) + dir * H + unit] = hv;
      }
    }
    __syncthreads();
  }
}

int lstm_recurrence(hipStream_t s, int dtype, const void* xproj, const void* whh, void* out, int W, int T) {
  if (W <= 0 || T <= 0) return OK;
  const dim3 grid(cdiv(W, 16), 2);
  if (dtype == DT_BF16) hipLaunchKernelGGL(lstm_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)xproj, (const bf16_t*)whh, (bf16_t*)out, W, T);
  else hipLaunchKernelGGL(lstm_kernel<float>, grid, dim3(256), 0, s, (const float*)xproj, (const float*)whh, (float*)out, W, T);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ classifier + log-softmax
template <typename T>
__global__ __launch_bounds__(256) void classifier_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ logp,
                                                         uint8_t* __restrict__ cls, int64_t M, int in, int C) {
  __shared__ float sw[16 * 256];
  for (int i = threadIdx.x; i < C * in; i += 256) sw[i] = w[i];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= M) return;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = (c < C) ? b[c] : 0.f;
  const T* xr = x + row * ldx;
  constexpr int VE = 16 / sizeof(T);
  for (int k0 = 0; k0 < in; k0 += VE) {
    const uint4 raw = *(const uint4*)(xr + k0);
    const T* xe = (const T*)&raw;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const float xv = Cvt<T>::to_f32(xe[e]);
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < C) acc[c] = fmaf(xv, sw[c * in + k0 + e], acc[c]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 16; ++c) if (c < C) mx = fmaxf(mx, acc[c]);
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) if (c < C) se += expf(acc[c] - mx);
  const float lse = mx + logf(se);
#pragma unroll
  for (int c = 0; c < 16; ++c) if (c < C) logp[row * C + c] = acc[c] - lse;
  if (cls) {                       // argmax class, first index on ties (torch.argmax / np.argmax)
    int best = 0;
#pragma unroll
    for (int c = 1; c < 16; ++c) if (c < C && acc[c] > acc[best]) best = c;
    cls[row] = (uint8_t)best;
  }
}

int classifier_logsoftmax(hipStream_t s, int dtype, const void* x, int ldx, const float* w, const float* b, float* logp,
                          uint8_t* cls, int64_t M, int in, int C) {
  if (C > 16 || in > 256 || (in % 8) || (ldx % 8)) { set_error("classifier_logsoftmax: needs C <= 16, in <= 256, in and ldx multiples of 8"); return E_ARG; }
  if (M <= 0) return OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(classifier_kernel<bf16_t>, dim3(cdiv(M, 256)), dim3(256), 0, s, (const bf16_t*)x, ldx, w, b, logp, cls, M, in, C);
  else hipLaunchKernelGGL(classifier_kernel<float>, dim3(cdiv(M, 256)), dim3(256), 0, s, (const float*)x, ldx, w, b, logp, cls, M, in, C);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
