// Speaker-embedding network kernels: WeSpeaker ResNet34 (BasicBlock [3,4,6,3], 32..256 channels) + masked
// statistics pooling, as pyannote's SpeakerDiarization pipeline runs it on every (window, local speaker)
// pair (reference call site /root/reference/diarization/infer_pyannote3.0.py:40).
//
// Layout: activations are NHWC with a one-pixel zero border, [B][F+2][T+2][C] in the compute dtype, so the
// 3x3 / pad 1 convolutions read without bounds checks and each pixel's channels are one contiguous run
// (C*2 bytes = one or more 64-byte MFMA K chunks).  BatchNorm is folded into the conv weights and bias.
//
// conv_kernel: direct convolution on MFMA.  A block owns 4 x 64 output pixels (freq x time) and NT output
// channels; per 64-byte chunk of input channels it stages the (4s+3-s) x (64s+3-s) input patch and the 9
// taps' weights in LDS once, and the 9 taps read shifted A fragments from the same patch (16 consecutive
// time positions x 64 bytes = a conflict-free 1 KiB run), i.e. every input byte is fetched from HBM/L2
// once per block instead of 9 times as an im2col GEMM would.
#include "kernels.h"

namespace rvb {

// ------------------------------------------------------------------------------------ per-window CMN statistics
// mean over the window's frames of each mel bin (pyannote WeSpeaker wrapper: features - features.mean(dim=1))
__global__ __launch_bounds__(256) void emb_mean_kernel(const float* __restrict__ fb, const int64_t* __restrict__ win,
                                                       int frames_per_step, int nfr, float* __restrict__ mean) {
  __shared__ float red[3][80];
  const int b = blockIdx.x;
  const int bin = threadIdx.x % 80, slot = threadIdx.x / 80;
  const float* x = fb + (size_t)win[b] * frames_per_step * 80;
  float s = 0.f;
  if (slot < 3)
    for (int t = slot; t < nfr; t += 3) s += x[(size_t)t * 80 + bin];
  if (slot < 3) red[slot][bin] = s;
  __syncthreads();
  if (threadIdx.x < 80) mean[b * 80 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x]) / (float)nfr;
}
int emb_window_mean(hipStream_t s, const float* fb, const int64_t* win, int B, int frames_per_step, int nfr, float* mean) {
  if (B <= 0) return OK;
  hipLaunchKernelGGL(emb_mean_kernel, dim3(B), dim3(256), 0, s, fb, win, frames_per_step, nfr, mean);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ stem: Conv2d(1, C, 3, pad 1) + BN + ReLU
// input plane x[f][t] = fbank[win*step + t][f] - mean[f] (the (B,T,F) -> (B,1,F,T) permute of the reference is
// just this indexing); one thread per output pixel, all C (<= 32) channels.
template <typename T>
__global__ __launch_bounds__(256) void emb_conv1_kernel(const float* __restrict__ fb, const int64_t* __restrict__ win,
                                                        const float* __restrict__ mean, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ out, int B, int F, int NT_,
                                                        int frames_per_step, int C) {
  __shared__ float sw[32 * 9 + 32];
  for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < C; i += 256) sw[32 * 9 + i] = bias[i];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)F * NT_;
  if (idx >= per * B) return;
  const int b = (int)(idx / per);
  const int rem = (int)(idx - (int64_t)b * per);
  const int f = rem / NT_, t = rem - f * NT_;
  const float* x = fb + (size_t)win[b] * frames_per_step * 80;
  const float* mu = mean + b * 80;
  float in[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ff = f + kh - 1, tt = t + kw - 1;
      in[kh * 3 + kw] = (ff >= 0 && ff < F && tt >= 0 && tt < NT_) ? x[(size_t)tt * 80 + ff] - mu[ff] : 0.f;
    }
  T* o = out + (((size_t)b * (F + 2) + f + 1) * (NT_ + 2) + t + 1) * C;
  constexpr int VE = 16 / (int)sizeof(T);
  for (int c0 = 0; c0 < C; c0 += VE) {
    T v[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float a = sw[32 * 9 + c0 + e];
#pragma unroll
      for (int k = 0; k < 9; ++k) a = fmaf(in[k], sw[(c0 + e) * 9 + k], a);
      v[e] = Cvt<T>::from_f32(fmaxf(a, 0.f));
    }
    *(uint4*)(o + c0) = *(const uint4*)v;
  }
}
int emb_conv1(hipStream_t s, int dtype, const float* fb, const int64_t* win, const float* mean, const float* w, const float* bias,
              void* out, int B, int F, int NT_, int frames_per_step, int C) {
  if (C > 32 || C % 8) { set_error("emb_conv1: stem channels must be a multiple of 8, <= 32"); return E_UNSUPPORTED; }
  if (B <= 0) return OK;
  const int64_t n = (int64_t)B * F * NT_;
  if (dtype == DT_BF16) hipLaunchKernelGGL(emb_conv1_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, s, fb, win, mean, w, bias, (bf16_t*)out, B, F, NT_, frames_per_step, C);
  else hipLaunchKernelGGL(emb_conv1_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, s, fb, win, mean, w, bias, (float*)out, B, F, NT_, frames_per_step, C);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ 3x3 / 1x1 convolution on MFMA
static constexpr int CV_TF = 4, CV_TT = 64, CV_MI = 4;

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cv_smem[];
  constexpr int CK = 64 / (int)sizeof(T);     // input channels per 64-byte chunk
  constexpr int VE = Mma16<T>::VE;
  constexpr int NJ = NT / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int s = p.stride, taps = p.taps;
  const int PF = s * (CV_TF - 1) + 3, PT = s * (CV_TT - 1) + 3;
  char* sP = cv_smem;
  char* sB = cv_smem + ((PF * PT * 64 + 127) & ~127);

  const int tiles_t = (p.To + CV_TT - 1) / CV_TT, tiles_f = (p.Fo + CV_TF - 1) / CV_TF, tiles_n = p.Cout / NT;
  int bid = blockIdx.x;
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tt = bid % tiles_t; bid /= tiles_t;
  const int tf = bid % tiles_f;
  const int b = bid / tiles_f;
  const int f0 = tf * CV_TF, t0 = tt * CV_TT, n0 = tn * NT;
  const int FiP = p.Fi + 2, TiP = p.Ti + 2;

  const T* __restrict__ in = (const T*)p.in;
  const T* __restrict__ w = (const T*)p.w;

  f32x4_t acc[CV_MI][NJ];
#pragma unroll
  for (int i = 0; i < CV_MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.Cin / CK;
  const int npvec = PF * PT * 4, nbvec = taps * NT * 4;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch) __syncthreads();
    for (int v = tid; v < npvec; v += 256) {
      const int px = v >> 2, piece = v & 3;
      const int pf = px / PT, pt = px - pf * PT;
      const int gf = f0 * s + pf, gt = t0 * s + pt;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (gf < FiP && gt < TiP) val = *(const uint4*)(in + (((size_t)b * FiP + gf) * TiP + gt) * p.Cin + ch * CK + piece * VE);
      *(uint4*)(sP + px * 64 + piece * 16) = val;
    }
    for (int v = tid; v < nbvec; v += 256) {
      const int row = v >> 2, piece = v & 3;
      const int tap = row / NT, n = row - tap * NT;
      *(uint4*)(sB + row * 64 + piece * 16) = *(const uint4*)(w + (((size_t)tap * nchunks + ch) * p.Cout + n0 + n) * CK + piece * VE);
    }
    __syncthreads();
    for (int tap = 0; tap < taps; ++tap) {
      const int kh = taps == 9 ? tap / 3 : 1, kw = taps == 9 ? tap - (tap / 3) * 3 : 1;
      uint4 bf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = *(const uint4*)(sB + (tap * NT + j * 16 + li) * 64 + lg * 16);
      const char* arow = sP + ((s * wave + kh) * PT + kw) * 64 + lg * 16;
#pragma unroll
      for (int mi = 0; mi < CV_MI; ++mi) {
        const uint4 a = *(const uint4*)(arow + (s * (mi * 16 + li)) * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j) Mma16<T>::run(a, bf[j], acc[mi][j]);
      }
    }
  }
  __syncthreads();

  // ---- epilogue: transpose each 16 x NT accumulator slab through LDS so a lane owns NT/4 consecutive channels
  // of one pixel: bias + residual + ReLU on 16-byte vectors, NHWC stores of full channel runs ----
  constexpr int SROW = NT * 4 + 16;
  constexpr int CW = NT / 4;                  // channels per lane
  char* slab = cv_smem + wave * (16 * SROW);
  const int crow = lg * 4;
  const int orow = lane >> 2, oseg = (lane & 3) * CW;
  const int f = f0 + wave;
  float bias_r[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) bias_r[e] = p.bias ? p.bias[n0 + oseg + e] : 0.f;
#pragma unroll
  for (int mi = 0; mi < CV_MI; ++mi) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) *(float*)(slab + (crow + r) * SROW + (j * 16 + li) * 4) = acc[mi][j][r];
    __builtin_amdgcn_wave_barrier();
    const int t = t0 + mi * 16 + orow;
    if (f >= p.Fo || t >= p.To) continue;
    const size_t pix = (((size_t)b * (p.Fo + 2) + f + 1) * (p.To + 2) + t + 1) * p.Cout + n0 + oseg;
    float v[CW];
#pragma unroll
    for (int q = 0; q < CW / 4; ++q) {
      const float4 x = *(const float4*)(slab + orow * SROW + (oseg + q * 4) * 4);
      v[q * 4 + 0] = x.x + bias_r[q * 4 + 0]; v[q * 4 + 1] = x.y + bias_r[q * 4 + 1];
      v[q * 4 + 2] = x.z + bias_r[q * 4 + 2]; v[q * 4 + 3] = x.w + bias_r[q * 4 + 3];
    }
    constexpr int OVE = 16 / (int)sizeof(T);
    if (p.res) {
      const T* rp = (const T*)p.res + pix;
#pragma unroll
      for (int q = 0; q < CW / OVE; ++q) {
        const uint4 raw = *(const uint4*)(rp + q * OVE);
        const T* re = (const T*)&raw;
#pragma unroll
        for (int e = 0; e < OVE; ++e) v[q * OVE + e] += Cvt<T>::to_f32(re[e]);
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < CW; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    T* op = (T*)p.out + pix;
#pragma unroll
    for (int q = 0; q < CW / OVE; ++q) {
      T o[OVE];
#pragma unroll
      for (int e = 0; e < OVE; ++e) o[e] = Cvt<T>::from_f32(v[q * OVE + e]);
      *(uint4*)(op + q * OVE) = *(const uint4*)o;
    }
  }
}

template <typename T, int NT>
static int launch_conv(hipStream_t st, const ConvArgs& p) {
  const int s = p.stride;
  const int PF = s * (CV_TF - 1) + 3, PT = s * (CV_TT - 1) + 3;
  size_t lds = (size_t)((PF * PT * 64 + 127) & ~127) + (size_t)p.taps * NT * 64;
  const size_t slab = (size_t)4 * 16 * (NT * 4 + 16);
  if (lds < slab) lds = slab;
  auto kern = conv_kernel<T, NT>;
  static size_t attr = 0;
  if (lds > attr) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  const int64_t blocks = (int64_t)p.B * cdiv(p.Fo, CV_TF) * cdiv(p.To, CV_TT) * (p.Cout / NT);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int conv2d(hipStream_t s, int dtype, const ConvArgs& p) {
  const int ck = dtype == DT_BF16 ? 32 : 16;
  if ((p.taps != 9 && p.taps != 1) || (p.stride != 1 && p.stride != 2) || p.Cin % ck || p.Cout % 32 ||
      p.Fo != (p.Fi - 1) / p.stride + 1 || p.To != (p.Ti - 1) / p.stride + 1) {
    set_error("conv2d: unsupported shape (3x3 pad 1 or 1x1, stride 1|2, channels multiples of 32)");
    return E_UNSUPPORTED;
  }
  if (p.B <= 0) return OK;
  const int nt = p.Cout % 128 == 0 ? 128 : (p.Cout % 64 == 0 ? 64 : 32);
  if (dtype == DT_BF16) {
    if (nt == 128) return launch_conv<bf16_t, 128>(s, p);
    if (nt == 64) return launch_conv<bf16_t, 64>(s, p);
    return launch_conv<bf16_t, 32>(s, p);
  }
  if (nt == 128) return launch_conv<float, 128>(s, p);
  if (nt == 64) return launch_conv<float, 64>(s, p);
  return launch_conv<float, 32>(s, p);
}

// ------------------------------------------------------------------------------------ masked statistics pooling (TSTP)
// pyannote StatsPool with frame weights: w = nearest-resampled mask; v1 = sum w + 1e-8; mean = sum(x w)/v1;
// var = sum(w (x-mean)^2) / (v1 - sum(w^2)/v1 + 1e-8); output [mean | std], feature index = channel*F + f
// (the reference flattens (C, F) channel-major before pooling).
template <typename T>
__global__ __launch_bounds__(256) void tstp_kernel(const T* __restrict__ x, const int* __restrict__ item_b,
                                                   const float* __restrict__ mask, int mask_len, int F, int TT, int C,
                                                   T* __restrict__ stats) {
  __shared__ float sw[256];
  __shared__ float sv[2];
  const int it = blockIdx.x;
  const int b = item_b[it];
  const float scale = (float)mask_len / (float)TT;
  for (int t = threadIdx.x; t < TT; t += 256) {
    int src = (int)floorf((float)t * scale);
    if (src > mask_len - 1) src = mask_len - 1;
    sw[t] = mask[(size_t)it * mask_len + src];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v1 = 0.f, v2 = 0.f;
    for (int t = 0; t < TT; ++t) { v1 += sw[t]; v2 += sw[t] * sw[t]; }
    v1 += 1e-8f;
    sv[0] = v1; sv[1] = v1 - v2 / v1 + 1e-8f;
  }
  __syncthreads();
  const float v1 = sv[0], den = sv[1];
  for (int c = threadIdx.x; c < C; c += 256) {
    for (int f = 0; f < F; ++f) {
      const T* xp = x + (((size_t)b * (F + 2) + f + 1) * (TT + 2) + 1) * C + c;
      float m = 0.f;
      for (int t = 0; t < TT; ++t) m = fmaf(Cvt<T>::to_f32(xp[(size_t)t * C]), sw[t], m);
      m /= v1;
      float q = 0.f;
      for (int t = 0; t < TT; ++t) { const float d = Cvt<T>::to_f32(xp[(size_t)t * C]) - m; q = fmaf(d * d, sw[t], q); }
      T* o = stats + (size_t)it * 2 * C * F;
      o[c * F + f] = Cvt<T>::from_f32(m);
      o[C * F + c * F + f] = Cvt<T>::from_f32(sqrtf(q / den));
    }
  }
}
int tstp_pool(hipStream_t s, int dtype, const void* x, const int* item_b, const float* mask, int mask_len, int n_items, int F,
              int TT, int C, void* stats) {
  if (TT > 256) { set_error("tstp_pool: more than 256 trunk frames per window"); return E_UNSUPPORTED; }
  if (n_items <= 0) return OK;
  if (dtype == DT_BF16) hipLaunchKernelGGL(tstp_kernel<bf16_t>, dim3(n_items), dim3(256), 0, s, (const bf16_t*)x, item_b, mask, mask_len, F, TT, C, (bf16_t*)stats);
  else hipLaunchKernelGGL(tstp_kernel<float>, dim3(n_items), dim3(256), 0, s, (const float*)x, item_b, mask, mask_len, F, TT, C, (float*)stats);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
