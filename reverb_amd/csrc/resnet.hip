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
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace rvb {

// ------------------------------------------------------------------------------------ per-window CMN statistics
// mean over the window's frames of each mel bin (pyannote WeSpeaker wrapper: features - features.mean(dim=1))
__global__ __launch_bounds__(256) void emb_mean_kernel(const float* __restrict__ fb, const int64_t* __restrict__ win,
                                                       int frames_per_step, int nfr, float* __restrict__ mean) {
  __shared__ float red[3][80];
  const int b = blockIdx.x;
  const int bin = threadIdx.x % 80, slot = threadIdx.x / 80;
  const float* x = fb + (size_t)(win ? win[b] : (int64_t)b) * frames_per_step * 80;
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
// just this indexing).  A block owns all F mel bins x 16 frames: the 18 fbank rows it needs are read as
// contiguous 320-byte rows into LDS (f fastest), the outputs are written t fastest, i.e. 16 pixels x C
// channels = 1 KiB contiguous runs of the NHWC plane.
static constexpr int ST_TT = 16, ST_F = 80;
template <typename T>
__global__ __launch_bounds__(256) void emb_conv1_kernel(const float* __restrict__ fb, const int64_t* __restrict__ win,
                                                        const float* __restrict__ mean, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ out, int B, int F, int NT_,
                                                        int frames_per_step, int C) {
  __shared__ float sw[32 * 9 + 32];
  __shared__ float sx[ST_TT + 2][ST_F + 2];
  for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < C; i += 256) sw[32 * 9 + i] = bias[i];
  const int tiles_t = (NT_ + ST_TT - 1) / ST_TT;
  const int b = blockIdx.x / tiles_t, t0 = (blockIdx.x - b * tiles_t) * ST_TT;
  const float* x = fb + (size_t)win[b] * frames_per_step * 80;
  const float* mu = mean + win[b] * 80;            // per-window CMN means of the whole file (computed at upload)
  for (int i = threadIdx.x; i < (ST_TT + 2) * (ST_F + 2); i += 256) {
    const int r = i / (ST_F + 2), cf = i - r * (ST_F + 2);
    const int tt = t0 + r - 1, ff = cf - 1;
    sx[r][cf] = (tt >= 0 && tt < NT_ && ff >= 0 && ff < F) ? x[(size_t)tt * 80 + ff] - mu[ff] : 0.f;
  }
  __syncthreads();
  // thread = (8-channel group cg, pixel lane): its 72 weights live in registers for all of its pixels, and the
  // four threads of a pixel write the four 16-byte pieces of its 64-byte channel run
  constexpr int CG = 8;
  const int ngrp = C / CG;                       // 1..4
  const int cg = threadIdx.x % ngrp, pl = threadIdx.x / ngrp;
  const int npl = 256 / ngrp;
  float wr[CG][9], br_[CG];
#pragma unroll
  for (int e = 0; e < CG; ++e) {
    br_[e] = sw[32 * 9 + cg * CG + e];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[e][k] = sw[(cg * CG + e) * 9 + k];
  }
  for (int pidx = pl; pidx < F * ST_TT; pidx += npl) {
    const int f = pidx / ST_TT, tl = pidx - f * ST_TT;
    const int t = t0 + tl;
    if (t >= NT_) continue;
    float in[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) in[kh * 3 + kw] = sx[tl + kw][f + kh];
    float v[CG];
#pragma unroll
    for (int e = 0; e < CG; ++e) {
      float a = br_[e];
#pragma unroll
      for (int k = 0; k < 9; ++k) a = fmaf(in[k], wr[e][k], a);
      v[e] = fmaxf(a, 0.f);
    }
    T* o = out + (((size_t)b * (F + 2) + f + 1) * (NT_ + 2) + t + 1) * C + cg * CG;
    if constexpr (sizeof(T) == 2) {
      uint4 pk;
      pk.x = pack2_bf16(v[0], v[1]);
      pk.y = pack2_bf16(v[2], v[3]);
      pk.z = pack2_bf16(v[4], v[5]);
      pk.w = pack2_bf16(v[6], v[7]);
      *(uint4*)o = pk;
    } else {
      ((float4*)o)[0] = make_float4(v[0], v[1], v[2], v[3]);
      ((float4*)o)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}
int emb_conv1(hipStream_t s, int dtype, const float* fb, const int64_t* win, const float* mean, const float* w, const float* bias,
              void* out, int B, int F, int NT_, int frames_per_step, int C) {
  if ((C != 8 && C != 16 && C != 32) || F > ST_F) { set_error("emb_conv1: stem channels must be 8, 16 or 32, and at most 80 mel bins"); return E_UNSUPPORTED; }
  if (B <= 0) return OK;
  const int blocks = B * cdiv(NT_, ST_TT);
  if (dtype == DT_BF16) hipLaunchKernelGGL(emb_conv1_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, fb, win, mean, w, bias, (bf16_t*)out, B, F, NT_, frames_per_step, C);
  else hipLaunchKernelGGL(emb_conv1_kernel<float>, dim3(blocks), dim3(256), 0, s, fb, win, mean, w, bias, (float*)out, B, F, NT_, frames_per_step, C);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ fp8 copies of activations (round 4 candidate)
// bf16 -> e4m3 at one scale per tensor (the A operand of conv_gemm.hip's fp8 kernel), and the running maximum of |x| that the
// calibration pass turns into that scale.  Whole bordered tensors: the zero border stays zero.  16 bytes in, 8 bytes out per thread.
__global__ __launch_bounds__(256) void act_quant_kernel(const bf16_t* __restrict__ in, uint8_t* __restrict__ out, size_t n8, float inv_scale,
                                                        unsigned* __restrict__ sat) {
  unsigned nclip = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 u = ((const uint4*)in)[i];
    const bf16_t* e = (const bf16_t*)&u;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = bf16_to_f32(e[k]) * inv_scale;
    if (sat) nclip += fp8_clipped(v[0], v[1], v[2], v[3]) + fp8_clipped(v[4], v[5], v[6], v[7]);
    ((uint2*)out)[i] = make_uint2(pack4_fp8(v[0], v[1], v[2], v[3]), pack4_fp8(v[4], v[5], v[6], v[7]));
  }
  if (sat && nclip) atomicAdd(sat, nclip);
}
__global__ __launch_bounds__(256) void act_amax_kernel(const bf16_t* __restrict__ in, size_t n8, unsigned* __restrict__ amax) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 u = ((const uint4*)in)[i];
    const bf16_t* e = (const bf16_t*)&u;
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(bf16_to_f32(e[k])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax, __float_as_uint(m));
}
int act_quant_fp8(hipStream_t s, const void* in_bf16, void* out_fp8, size_t n, float scale, unsigned* sat) {
  if (n == 0) return OK;
  if (n % 8 || !(scale > 0.f)) { set_error("act_quant_fp8: element count must be a multiple of 8 and the scale positive"); return E_ARG; }
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(act_quant_kernel, dim3((unsigned)std::min<size_t>((n8 + 255) / 256, 256 * 16)), dim3(256), 0, s, (const bf16_t*)in_bf16,
                     (uint8_t*)out_fp8, n8, 1.f / scale, sat);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}
int act_amax_bf16(hipStream_t s, const void* in_bf16, size_t n, unsigned* amax) {
  if (n == 0) return OK;
  if (n % 8) { set_error("act_amax_bf16: element count must be a multiple of 8"); return E_ARG; }
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(act_amax_kernel, dim3((unsigned)std::min<size_t>((n8 + 255) / 256, 256 * 16)), dim3(256), 0, s, (const bf16_t*)in_bf16, n8, amax);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------ 3x3 / 1x1 convolution on MFMA
static constexpr int CV_TF = 4, CV_TT = 64, CV_MI = 4;

template <typename T, int NT, int STRIDE, int TAPS>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cv_smem[];
  constexpr int CK = 64 / (int)sizeof(T);     // input channels per 64-byte chunk
  constexpr int VE = Mma16<T>::VE;
  constexpr int NJ = NT / 16;
  constexpr int s = STRIDE;
  constexpr int PF = s * (CV_TF - 1) + 3, PT = s * (CV_TT - 1) + 3;
  constexpr int NPV = PF * PT * 4;                 // 16-byte vectors of the input patch
  constexpr int PV = (NPV + 255) / 256;            // ... per thread
  constexpr int BV = (9 * NT * 4 + 255) / 256;     // 16-byte vectors of the 9 taps' weights per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int taps = TAPS;
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
  const int nchunks = p.Cin / CK;

  const T* __restrict__ in = (const T*)p.in + (size_t)b * FiP * TiP * p.Cin;
  const T* __restrict__ w = (const T*)p.w;

  // per-thread staging coordinates (the same for every channel chunk): global element offset, -1 = zero fill
  int poff[PV], boff[BV];
  const int nbvec = taps * NT * 4;
#pragma unroll
  for (int i = 0; i < PV; ++i) {
    const int v = tid + i * 256;
    const int px = v >> 2, piece = v & 3;
    const int pf = px / PT, pt = px - pf * PT;
    const int gf = f0 * s + pf, gt = t0 * s + pt;
    poff[i] = (v < NPV && gf < FiP && gt < TiP) ? (gf * TiP + gt) * p.Cin + piece * VE : -1;
  }
#pragma unroll
  for (int i = 0; i < BV; ++i) {
    const int v = tid + i * 256;
    const int row = v >> 2, piece = v & 3;
    const int tap = row / NT, n = row - tap * NT;
    boff[i] = v < nbvec ? (tap * nchunks * p.Cout + n0 + n) * CK + piece * VE : -1;
  }
  uint4 pr[PV], br[BV];
  auto load_regs = [&](int ch) {
#pragma unroll
    for (int i = 0; i < PV; ++i) pr[i] = poff[i] >= 0 ? *(const uint4*)(in + poff[i] + ch * CK) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < BV; ++i) br[i] = boff[i] >= 0 ? *(const uint4*)(w + boff[i] + (size_t)ch * p.Cout * CK) : make_uint4(0, 0, 0, 0);
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < PV; ++i) { const int v = tid + i * 256; if (v < NPV) *(uint4*)(sP + v * 16) = pr[i]; }
#pragma unroll
    for (int i = 0; i < BV; ++i) { const int v = tid + i * 256; if (v < nbvec) *(uint4*)(sB + v * 16) = br[i]; }
  };

  f32x4_t acc[CV_MI][NJ];
#pragma unroll
  for (int i = 0; i < CV_MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_regs(0);
  store_lds();
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch + 1 < nchunks) load_regs(ch + 1);      // next chunk's global loads fly under this chunk's MFMAs
    // taps fully unrolled with the NEXT tap's fragments read from LDS before the current tap's MFMAs are issued:
    // with one wave per SIMD nothing else hides the LDS latency
    uint4 bf[2][NJ], af[2][CV_MI];
    auto read_frags = [&](int tap, int buf) {
      const int kh = taps == 9 ? tap / 3 : 1, kw = taps == 9 ? tap - (tap / 3) * 3 : 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[buf][j] = *(const uint4*)(sB + (tap * NT + j * 16 + li) * 64 + lg * 16);
      const char* arow = sP + ((s * wave + kh) * PT + kw) * 64 + lg * 16;
#pragma unroll
      for (int mi = 0; mi < CV_MI; ++mi) af[buf][mi] = *(const uint4*)(arow + (s * (mi * 16 + li)) * 64);
    };
    read_frags(0, 0);
#pragma unroll
    for (int tap = 0; tap < taps; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < taps) read_frags(tap + 1, cur ^ 1);
#pragma unroll
      for (int mi = 0; mi < CV_MI; ++mi)
#pragma unroll
        for (int j = 0; j < NJ; ++j) Mma16<T>::run(af[cur][mi], bf[cur][j], acc[mi][j]);
    }
    if (ch + 1 < nchunks) {
      __syncthreads();
      store_lds();
      __syncthreads();
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
  // the residual vectors of all four slabs are requested before the first slab is transposed: inside the slab loop each
  // slab waited a full memory latency for its own residual before it could store (round 2, by elimination: the 32-channel
  // stage ran 74 -> 60 ms without the residual reads, 57 ms without the stores, 39 ms without both -- neither MFMAs nor
  // LDS fragment reads moved it).  bf16 only (a quarter of a 128-channel f32 tile would be 64 registers).
  constexpr int OVE0 = 16 / (int)sizeof(T);
  constexpr bool RPF = sizeof(T) == 2 && NT <= 64;
  uint4 rpre[RPF ? CV_MI : 1][RPF ? CW / OVE0 : 1];
  if constexpr (RPF) {
    if (p.res) {
#pragma unroll
      for (int mi = 0; mi < CV_MI; ++mi) {
        const int t = t0 + mi * 16 + orow;
        const bool ok = f < p.Fo && t < p.To;
        const size_t pix = (((size_t)b * (p.Fo + 2) + (ok ? f : 0) + 1) * (p.To + 2) + (ok ? t : 0) + 1) * p.Cout + n0 + oseg;
#pragma unroll
        for (int q = 0; q < CW / OVE0; ++q) rpre[mi][q] = *(const uint4*)((const T*)p.res + pix + q * OVE0);
      }
    }
  }
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
        uint4 raw;
        if constexpr (RPF) raw = rpre[mi][q]; else raw = *(const uint4*)(rp + q * OVE);
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

template <typename T, int NT, int STRIDE, int TAPS>
static int launch_conv_st(hipStream_t st, const ConvArgs& p) {
  constexpr int s = STRIDE;
  const int PF = s * (CV_TF - 1) + 3, PT = s * (CV_TT - 1) + 3;
  size_t lds = (size_t)((PF * PT * 64 + 127) & ~127) + (size_t)p.taps * NT * 64;
  const size_t slab = (size_t)4 * 16 * (NT * 4 + 16);
  if (lds < slab) lds = slab;
  auto kern = conv_kernel<T, NT, STRIDE, TAPS>;
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

template <typename T, int NT>
static int launch_conv(hipStream_t st, const ConvArgs& p) {
  if (p.taps == 1) return p.stride == 2 ? launch_conv_st<T, NT, 2, 1>(st, p) : launch_conv_st<T, NT, 1, 1>(st, p);
  return p.stride == 2 ? launch_conv_st<T, NT, 2, 9>(st, p) : launch_conv_st<T, NT, 1, 9>(st, p);
}

int conv2d(hipStream_t s, int dtype, const ConvArgs& p) {
  const int ck = dtype == DT_BF16 ? 32 : 16;
  if ((p.taps != 9 && p.taps != 1) || (p.stride != 1 && p.stride != 2) || p.Cin % ck || p.Cout % 32 ||
      p.Fo != (p.Fi - 1) / p.stride + 1 || p.To != (p.Ti - 1) / p.stride + 1) {
    set_error("conv2d: unsupported shape (3x3 pad 1 or 1x1, stride 1|2, channels multiples of 32)");
    return E_UNSUPPORTED;
  }
  if (p.B <= 0) return OK;
  if (conv_igemm_applicable(dtype, p)) return conv_igemm(s, p);     // only when the engine packed w_ig (RVD_CONV_IGEMM=1)
  if (conv_stream_applicable(dtype, p)) return conv_stream(s, p);
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

// ------------------------------------------------------------------------------------ fused BasicBlock, 32 channels (bf16)
// out = relu(conv_b(relu(conv_a(x) + ba)) + bb + x): the two 3x3 convolutions of a stride-1 residual block in one kernel.
// As two conv2d launches the 32-channel stage moves five tensor passes through HBM per block (x read, mid written, mid
// read, x read again as the residual, out written) for 18 kFLOP per pixel -- it runs at 4.5 TB/s, not on the MFMA pipe.
// Here x is read once and out written once; the intermediate tile (with its one-pixel halo recomputed) stays in LDS.
//
// A workgroup (4 waves) owns 2 mel rows of one window and walks their 62-frame tiles: output 2 x 62, intermediate 4 x 64,
// input patch 6 x 66 pixels.  conv_a: wave w computes intermediate row w (4 m-tiles x 9 taps x 2 channel halves = 72
// MFMAs), adds the bias, applies ReLU, zeroes what lies outside the image (conv_b's zero padding) and writes bf16 into LDS
// -- exactly the value the unfused path stores.  conv_b: wave w computes output row w >> 1, frames 32 (w & 1) .. +31
// (36 MFMAs), adds bias and the residual (the patch's centre, still in LDS), ReLU, and stores through a transposing slab.
// Both weight sets stay in LDS for all tiles of the row; the next tile's patch is in flight (registers) under the MFMAs.
// Accumulation order, rounding points and operand values are those of conv_kernel: results are bit-identical.
namespace {
constexpr int CP_OT = 62, CP_MT = 64, CP_PT = 66;     // frames per tile: output, intermediate, patch
constexpr int CP_OF = 2, CP_MF = 4, CP_PF = 6;        // mel rows
constexpr int CP_PATCH = CP_PF * CP_PT * 64;          // 25 344 B
constexpr int CP_MID = CP_MF * CP_MT * 64 + 128;      // + overrun of the last row's shifted reads
constexpr int CP_W = 9 * 32 * 64;                     // one convolution's weights
constexpr int CP_LDS = CP_PATCH + CP_MID + 2 * CP_W;  // 78 720 B: two workgroups per CU
}  // namespace

__global__ __launch_bounds__(256, 2) void conv_pair32_kernel(ConvPairArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cp_smem[];
  char* sP = cp_smem;
  char* sM = cp_smem + CP_PATCH;
  char* sWa = sM + CP_MID;
  char* sWb = sWa + CP_W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int F = p.F, T = p.T, FP = F + 2, TP = T + 2;
  const int tiles_f = (F + CP_OF - 1) / CP_OF, tiles_t = (T + CP_OT - 1) / CP_OT;
  const int b = blockIdx.x / tiles_f, tf = blockIdx.x - b * tiles_f;
  const int f0 = tf * CP_OF;
  const bf16_t* __restrict__ in = (const bf16_t*)p.in + (size_t)b * FP * TP * 32;
  bf16_t* __restrict__ out = (bf16_t*)p.out + (size_t)b * FP * TP * 32;

  // weights: 2 x 1152 16-byte vectors, staged once
  for (int v = tid; v < 2 * (CP_W / 16); v += 256) {
    const bool second = v >= CP_W / 16;
    const int vv = second ? v - CP_W / 16 : v;
    *(uint4*)((second ? sWb : sWa) + vv * 16) = *(const uint4*)((const char*)(second ? p.wb : p.wa) + vv * 16);
  }
  // patch staging coordinates of this thread (padded-plane row / column relative to the tile's patch origin)
  constexpr int NPV = CP_PF * CP_PT * 4, PV = (NPV + 255) / 256;
  int prow[PV], pcol[PV];
#pragma unroll
  for (int i = 0; i < PV; ++i) {
    const int v = tid + i * 256, px = v >> 2;
    prow[i] = v < NPV ? px / CP_PT : -1000;
    pcol[i] = px - (px / CP_PT) * CP_PT;
  }
  uint4 pr[PV];
  auto load_patch = [&](int tt) {
    const int pf0 = f0 - 1, pt0 = tt * CP_OT - 1;       // padded index of the patch origin (unpadded f0 - 2, t0 - 2)
#pragma unroll
    for (int i = 0; i < PV; ++i) {
      const int gf = pf0 + prow[i], gt = pt0 + pcol[i];
      const int piece = (tid + i * 256) & 3;
      pr[i] = (prow[i] >= 0 && gf >= 0 && gf < FP && gt >= 0 && gt < TP) ? *(const uint4*)(in + ((size_t)gf * TP + gt) * 32 + piece * 8)
                                                                        : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < PV; ++i) { const int v = tid + i * 256; if (v < NPV) *(uint4*)(sP + v * 16) = pr[i]; }
  };
  float ba[2], bb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { ba[j] = p.ba[j * 16 + li]; bb[j] = p.bb[j * 16 + li]; }

  load_patch(0);
  store_patch();
  __syncthreads();
  for (int tt = 0; tt < tiles_t; ++tt) {
    const int t0 = tt * CP_OT;
    if (tt + 1 < tiles_t) load_patch(tt + 1);            // in flight under this tile's MFMAs

    // ---- conv_a: intermediate row `wave`, 64 frames ----
    f32x4_t acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
      uint4 bf[2][2], af[2][4];
      auto read_frags = [&](int tap, int buf) {
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[buf][j] = *(const uint4*)(sWa + (tap * 32 + j * 16 + li) * 64 + lg * 16);
        const char* arow = sP + ((wave + kh) * CP_PT + kw) * 64 + lg * 16;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[buf][mi] = *(const uint4*)(arow + (mi * 16 + li) * 64);
      };
      read_frags(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1;
        if (tap + 1 < 9) read_frags(tap + 1, cur ^ 1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma16<bf16_t>::run(af[cur][mi], bf[cur][j], acc[mi][j]);
      }
    }
    {   // bias + ReLU + zero outside the image -> bf16 intermediate tile (C layout: pixel 4 lg + r of the m-tile, channel li + 16 j)
      const int fm = f0 - 1 + wave;
      const bool frow_ok = fm >= 0 && fm < F;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mc = mi * 16 + lg * 4 + r;
          const int tm = t0 - 1 + mc;
          const bool ok = frow_ok && tm >= 0 && tm < T;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float v = ok ? fmaxf(acc[mi][j][r] + ba[j], 0.f) : 0.f;
            *(bf16_t*)(sM + (wave * CP_MT + mc) * 64 + (j * 16 + li) * 2) = f32_to_bf16(v);
          }
        }
    }
    __syncthreads();

    // ---- conv_b: output row wave >> 1, frames 32 (wave & 1) .. +31 ----
    const int orow = wave >> 1, oc0 = (wave & 1) * 32;
    f32x4_t acb[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) acb[mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
      uint4 bf[2][2], af[2][2];
      auto read_frags = [&](int tap, int buf) {
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[buf][j] = *(const uint4*)(sWb + (tap * 32 + j * 16 + li) * 64 + lg * 16);
        const char* arow = sM + ((orow + kh) * CP_MT + oc0 + kw) * 64 + lg * 16;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[buf][mi] = *(const uint4*)(arow + (mi * 16 + li) * 64);
      };
      read_frags(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1;
        if (tap + 1 < 9) read_frags(tap + 1, cur ^ 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma16<bf16_t>::run(af[cur][mi], bf[cur][j], acb[mi][j]);
      }
    }
    // bias + residual (patch centre: row orow + 2, frame o + 2) + ReLU, still in the C layout
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = oc0 + mi * 16 + lg * 4 + r;
        const char* xr = sP + ((orow + 2) * CP_PT + o + 2) * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float res = bf16_to_f32(*(const bf16_t*)(xr + (j * 16 + li) * 2));
          acb[mi][j][r] = fmaxf(acb[mi][j][r] + bb[j] + res, 0.f);
        }
      }
    __syncthreads();                                     // every wave is done with the patch and the intermediate tile

    // ---- store: 16-pixel slabs transposed through LDS (over the patch) -> one 16-byte vector of 8 channels per lane ----
    {
      constexpr int SROW = 32 * 4 + 16;
      char* slab = sP + wave * (16 * SROW);
      const int spx = lane >> 2, sch = (lane & 3) * 8;
      const int f = f0 + orow;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) *(float*)(slab + (lg * 4 + r) * SROW + (j * 16 + li) * 4) = acb[mi][j][r];
        __builtin_amdgcn_wave_barrier();
        const float4 x0 = *(const float4*)(slab + spx * SROW + sch * 4);
        const float4 x1 = *(const float4*)(slab + spx * SROW + sch * 4 + 16);
        const int o = oc0 + mi * 16 + spx, t = t0 + o;
        if (o < CP_OT && t < T && f < F)
          *(uint4*)(out + ((size_t)(f + 1) * TP + t + 1) * 32 + sch) =
              make_uint4(pack2_bf16(x0.x, x0.y), pack2_bf16(x0.z, x0.w), pack2_bf16(x1.x, x1.y), pack2_bf16(x1.z, x1.w));
      }
    }
    if (tt + 1 < tiles_t) {
      __syncthreads();                                   // slabs read
      store_patch();
      __syncthreads();
    }
  }
}

bool conv_pair32_applicable(int dtype, int cin, int cmid, int cout, int stride_a, int stride_b, int taps_a, int taps_b) {
  // opt-in (RVD_CONV_FUSE=1): measured 73.5 ms for the 32-channel stage against 67 ms for two conv2d launches once their
  // residual reads were taken off the slab loop -- two HBM passes instead of five, but five barriers per 124 output pixels
  const char* e = getenv("RVD_CONV_FUSE");          // read per call: the tests switch it between engines
  const bool off = !(e && atoi(e) == 1);
  return !off && dtype == DT_BF16 && cin == 32 && cmid == 32 && cout == 32 && stride_a == 1 && stride_b == 1 && taps_a == 9 && taps_b == 9;
}

int conv_pair32(hipStream_t s, const ConvPairArgs& a) {
  if (a.B <= 0) return OK;
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pair32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS));
    attr_set = true;
  }
  const int blocks = a.B * ((a.F + CP_OF - 1) / CP_OF);
  hipLaunchKernelGGL(conv_pair32_kernel, dim3(blocks), dim3(256), CP_LDS, s, a);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
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
  __shared__ float sv[3];
  const int it = blockIdx.x, f = blockIdx.y;
  const int b = item_b[it];
  const float scale = (float)mask_len / (float)TT;
  for (int t = threadIdx.x; t < TT; t += 256) {
    int src = (int)floorf((float)t * scale);
    if (src > mask_len - 1) src = mask_len - 1;
    sw[t] = mask[(size_t)it * mask_len + src];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < TT; ++t) { s1 += sw[t]; s2 += sw[t] * sw[t]; }
    const float v1 = s1 + 1e-8f;
    sv[0] = v1; sv[1] = v1 - s2 / v1 + 1e-8f; sv[2] = s1;
  }
  __syncthreads();
  const float v1 = sv[0], den = sv[1], s1 = sv[2];
  for (int c = threadIdx.x; c < C; c += 256) {
    const T* xp = x + (((size_t)b * (F + 2) + f + 1) * (TT + 2) + 1) * C + c;
    double a1 = 0.0, a2 = 0.0;          // sum w x, sum w x^2 (one pass; fp64 keeps the variance exact enough)
#pragma unroll 5
    for (int t = 0; t < TT; ++t) {
      const float v = Cvt<T>::to_f32(xp[(size_t)t * C]);
      const double wv = (double)sw[t] * v;
      a1 += wv; a2 += wv * v;
    }
    const double m = a1 / (double)v1;
    double q = a2 - 2.0 * m * a1 + m * m * (double)s1;      // = sum w (x - m)^2
    if (q < 0.0) q = 0.0;
    T* o = stats + (size_t)it * 2 * C * F;
    o[c * F + f] = Cvt<T>::from_f32((float)m);
    o[C * F + c * F + f] = Cvt<T>::from_f32(sqrtf((float)(q / (double)den)));
  }
}
int tstp_pool(hipStream_t s, int dtype, const void* x, const int* item_b, const float* mask, int mask_len, int n_items, int F,
              int TT, int C, void* stats) {
  if (TT > 256) { set_error("tstp_pool: more than 256 trunk frames per window"); return E_UNSUPPORTED; }
  if (n_items <= 0) return OK;
  const dim3 grid(n_items, F);
  if (dtype == DT_BF16) hipLaunchKernelGGL(tstp_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, item_b, mask, mask_len, F, TT, C, (bf16_t*)stats);
  else hipLaunchKernelGGL(tstp_kernel<float>, grid, dim3(256), 0, s, (const float*)x, item_b, mask, mask_len, F, TT, C, (float*)stats);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
