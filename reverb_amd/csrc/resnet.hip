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
  if (conv_row64_applicable(dtype, p)) return conv_row64(s, p);
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
