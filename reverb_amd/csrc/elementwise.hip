// HBM-bound row / elementwise kernels of the Reverb-ASR hot path (gfx950).
// Reference ops replaced (paths relative to asr/wenet/):
//   subsample_conv1   transformer/cmvn.py:36-47 + subsampling.py:187-188 (Conv2d(1,d,3,2)+ReLU)
//   rownorm           every nn.LayerNorm on the path (encoder_layer.py:149-159, encoder.py:107,
//                     decoder_layer.py:56-58,241-243, decoder.py:90) and the conv-module norm
//                     (+SiLU, convolution.py:133-137; BatchNorm1d folded to a per-channel affine)
//   glu_dwconv        convolution.py:107-131 (mask -> [pw-conv1 by GEMM] -> GLU -> depthwise conv)
//   embed_tokens      decoder.py:82-87,157 + embedding.py:73-76
//   logsoftmax_topk   ctc.py:106-114, asr_model.py:318-329, search.py:111,155
//   lse_gather        asr_model.py:969 + search.py:417-437 (only the needed log-probs)
#include "common.h"
#include "kernels.h"

namespace rvb {

// ------------------------------------------------------------------------------------------------
// CMVN + conv1 (1 -> d channels, 3x3, stride 2) + ReLU, NHWC output
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ feats, const float* __restrict__ mean,
                                                    const float* __restrict__ istd, const float* __restrict__ w,
                                                    const float* __restrict__ bias, T* __restrict__ out, int T0,
                                                    int F0, int T1, int F1, int d) {
  extern __shared__ float s_in[];  // [3][F0]
  const int t1 = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * F0; i += 256) {
    const int kh = i / F0, f = i - kh * F0;
    s_in[i] = (feats[((size_t)b * T0 + 2 * t1 + kh) * F0 + f] - mean[f]) * istd[f];
  }
  __syncthreads();
  const int ncg = d >> 2;                       // channel groups of 4
  const int per = ncg < 256 ? ncg : 256;        // groups handled per pass
  const int nslots = 256 / per;                 // f1 slots sharing the block
  const int cgl = threadIdx.x % per, fslot = threadIdx.x / per;
  if (fslot >= nslots) return;
  T* orow = out + ((size_t)b * T1 + t1) * F1 * d;
  for (int cg = cgl; cg < ncg; cg += per) {
    float wr[4][9], br[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      br[c] = bias[cg * 4 + c];
#pragma unroll
      for (int k = 0; k < 9; ++k) wr[c][k] = w[(cg * 4 + c) * 9 + k];
    }
    for (int f1 = fslot; f1 < F1; f1 += nslots) {
      float xin[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xin[kh * 3 + kw] = s_in[kh * F0 + 2 * f1 + kw];
      T o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc = br[c];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += wr[c][k] * xin[k];
        o[c] = Cvt<T>::from_f32(fmaxf(acc, 0.f));
      }
      T* dst = orow + (size_t)f1 * d + cg * 4;
      if constexpr (sizeof(T) == 2) {
        uint2 pk;
        pk.x = (uint32_t)(*(uint16_t*)&o[0]) | ((uint32_t)(*(uint16_t*)&o[1]) << 16);
        pk.y = (uint32_t)(*(uint16_t*)&o[2]) | ((uint32_t)(*(uint16_t*)&o[3]) << 16);
        *(uint2*)dst = pk;
      } else {
        *(float4*)dst = make_float4(*(float*)&o[0], *(float*)&o[1], *(float*)&o[2], *(float*)&o[3]);
      }
    }
  }
}

int subsample_conv1(hipStream_t s, int dtype, const float* feats, const float* mean, const float* istd,
                    const float* w, const float* b, void* out, int B, int T0, int F0, int d) {
  const int T1 = (T0 - 3) / 2 + 1, F1 = (F0 - 3) / 2 + 1;
  if (B <= 0 || T1 <= 0) return OK;
  if (d % 4) { set_error("subsample_conv1: d must be a multiple of 4"); return E_ARG; }
  dim3 grid(T1, B);
  const size_t sh = 3 * F0 * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(conv1_kernel<bf16_t>, grid, dim3(256), sh, s, feats, mean, istd, w, b, (bf16_t*)out, T0, F0, T1, F1, d);
  else
    hipLaunchKernelGGL(conv1_kernel<float>, grid, dim3(256), sh, s, feats, mean, istd, w, b, (float*)out, T0, F0, T1, F1, d);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// Row normalisation: one wave64 per row, shuffle reductions, fp32 statistics
// ------------------------------------------------------------------------------------------------
template <typename OutT, typename AddT>
__global__ __launch_bounds__(256) void rownorm_kernel(NormArgs a) {
  constexpr int NV = 8;  // float4 per lane -> d <= 2048
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int d = a.d;
  const float* x = a.x + (size_t)row * d;
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = c < d ? *(const float4*)(x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  float mean = 0.f, rstd = 1.f;
  if (a.mode == NORM_LN) {
    mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        sq += dx * dx + dy * dy + dz * dz + dw * dw;
      }
    }
    rstd = rsqrtf(wave_sum(sq) / (float)d + a.eps);
  }
  OutT* out = (OutT*)a.out + (size_t)row * d;
  const AddT* add = a.add ? (const AddT*)a.add + (size_t)row * d : nullptr;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c >= d) continue;
    const float4 g = *(const float4*)(a.gamma + c);
    const float4 be = *(const float4*)(a.beta + c);
    float o[4] = {(v[i].x - mean) * rstd * g.x + be.x, (v[i].y - mean) * rstd * g.y + be.y,
                  (v[i].z - mean) * rstd * g.z + be.z, (v[i].w - mean) * rstd * g.w + be.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a.silu) o[e] = o[e] / (1.0f + expf(-o[e]));
      if (add) o[e] += Cvt<AddT>::to_f32(add[c + e]);
    }
    if constexpr (sizeof(OutT) == 2) {
      uint2 pk;
      pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
      *(uint2*)(out + c) = pk;
    } else {
      *(float4*)(out + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

int rownorm(hipStream_t s, int dtype, const NormArgs& a) {
  if (a.M <= 0) return OK;
  if (a.d % 4 || a.d > 2048) { set_error("rownorm: d must be a multiple of 4 and <= 2048"); return E_ARG; }
  dim3 grid(cdiv(a.M, 4));
  if (dtype == DT_BF16) {
    if (a.out_f32) hipLaunchKernelGGL((rownorm_kernel<float, bf16_t>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rownorm_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((rownorm_kernel<float, float>), grid, dim3(256), 0, s, a);
  }
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// GLU + depthwise conv along time (per chunk), LDS halo tile of 64 time steps x 64 channels
// ------------------------------------------------------------------------------------------------
static constexpr int DW_TT = 64, DW_CT = 64, DW_KMAX = 63;

template <typename T>
__global__ __launch_bounds__(256) void glu_dw_kernel(GluDwArgs a) {
  __shared__ float s_g[DW_TT + DW_KMAX - 1][DW_CT];
  __shared__ float s_w[DW_KMAX][DW_CT];
  const int c = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int t0 = blockIdx.x * DW_TT, c0 = blockIdx.y * DW_CT, b = blockIdx.z;
  const int ch = c0 + c;
  const bool cok = ch < a.d;
  const int K = a.K, pad = (K - 1) / 2;
  const int len = a.lens[b];
  const T* G = (const T*)a.G;
  float gpad = 0.f;   // GLU of the pointwise-conv1 bias: what a zero-masked (padded) frame produces
  if (cok) {
    const float pa = a.pw1_bias[ch], pb = a.pw1_bias[a.d + ch];
    gpad = pa / (1.0f + expf(-pb));
  }
  for (int k = slot; k < K; k += 4) s_w[k][c] = cok ? a.dw_w[(size_t)ch * K + k] : 0.f;
  const int rows = DW_TT + K - 1;
  for (int r = slot; r < rows; r += 4) {
    const int t = t0 - pad + r;
    float g = 0.f;
    if (cok && t >= 0 && t < a.T) {
      if (t < len) {
        const T* gr = G + ((size_t)b * a.T + t) * 2 * a.d;
        const float ga = Cvt<T>::to_f32(gr[ch]), gb = Cvt<T>::to_f32(gr[a.d + ch]);
        g = ga / (1.0f + expf(-gb));
      } else {
        g = gpad;
      }
    }
    s_g[r][c] = g;
  }
  __syncthreads();
  if (!cok) return;
  float acc[DW_TT / 4];
  const float bv = a.dw_b[ch];
#pragma unroll
  for (int i = 0; i < DW_TT / 4; ++i) acc[i] = bv;
  for (int k = 0; k < K; ++k) {
    const float wk = s_w[k][c];
#pragma unroll
    for (int i = 0; i < DW_TT / 4; ++i) acc[i] += wk * s_g[slot + 4 * i + k][c];
  }
#pragma unroll
  for (int i = 0; i < DW_TT / 4; ++i) {
    const int t = t0 + slot + 4 * i;
    if (t < a.T) a.out[((size_t)b * a.T + t) * a.d + ch] = acc[i];
  }
}

int glu_dwconv(hipStream_t s, int dtype, const GluDwArgs& a) {
  if (a.B <= 0 || a.T <= 0) return OK;
  if (a.K > DW_KMAX || (a.K % 2) == 0) { set_error("glu_dwconv: kernel must be odd and <= 63"); return E_ARG; }
  dim3 grid(cdiv(a.T, DW_TT), cdiv(a.d, DW_CT), a.B);
  if (dtype == DT_BF16) hipLaunchKernelGGL(glu_dw_kernel<bf16_t>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(glu_dw_kernel<float>, grid, dim3(256), 0, s, a);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ E, const float* __restrict__ pe,
                                                    const int* __restrict__ tok, const int* __restrict__ pos,
                                                    float* __restrict__ out, int rows, int d, float scale) {
  const int row = blockIdx.x;
  const float* e = E + (size_t)tok[row] * d;
  const float* p = pe + (size_t)pos[row] * d;
  for (int c = threadIdx.x; c < d; c += 256) out[(size_t)row * d + c] = e[c] * scale + p[c];
}

int embed_tokens(hipStream_t s, const float* E, const float* pe, const int* tok, const int* pos, float* out,
                 int rows, int d, float scale) {
  if (rows <= 0) return OK;
  hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(256), 0, s, E, pe, tok, pos, out, rows, d, scale);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// log-softmax + top-k per row: one wave per row.  Each lane keeps a sorted top-16 of its strided
// slice in registers (static indices only), then the wave pops the global maximum k times.
// ------------------------------------------------------------------------------------------------
static constexpr int TOPK_MAX = 16;

__device__ inline float row_logit(const float* x, int i, float pen, int blank) {
  float v = x[i];
  if (i == blank) v -= pen;
  return v;
}

__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const float* __restrict__ logits, int M, int V, int ld,
                                                              int k, float pen, int blank, float* __restrict__ tv,
                                                              int* __restrict__ ti, float* __restrict__ lp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* x = logits + (size_t)row * ld;
  float bv[TOPK_MAX];
  int bi[TOPK_MAX];
#pragma unroll
  for (int i = 0; i < TOPK_MAX; ++i) { bv[i] = -INFINITY; bi[i] = 0x7fffffff; }
  float mx = -INFINITY;
  for (int i = lane; i < V; i += 64) {
    float v = row_logit(x, i, pen, blank);
    mx = fmaxf(mx, v);
    if (v > bv[TOPK_MAX - 1]) {
      int vi = i;
#pragma unroll
      for (int j = 0; j < TOPK_MAX; ++j) {
        if (v > bv[j]) {   // strict: on ties the earlier (lower) index stays ahead
          const float tvv = bv[j]; const int tii = bi[j];
          bv[j] = v; bi[j] = vi; v = tvv; vi = tii;
        }
      }
    }
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int i = lane; i < V; i += 64) se += expf(row_logit(x, i, pen, blank) - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  if (lp) {
    float* o = lp + (size_t)row * V;
    for (int i = lane; i < V; i += 64) o[i] = row_logit(x, i, pen, blank) - lse;
  }
  for (int r = 0; r < k; ++r) {
    float hv = bv[0];
    int hi = bi[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(hv, o, 64);
      const int oi = __shfl_xor(hi, o, 64);
      if (ov > hv || (ov == hv && oi < hi)) { hv = ov; hi = oi; }
    }
    if (bi[0] == hi && bv[0] == hv) {   // the unique winner pops its head
#pragma unroll
      for (int j = 0; j < TOPK_MAX - 1; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
      bv[TOPK_MAX - 1] = -INFINITY; bi[TOPK_MAX - 1] = 0x7fffffff;
    }
    if (lane == 0) {
      tv[(size_t)row * k + r] = hv - lse;
      ti[(size_t)row * k + r] = hi;
    }
  }
}

int logsoftmax_topk(hipStream_t s, const float* logits, int M, int V, int ld, int k, float blank_penalty,
                    int blank_id, float* topk_val, int* topk_idx, float* logp_out) {
  if (M <= 0) return OK;
  if (k < 1 || k > TOPK_MAX || k > V) { set_error("logsoftmax_topk: beam must be in [1,16] and <= vocab"); return E_ARG; }
  hipLaunchKernelGGL(logsoftmax_topk_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, logits, M, V, ld, k,
                     blank_penalty, blank_id, topk_val, topk_idx, logp_out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

__global__ __launch_bounds__(256) void lse_gather_kernel(const float* __restrict__ logits, int R, int V, int ld,
                                                         const int* __restrict__ target, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* x = logits + (size_t)row * ld;
  float mx = -INFINITY;
  for (int i = lane; i < V; i += 64) mx = fmaxf(mx, x[i]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int i = lane; i < V; i += 64) se += expf(x[i] - mx);
  se = wave_sum(se);
  if (lane == 0) out[row] = x[target[row]] - mx - logf(se);
}

int lse_gather(hipStream_t s, const float* logits, int R, int V, int ld, const int* target, float* out) {
  if (R <= 0) return OK;
  hipLaunchKernelGGL(lse_gather_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, logits, R, V, ld, target, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dst[i] = Cvt<T>::from_f32(src[i]);
}

int convert_f32(hipStream_t s, int dtype, const float* src, void* dst, size_t n) {
  if (n == 0) return OK;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == DT_BF16) hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, n);
  else hipLaunchKernelGGL(convert_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, n);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
