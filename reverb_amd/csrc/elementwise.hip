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
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace rvb {

// ------------------------------------------------------------------------------------------------
// CMVN + conv1 (1 -> d channels, 3x3, stride 2) + ReLU, NHWC output
// ------------------------------------------------------------------------------------------------
static constexpr int CONV1_ROWS = 4;      // output time rows per workgroup: the 72 weights of a thread are loaded once for all of them

// T = fp8_t (round 4, RVB_FP8 with conv2 in the policy): the output is value / scale in e4m3 (ReLU output: >= 0), 8 bytes per
// thread and pixel, values beyond 448 clipped and counted in *sat.  amax (calibration pass, bf16 output): the running maximum
// of the output as float bits -- the values are non-negative, so an unsigned atomicMax orders them.
template <typename T>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ feats, const float* __restrict__ mean,
                                                    const float* __restrict__ istd, const float* __restrict__ w,
                                                    const float* __restrict__ bias, T* __restrict__ out, int T0,
                                                    int F0, int T1, int F1, int d, float inv_scale, unsigned* __restrict__ amax,
                                                    unsigned* __restrict__ sat) {
  extern __shared__ float s_in[];  // [2 * CONV1_ROWS + 1][F0]: the input rows of CONV1_ROWS output rows (stride 2, 3 taps)
  const int t1_0 = blockIdx.x * CONV1_ROWS, b = blockIdx.y;
  const int nrow = min(CONV1_ROWS, T1 - t1_0);
  for (int i = threadIdx.x; i < (2 * nrow + 1) * F0; i += 256) {
    const int kh = i / F0, f = i - kh * F0;
    s_in[i] = (feats[((size_t)b * T0 + 2 * t1_0 + kh) * F0 + f] - mean[f]) * istd[f];
  }
  __syncthreads();
  // a thread owns 8 adjacent output channels (one 16-byte bf16 store, two for f32): full-width stores -- 8-byte
  // stores reached 3.4 TB/s on the 14.4 GB this kernel writes per hour of audio
  const int ncg = d >> 3;                       // channel groups of 8
  const int per = ncg < 256 ? ncg : 256;        // groups handled per pass
  const int nslots = 256 / per;                 // f1 slots sharing the block
  const int cgl = threadIdx.x % per, fslot = threadIdx.x / per;
  if (fslot >= nslots) return;
  float vmax = 0.f;
  unsigned nclip = 0;
  for (int cg = cgl; cg < ncg; cg += per) {
    float wr[8][9], br[8];
    // weights tap-major [9][d]: the 8 channels of a thread are 32 contiguous bytes, the threads of a wave contiguous
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 lo = *(const float4*)(w + (size_t)k * d + cg * 8), hi = *(const float4*)(w + (size_t)k * d + cg * 8 + 4);
      wr[0][k] = lo.x; wr[1][k] = lo.y; wr[2][k] = lo.z; wr[3][k] = lo.w;
      wr[4][k] = hi.x; wr[5][k] = hi.y; wr[6][k] = hi.z; wr[7][k] = hi.w;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) br[c] = bias[cg * 8 + c];
    for (int row = 0; row < nrow; ++row) {
      T* orow = out + ((size_t)b * T1 + t1_0 + row) * F1 * d;
      const float* in = s_in + 2 * row * F0;
      for (int f1 = fslot; f1 < F1; f1 += nslots) {
        float xin[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) xin[kh * 3 + kw] = in[kh * F0 + 2 * f1 + kw];
        float o[8];
        if constexpr (sizeof(T) == 4) {
          // f32 engine: multiply and add rounded separately, as since round 1 (its token-exact goldens were produced with these)
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float acc = br[c];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += wr[c][k] * xin[k];
            o[c] = fmaxf(acc, 0.f);
          }
        } else {
          // bf16 / fp8 engines: two channels per v_pk_fma_f32 (36 packed FMAs per 8 channels; the separate form compiled to 36
          // v_pk_mul_f32 + 64 v_add_f32 and the kernel sat between its VALU work and its 14.4 GB of stores)
          typedef float c1_f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int c = 0; c < 8; c += 2) {
            c1_f32x2 acc = {br[c], br[c + 1]};
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = __builtin_elementwise_fma((c1_f32x2){wr[c][k], wr[c + 1][k]}, (c1_f32x2){xin[k], xin[k]}, acc);
            o[c] = fmaxf(acc.x, 0.f); o[c + 1] = fmaxf(acc.y, 0.f);
          }
        }
        T* dst = orow + (size_t)f1 * d + cg * 8;
        if (amax) {
#pragma unroll
          for (int c = 0; c < 8; ++c) vmax = fmaxf(vmax, o[c]);
        }
        if constexpr (sizeof(T) == 1) {
#pragma unroll
          for (int c = 0; c < 8; ++c) o[c] *= inv_scale;
          if (sat) nclip += fp8_clipped(o[0], o[1], o[2], o[3]) + fp8_clipped(o[4], o[5], o[6], o[7]);
          *(uint2*)dst = make_uint2(pack4_fp8(o[0], o[1], o[2], o[3]), pack4_fp8(o[4], o[5], o[6], o[7]));
        } else if constexpr (sizeof(T) == 2) {
          *(uint4*)dst = make_uint4(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7]));
        } else {
          ((float4*)dst)[0] = make_float4(o[0], o[1], o[2], o[3]);
          ((float4*)dst)[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    }
  }
  if (amax) {
    vmax = wave_max(vmax);
    if ((threadIdx.x & 63) == 0 && vmax > 0.f) atomicMax(amax, __float_as_uint(vmax));
  }
  if (sat && nclip) atomicAdd(sat, nclip);
}

int subsample_conv1(hipStream_t s, int dtype, const float* feats, const float* mean, const float* istd,
                    const float* w, const float* b, void* out, int B, int T0, int F0, int d, float out_fp8_scale, unsigned* amax,
                    unsigned* sat) {
  const int T1 = (T0 - 3) / 2 + 1, F1 = (F0 - 3) / 2 + 1;
  if (B <= 0 || T1 <= 0) return OK;
  if (d % 8) { set_error("subsample_conv1: d must be a multiple of 8"); return E_ARG; }
  dim3 grid(cdiv(T1, CONV1_ROWS), B);
  const size_t sh = (2 * CONV1_ROWS + 1) * F0 * sizeof(float);
  if (out_fp8_scale > 0.f) {
    if (dtype != DT_BF16) { set_error("subsample_conv1: fp8 output belongs to the bf16 engine"); return E_ARG; }
    hipLaunchKernelGGL(conv1_kernel<fp8_t>, grid, dim3(256), sh, s, feats, mean, istd, w, b, (fp8_t*)out, T0, F0, T1, F1, d,
                       1.f / out_fp8_scale, (unsigned*)nullptr, sat);
  } else if (dtype == DT_BF16)
    hipLaunchKernelGGL(conv1_kernel<bf16_t>, grid, dim3(256), sh, s, feats, mean, istd, w, b, (bf16_t*)out, T0, F0, T1, F1, d, 1.f, amax,
                       (unsigned*)nullptr);
  else
    hipLaunchKernelGGL(conv1_kernel<float>, grid, dim3(256), sh, s, feats, mean, istd, w, b, (float*)out, T0, F0, T1, F1, d, 1.f, amax,
                       (unsigned*)nullptr);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// Row normalisation: a wave64 walks rows (grid-stride), gamma / beta live in registers, the next row's loads are in
// flight while the current one is reduced and stored; shuffle reductions, fp32 statistics.  Optional second stage:
// out2 = LN2(result) in the compute dtype -- the encoder's `norm_final` is always followed by the next block's first
// LayerNorm (or `after_norm`), so the pair is one pass over the row (encoder_layer.py:242-244 -> :199-201).
// ------------------------------------------------------------------------------------------------
// SW >= 0: the mode switches are compile-time (bit 0 bf16 input, bit 1 LayerNorm statistics, bit 2 SiLU, bit 3 `add`, bit 4 fp8
// second-stage output), so the row loop has no branches and the compiler's waitcnt pass emits COUNTED waits: with the run-time
// switches (SW = -1) it has to assume that any conditional load may be pending at every join and answers with `s_waitcnt vmcnt(0)`
// right behind the next row's prefetch and in front of every store -- the loop then runs one memory round trip at a time.
template <typename OutT, typename AddT, int NV, bool TWO, int SW>
__global__ __launch_bounds__(256, (NV <= 4 ? 4 : 2)) void rownorm_kernel(NormArgs a) {
  constexpr bool CT = SW >= 0;
  // the bf16 engine's statistics use the LDS-free reduction (its summation tree differs in the last bit from __shfl_xor's, which the
  // f32 engine keeps so that its results do not move)
  auto wsum = [](float x) __attribute__((always_inline)) { if constexpr (sizeof(AddT) == 2) return wave_sum_dpp(x); else return wave_sum(x); };
  const bool sw_xb = CT ? (bool)(SW & 1) : (bool)a.x_bf16;
  const bool sw_ln = CT ? (bool)(SW & 2) : a.mode == NORM_LN;
  const bool sw_silu = CT ? (bool)(SW & 4) : (bool)a.silu;
  const bool sw_add = CT ? (bool)(SW & 8) : a.add != nullptr;
  const bool sw_o2f8 = CT ? (bool)(SW & 16) : (bool)a.out2_fp8;
  // a lane owns NV/2 runs of 8 consecutive columns (two float4 loads, one 16-byte bf16 / 8-byte fp8 store per run):
  // vector i covers columns COL(i) .. COL(i)+3
  const int lane = threadIdx.x & 63;
  const int d = a.d;
  const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
#define COL(i) ((((lane) + 64 * ((i) >> 1)) << 3) + (((i) & 1) << 2))
  float4 g[NV], be[NV], g2[TWO ? NV : 1], be2[TWO ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = COL(i);
    const bool ok = c < d;
    g[i] = ok ? *(const float4*)(a.gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    be[i] = ok ? *(const float4*)(a.beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (TWO) {
      g2[i] = ok ? *(const float4*)(a.gamma2 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      be2[i] = ok ? *(const float4*)(a.beta2 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // Loads are UNCONDITIONAL (row and column clamped into the matrix, the value masked afterwards): a predicated load is a branch
  // around the instruction, and behind a branch the waitcnt pass no longer knows how many loads are in flight -- it then waits
  // for all of them (`vmcnt(0)`) at the first use, i.e. right behind the prefetch.
  auto load_row = [&](int row, float4* v) {
    const int rr = min(row, a.M - 1);
    const bool rok = row < a.M;
    if (sw_xb) {      // the depthwise convolution's output of the bf16 engine
      const bf16_t* x = (const bf16_t*)a.x + (size_t)rr * d;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = COL(i);
        const uint2 u = *(const uint2*)(x + min(c, d - 4));
        const bool ok = rok && c < d;
        v[i] = make_float4(ok ? bf16_to_f32((bf16_t)(u.x & 0xffffu)) : 0.f, ok ? bf16_to_f32((bf16_t)(u.x >> 16)) : 0.f,
                           ok ? bf16_to_f32((bf16_t)(u.y & 0xffffu)) : 0.f, ok ? bf16_to_f32((bf16_t)(u.y >> 16)) : 0.f);
      }
      return;
    }
    const float* x = a.x + (size_t)rr * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = COL(i);
      const float4 u = *(const float4*)(x + min(c, d - 4));
      const bool ok = rok && c < d;
      v[i] = make_float4(ok ? u.x : 0.f, ok ? u.y : 0.f, ok ? u.z : 0.f, ok ? u.w : 0.f);
    }
  };
  float4 v[NV], nx[NV];
  unsigned nsat = 0, nsat2 = 0;                 // fp8 outputs: values clipped at +-448 (counted only in the fp8 store branches)
  load_row(wave0, v);
  for (int row = wave0; row < a.M; row += nwaves) {
    load_row(row + nwaves, nx);                 // next row of this wave: in flight under the reductions below
    float mean = 0.f, rstd = 1.f;
    if (sw_ln) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) sum += v[i].x + v[i].y + v[i].z + v[i].w;
      mean = wsum(sum) / (float)d;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (COL(i) < d) {
          const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
          sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
      }
      rstd = rsqrtf(wsum(sq) / (float)d + a.eps);
    }
    // The next row's values are taken out of the memory pipe HERE, before this row's stores are issued: `vmcnt` counts loads and
    // stores in one in-order queue, so a wait for the prefetch placed behind the stores would also wait for their round trip.
    if constexpr (CT) {
#pragma unroll
      for (int i = 0; i < NV; ++i)      // tied to rstd so that the scheduler cannot lift it above the reductions
        asm volatile("" : "+v"(nx[i].x), "+v"(nx[i].y), "+v"(nx[i].z), "+v"(nx[i].w), "+v"(rstd));
    }
    OutT* out = (OutT*)a.out + (size_t)row * d;
    const AddT* add = sw_add ? (const AddT*)a.add + (size_t)row * d : nullptr;
    float sum2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = COL(i);
      if (c >= d) continue;
      float o[4] = {(v[i].x - mean) * rstd * g[i].x + be[i].x, (v[i].y - mean) * rstd * g[i].y + be[i].y,
                    (v[i].z - mean) * rstd * g[i].z + be[i].z, (v[i].w - mean) * rstd * g[i].w + be[i].w};
      if (sw_silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (sizeof(OutT) <= 2) o[e] = o[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * o[e]));
          else o[e] = o[e] / (1.0f + expf(-o[e]));
        }
      }
      if (add) {
        if constexpr (sizeof(AddT) == 2) {
          const uint2 u = *(const uint2*)(add + c);
          o[0] += bf16_to_f32((bf16_t)(u.x & 0xffffu)); o[1] += bf16_to_f32((bf16_t)(u.x >> 16));
          o[2] += bf16_to_f32((bf16_t)(u.y & 0xffffu)); o[3] += bf16_to_f32((bf16_t)(u.y >> 16));
        } else {
          const float4 u = *(const float4*)(add + c);
          o[0] += u.x; o[1] += u.y; o[2] += u.z; o[3] += u.w;
        }
      }
      if constexpr (sizeof(OutT) == 4) *(float4*)(out + c) = make_float4(o[0], o[1], o[2], o[3]);
      v[i] = make_float4(o[0], o[1], o[2], o[3]);      // kept for the narrow stores below and for the second stage
      sum2 += o[0] + o[1] + o[2] + o[3];
    }
    if constexpr (sizeof(OutT) <= 2) {                 // one store per 8-column run (d % 8 == 0 is checked by the launcher)
#pragma unroll
      for (int i = 0; i < NV; i += 2) {
        const int c = COL(i);
        if (c >= d) continue;
        if constexpr (sizeof(OutT) == 1) {
          const float qs = a.out_inv_scale;            // fp8 operand of the next GEMM: value / (calibrated per-tensor scale)
          nsat += fp8_clipped(v[i].x * qs, v[i].y * qs, v[i].z * qs, v[i].w * qs) + fp8_clipped(v[i + 1].x * qs, v[i + 1].y * qs, v[i + 1].z * qs, v[i + 1].w * qs);
          *(uint2*)(out + c) = make_uint2(pack4_fp8(v[i].x * qs, v[i].y * qs, v[i].z * qs, v[i].w * qs),
                                          pack4_fp8(v[i + 1].x * qs, v[i + 1].y * qs, v[i + 1].z * qs, v[i + 1].w * qs));
        } else {
          *(uint4*)(out + c) = make_uint4(pack2_bf16(v[i].x, v[i].y), pack2_bf16(v[i].z, v[i].w),
                                          pack2_bf16(v[i + 1].x, v[i + 1].y), pack2_bf16(v[i + 1].z, v[i + 1].w));
        }
      }
    }
    if constexpr (TWO) {       // second LayerNorm on the row just produced (fp32 values, exactly what a separate pass would read)
      const float mean2 = wsum(sum2) / (float)d;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (COL(i) < d) {
          const float dx = v[i].x - mean2, dy = v[i].y - mean2, dz = v[i].z - mean2, dw = v[i].w - mean2;
          sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
      }
      const float rstd2 = rsqrtf(wsum(sq) / (float)d + a.eps2);
      AddT* o2 = (AddT*)a.out2 + (size_t)row * d;      // AddT is the compute dtype
#pragma unroll
      for (int i = 0; i < NV; i += 2) {
        const int c = COL(i);
        if (c >= d) continue;
        float q[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          q[4 * h + 0] = (v[i + h].x - mean2) * rstd2 * g2[i + h].x + be2[i + h].x;
          q[4 * h + 1] = (v[i + h].y - mean2) * rstd2 * g2[i + h].y + be2[i + h].y;
          q[4 * h + 2] = (v[i + h].z - mean2) * rstd2 * g2[i + h].z + be2[i + h].z;
          q[4 * h + 3] = (v[i + h].w - mean2) * rstd2 * g2[i + h].w + be2[i + h].w;
        }
        if (sw_o2f8) {
          const float qs = a.out2_inv_scale;
          nsat2 += fp8_clipped(q[0] * qs, q[1] * qs, q[2] * qs, q[3] * qs) + fp8_clipped(q[4] * qs, q[5] * qs, q[6] * qs, q[7] * qs);
          *(uint2*)((fp8_t*)a.out2 + (size_t)row * d + c) = make_uint2(pack4_fp8(q[0] * qs, q[1] * qs, q[2] * qs, q[3] * qs),
                                                                       pack4_fp8(q[4] * qs, q[5] * qs, q[6] * qs, q[7] * qs));
        } else if constexpr (sizeof(AddT) == 2) {
          *(uint4*)(o2 + c) = make_uint4(pack2_bf16(q[0], q[1]), pack2_bf16(q[2], q[3]), pack2_bf16(q[4], q[5]), pack2_bf16(q[6], q[7]));
        } else {
          *(float4*)(o2 + c) = make_float4(q[0], q[1], q[2], q[3]);
          *(float4*)(o2 + c + 4) = make_float4(q[4], q[5], q[6], q[7]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = nx[i];
  }
  if (nsat != 0 && a.sat) atomicAdd(a.sat, nsat);          // rare: only lanes that clipped something touch the counter
  if (nsat2 != 0 && a.sat2) atomicAdd(a.sat2, nsat2);
#undef COL
}

template <typename OutT, typename AddT>
static void launch_rownorm(hipStream_t s, const NormArgs& a) {
  // enough waves to cover the latency of a row's loads, few enough that gamma / beta are fetched once per many rows
  const int blocks = std::min(cdiv(a.M, 4), 256 * 8);
  const int sw = (a.x_bf16 ? 1 : 0) | (a.mode == NORM_LN ? 2 : 0) | (a.silu ? 4 : 0) | (a.add ? 8 : 0) | (a.out2_fp8 ? 16 : 0);
  // the encoder's shapes (d <= 1024, i.e. NV = 4) and switch combinations get branch-free instantiations; everything else runs
  // the generic kernel
  if (a.d > 512 && a.d <= 1024) {
#define RVB_NORM_CASE(TWO_, SW_) \
    if ((a.out2 != nullptr) == (TWO_) && sw == (SW_)) { hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 4, TWO_, SW_>), dim3(blocks), dim3(256), 0, s, a); return; }
    RVB_NORM_CASE(false, 2)        // LayerNorm in front of a GEMM
    RVB_NORM_CASE(false, 2 | 4 | 1)   // the convolution module's LayerNorm + SiLU on the bf16 depthwise output
    RVB_NORM_CASE(false, 4 | 1)       // ... its BatchNorm (affine) form
    RVB_NORM_CASE(false, 2 | 4)       // the same two on an fp32 depthwise output
    RVB_NORM_CASE(false, 4)
    RVB_NORM_CASE(true, 2)         // norm_final + the next block's first LayerNorm
    RVB_NORM_CASE(true, 2 | 8)     // ... with the language-specific `add`
    RVB_NORM_CASE(true, 2 | 16)    // ... with an fp8 second output
    RVB_NORM_CASE(true, 2 | 8 | 16)
#undef RVB_NORM_CASE
  }
  if (a.out2) {
    if (a.d <= 512) hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 2, true, -1>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.d <= 1024) hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 4, true, -1>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 8, true, -1>), dim3(blocks), dim3(256), 0, s, a);
  } else {
    if (a.d <= 512) hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 2, false, -1>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.d <= 1024) hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 4, false, -1>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rownorm_kernel<OutT, AddT, 8, false, -1>), dim3(blocks), dim3(256), 0, s, a);
  }
}

int rownorm(hipStream_t s, int dtype, const NormArgs& a) {
  if (a.M <= 0) return OK;
  if (a.d % 8 || a.d > 2048) { set_error("rownorm: d must be a multiple of 8 and <= 2048"); return E_ARG; }
  if (a.x_bf16 && dtype != DT_BF16) { set_error("rownorm: bf16 input belongs to the bf16 engine"); return E_ARG; }
  if (a.out2 && (a.mode != NORM_LN || !(a.out_f32 || dtype == DT_F32))) {
    set_error("rownorm: the fused second LayerNorm follows a LayerNorm with fp32 output"); return E_ARG;
  }
  if ((a.out_fp8 || a.out2_fp8) && dtype != DT_BF16) { set_error("rownorm: fp8 outputs belong to the bf16 engine"); return E_ARG; }
  if (dtype == DT_BF16) {
    if (a.out_fp8) launch_rownorm<fp8_t, bf16_t>(s, a);
    else if (a.out_f32) launch_rownorm<float, bf16_t>(s, a);
    else launch_rownorm<bf16_t, bf16_t>(s, a);
  } else {
    launch_rownorm<float, float>(s, a);
  }
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// GLU + depthwise conv along time (per chunk), LDS halo tile of 128 time steps x 128 channels
// ------------------------------------------------------------------------------------------------
static constexpr int DW_TT = 128, DW_CT = 128, DW_KMAX = 31, DW_THREADS = 512;

// the gated value as it sits in LDS: fp32 in the f32 engine; bf16 in the bf16 engine (F.glu under autocast returns bf16 as
// well, and half the LDS bytes let four waves per SIMD stay resident)
template <typename T> struct Gate;
template <> struct Gate<float> {
  typedef float2 pair_t;
  __device__ static inline float put(float v) { return v; }
  __device__ static inline void get(pair_t p, float& x, float& y) { x = p.x; y = p.y; }
};
template <> struct Gate<bf16_t> {
  typedef uint32_t pair_t;
  __device__ static inline bf16_t put(float v) { return f32_to_bf16(v); }
  __device__ static inline void get(pair_t p, float& x, float& y) { x = __uint_as_float(p << 16); y = __uint_as_float(p & 0xffff0000u); }
};

// block = 128 time steps x 128 channels, 8 waves.  Staging: one thread per (row, 16-byte channel group) -- the two GLU halves
// are read as 16-byte vectors (many bytes in flight per lane), gated, and written to LDS.  Compute: a lane owns 2 adjacent
// channels and 16 consecutive frames (4- or 8-byte stores); its 31 taps (tap-major in memory: one coalesced 512-byte load per
// wave and tap) are loaded before the staging so that their latency hides under it.
template <typename T>
__global__ __launch_bounds__(DW_THREADS) void glu_dw_kernel(GluDwArgs a) {
  typedef decltype(Gate<T>::put(0.f)) S;
  constexpr int rows = DW_TT + DW_KMAX - 1;        // always the full window: rows past DW_TT + K - 1 are zero-filled
  __shared__ __attribute__((aligned(16))) S s_g[rows][DW_CT];
  const int c = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int t0 = blockIdx.x * DW_TT, c0 = blockIdx.y * DW_CT, b = blockIdx.z;
  const int ch = c0 + 2 * c;
  const bool cok = ch < a.d;            // d is even: both channels of the pair are in range together
  const int K = a.K, pad = a.causal ? K - 1 : (K - 1) / 2;
  const int len = a.lens[b];
  const int lorder = K - 1;
  const T* G = (const T*)a.G;
  const bool gated = a.gated != 0;                       // G is [B*T, d], gated by the pointwise GEMM's epilogue (ACT_GLU); vector path only
  const size_t gstride = gated ? (size_t)a.d : (size_t)2 * a.d;
  // the two channels of a lane as one 2-wide vector: v_pk_fma_f32 does both FMAs of a tap in one instruction
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t wk[DW_KMAX];
#pragma unroll
  for (int k = 0; k < DW_KMAX; ++k)
    wk[k] = (k < K && cok) ? *(const f32x2_t*)(a.dw_w + (size_t)k * a.d + ch) : (f32x2_t){0.f, 0.f};
  constexpr int VE = 16 / (int)sizeof(T);          // channels per 16-byte vector
  constexpr int NG = DW_CT / VE;                   // vector groups per row
  const bool vec_ok = (a.d % VE) == 0;
  // Staging in two sweeps when rows are whole 16-byte vectors (every model shape): first ALL of a thread's loads, unconditional
  // (addresses clamped into the chunk; what lies outside is replaced below), then the gating and the LDS writes.  With the loads
  // predicated inside one loop the compiler waited for each item's pair of vectors before it issued the next item's: five
  // memory round trips per workgroup instead of one.
  if (vec_ok) {
    constexpr int ITEMS_ALL = (rows * NG + DW_THREADS - 1) / DW_THREADS;
    constexpr int ITEMS = 3;                       // per sweep: 24 registers of loads in flight (the 31 taps are live as well)
#pragma unroll
    for (int it0 = 0; it0 < ITEMS_ALL; it0 += ITEMS) {
    uint4 ra[ITEMS], rb[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int v = min((int)threadIdx.x + (it0 + it) * DW_THREADS, rows * NG - 1);
      const int r = v / NG, grp = v - r * NG;
      const int t = t0 - pad + r;
      const int cg = min(c0 + grp * VE, a.d - VE);
      const bool from_hist = a.causal && t < 0 && t >= -a.hist_rows;
      const T* gr = from_hist ? (const T*)a.hist + (size_t)(lorder + t) * 2 * a.d
                              : G + ((size_t)b * a.T + min(max(t, 0), a.T - 1)) * gstride;
      ra[it] = *(const uint4*)(gr + cg);
      rb[it] = gated ? make_uint4(0, 0, 0, 0) : *(const uint4*)(gr + a.d + cg);
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int v = threadIdx.x + (it0 + it) * DW_THREADS;
      if (v >= rows * NG) break;
      const int r = v / NG, grp = v - r * NG;
      const int t = t0 - pad + r;
      const int cg = c0 + grp * VE;
      float g[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = 0.f;
      const bool from_hist = a.causal && t < 0 && t >= -a.hist_rows;
      if (cg < a.d && (t >= 0 || a.causal) && t < a.T && r < DW_TT + K - 1) {
        if ((t >= 0 && t < len) || from_hist) {
          T av[VE], bvv[VE];
          *(uint4*)av = ra[it];
          *(uint4*)bvv = rb[it];
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const float a0 = Cvt<T>::to_f32(av[e]), b0 = Cvt<T>::to_f32(bvv[e]);
            if (gated) g[e] = a0;          // the pointwise GEMM's epilogue gated it (ACT_GLU)
            else if constexpr (sizeof(T) == 2) g[e] = a0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * b0));
            else g[e] = a0 / (1.0f + expf(-b0));
          }
        } else {
#pragma unroll
          for (int e = 0; e < VE; ++e)
            if (cg + e < a.d) g[e] = a.pw1_bias[cg + e] / (1.0f + expf(-a.pw1_bias[a.d + cg + e]));
        }
      }
      S sv[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) sv[e] = Gate<T>::put(g[e]);
      uint4* dst = (uint4*)&s_g[r][grp * VE];
#pragma unroll
      for (int q = 0; q < (int)(VE * sizeof(S)) / 16; ++q) dst[q] = ((const uint4*)sv)[q];
    }
    }
  } else {
#pragma unroll 4
  for (int v = threadIdx.x; v < rows * NG; v += DW_THREADS) {
    const int r = v / NG, grp = v - r * NG;
    const int t = t0 - pad + r;
    const int cg = c0 + grp * VE;                  // first channel of the group
    float g[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) g[e] = 0.f;
    // left context of a causal module: cached frames, or the zero padding seen through pointwise_conv1 + GLU
    const bool from_hist = a.causal && t < 0 && t >= -a.hist_rows;
    if (cg < a.d && (t >= 0 || a.causal) && t < a.T && r < DW_TT + K - 1) {
      if ((t >= 0 && t < len) || from_hist) {
        const T* gr = from_hist ? (const T*)a.hist + (size_t)(lorder + t) * 2 * a.d : G + ((size_t)b * a.T + t) * 2 * a.d;
        T av[VE], bvv[VE];
        if (vec_ok) {
          *(uint4*)av = *(const uint4*)(gr + cg);
          *(uint4*)bvv = *(const uint4*)(gr + a.d + cg);
        } else {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            av[e] = cg + e < a.d ? gr[cg + e] : Cvt<T>::from_f32(0.f);
            bvv[e] = cg + e < a.d ? gr[a.d + cg + e] : Cvt<T>::from_f32(0.f);
          }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float a0 = Cvt<T>::to_f32(av[e]), b0 = Cvt<T>::to_f32(bvv[e]);
          if constexpr (sizeof(T) == 2)   // bf16 mode: hardware exp2/rcp (the gate is rounded to bf16 precision anyway)
            g[e] = a0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * b0));
          else
            g[e] = a0 / (1.0f + expf(-b0));
        }
      } else {
        // GLU of the pointwise-conv1 bias: what a zero-masked (padded) frame yields (convolution.py:107-118)
#pragma unroll
        for (int e = 0; e < VE; ++e)
          if (cg + e < a.d) g[e] = a.pw1_bias[cg + e] / (1.0f + expf(-a.pw1_bias[a.d + cg + e]));
      }
    }
    S sv[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) sv[e] = Gate<T>::put(g[e]);
    uint4* dst = (uint4*)&s_g[r][grp * VE];        // 16 bytes (bf16) or 2 x 16 bytes (fp32) per group
#pragma unroll
    for (int q = 0; q < (int)(VE * sizeof(S)) / 16; ++q) dst[q] = ((const uint4*)sv)[q];
  }
  }
  __syncthreads();
  if (!cok) return;
  // Each thread owns 16 CONSECUTIVE output frames of its channel pair: the 46 input rows they touch are read from
  // LDS once each and fanned out to the (up to 16) outputs they contribute to, taps in registers -- 46 LDS reads
  // per thread instead of one per FMA pair (the strided version was LDS-bandwidth bound).
  constexpr int OPT = DW_TT / (DW_THREADS / 64);   // outputs per thread
  f32x2_t acc[OPT];
  const f32x2_t bv = {a.dw_b[ch], a.dw_b[ch + 1]};
#pragma unroll
  for (int i = 0; i < OPT; ++i) acc[i] = bv;
  const int rbase = slot * OPT;
#pragma unroll
  for (int r = 0; r < OPT + DW_KMAX - 1; ++r) {
    float gx, gy;
    Gate<T>::get(*(const typename Gate<T>::pair_t*)&s_g[rbase + r][2 * c], gx, gy);
    const f32x2_t g = {gx, gy};
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
      const int k = r - i;                        // compile-time after unrolling
      if (k >= 0 && k < DW_KMAX) acc[i] = __builtin_elementwise_fma(wk[k], g, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < OPT; ++i) {
    const int t = t0 + rbase + i;
    if (t < a.T) {
      if (a.out_bf16) *(uint32_t*)((bf16_t*)a.out + ((size_t)b * a.T + t) * a.d + ch) = pack2_bf16(acc[i].x, acc[i].y);
      else *(float2*)(a.out + ((size_t)b * a.T + t) * a.d + ch) = make_float2(acc[i].x, acc[i].y);
    }
  }
}

int glu_dwconv(hipStream_t s, int dtype, const GluDwArgs& a) {
  if (a.B <= 0 || a.T <= 0) return OK;
  if (a.K > DW_KMAX || a.K < 1 || (!a.causal && (a.K % 2) == 0) || (a.d % 2)) {
    set_error("glu_dwconv: kernel must be <= 31 (and odd unless causal), d even"); return E_ARG;
  }
  if (a.gated && (a.hist_rows > 0 || (a.d % (dtype == DT_BF16 ? 8 : 4)) != 0)) {
    set_error("glu_dwconv: a gated input comes in whole 16-byte vectors and without a streaming history"); return E_ARG;
  }
  if (a.hist_rows > 0 && (!a.causal || a.B != 1 || a.hist_rows > a.K - 1 || !a.hist)) {
    set_error("glu_dwconv: a left-context history belongs to one causal stream"); return E_ARG;
  }
  dim3 grid(cdiv(a.T, DW_TT), cdiv(a.d, DW_CT), a.B);
  if (dtype == DT_BF16) hipLaunchKernelGGL(glu_dw_kernel<bf16_t>, grid, dim3(DW_THREADS), 0, s, a);
  else hipLaunchKernelGGL(glu_dw_kernel<float>, grid, dim3(DW_THREADS), 0, s, a);
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
// calibration of the fp8 activation scales: running max |x| of a tensor
template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ x, size_t n, float* __restrict__ slot) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(Cvt<T>::to_f32(x[i])));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned int*)slot, __float_as_uint(m));     // non-negative floats order as their bits
}
int amax_abs(hipStream_t s, int dtype, const void* x, size_t n, float* slot) {
  if (n == 0) return OK;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
  if (dtype == DT_BF16) hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, n, slot);
  else hipLaunchKernelGGL(amax_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, n, slot);
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

// ------------------------------------------------------------------------------------------------
// beam reorder of a per-hypothesis cache: dst[r][t][:] = src[parent[r]][t][:] for t < rows
// (search.py:307-312 `torch.index_select(c, dim=0, index=cache_index)`); row_bytes % 16 == 0
__global__ __launch_bounds__(256) void gather_cache_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                           const int* __restrict__ parent, int L, int rows, int row_bytes) {
  const int r = blockIdx.y;
  const size_t so = (size_t)parent[r] * L * row_bytes, dof = (size_t)r * L * row_bytes;
  const int nvec = rows * (row_bytes / 16);
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nvec; v += gridDim.x * 256)
    ((uint4*)(dst + dof))[v] = ((const uint4*)(src + so))[v];
}
int gather_cache(hipStream_t s, const void* src, void* dst, const int* parent, int R, int L, int rows, int row_bytes) {
  if (R <= 0 || rows <= 0) return OK;
  if (row_bytes % 16) { set_error("gather_cache: rows must be multiples of 16 bytes"); return E_ARG; }
  const int nvec = rows * (row_bytes / 16);
  dim3 grid(cdiv(nvec, 256) < 8 ? cdiv(nvec, 256) : 8, R);
  hipLaunchKernelGGL(gather_cache_kernel, grid, dim3(256), 0, s, (const char*)src, (char*)dst, parent, L, rows, row_bytes);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
