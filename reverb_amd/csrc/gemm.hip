// MFMA GEMM for gfx950:  C[M,N] = res + alpha * act(A[M,K] . W[N,K]^T + bias)
//
// Replaces every torch.nn.Linear / 1x1 Conv1d / (as implicit GEMM) the second 3x3 stride-2
// Conv2d of Conv2dSubsampling4 on the reference hot path:
//   asr/wenet/transformer/positionwise_feed_forward.py:47-55, attention.py:72-74,127,374,
//   convolution.py:107-144 (pointwise convs), subsampling.py:189-190,223 , ctc.py:114,
//   decoder.py:164-167.
//
// One kernel template serves both compute modes (Mma16<T> in common.h):
//   T = bf16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate       (throughput mode)
//   T = f32  : v_mfma_f32_16x16x4_f32, exact f32 fma chain     (parity mode)
// Tile 128x128, 128 bytes of K per row per step (64 bf16 / 32 f32), 256 threads = 4 waves in
// 2x2, each wave 64x64 = 4x4 MFMA fragments.  Operands are staged HBM -> VGPR -> LDS with the
// next tile's global loads issued before the current tile's MFMAs (loads fly under compute),
// LDS rows padded 128 -> 144 bytes so the 16 lanes of a ds_read_b128 group hit 16 distinct
// 16-byte slots.  Block ids are remapped so each XCD (private L2) owns a contiguous run of
// output tiles that share A rows.
#include "common.h"
#include "kernels.h"

namespace rvb {

static constexpr int BM = 128, BN = 128;
static constexpr int ROWB = 128;      // bytes of K per tile row
static constexpr int LDSB = 144;      // padded LDS row stride (bytes)
static constexpr int GEMM_LDS = 2 * (BM + BN) * LDSB;  // double buffered: 73,728 B

__device__ inline float act_apply(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_LRELU) return v > 0.0f ? v : 0.01f * v;
  return v;
}

template <typename T, typename OutT, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VE = Mma16<T>::VE;        // elements per 16-byte vector
  constexpr int BKE = ROWB / sizeof(T);   // K elements per tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware tile mapping (bijective for any grid size) ----
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;

  // ---- per-thread load coordinates: vector column c16 of rows lr + 32*i ----
  const int c16 = tid & 7;
  const int lr = tid >> 3;
  size_t a_off[4];
  bool a_ok[4], w_ok[4];
  size_t w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lr + 32 * i;
    a_ok[i] = m < p.M;
    if (CONV) {
      // A[m][k] = X1[b, 2*t2+kh, 2*f2+kw, c]  (NHWC), m = (b*T2 + t2)*F2 + f2
      const int tf = p.cT2 * p.cF2;
      const int b = m / tf;
      const int rem = m - b * tf;
      const int t2 = rem / p.cF2, f2 = rem - t2 * p.cF2;
      a_off[i] = (((size_t)b * p.cT1 + 2 * t2) * p.cF1 + 2 * f2) * (size_t)p.cC;
    } else {
      a_off[i] = (size_t)m * p.lda;
    }
    const int n = n0 + lr + 32 * i;
    w_ok[i] = n < p.N;
    w_off[i] = (size_t)n * p.ldw;
  }

  uint4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
    const int kv = k0 + c16 * VE;
    const bool kok = kv < p.K;
    size_t koff = kv;
    if (CONV) {
      const int kk = kv / p.cC;
      const int cin = kv - kk * p.cC;
      const int kh = kk / 3, kw = kk - kh * 3;
      koff = ((size_t)kh * p.cF1 + kw) * p.cC + cin;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = (a_ok[i] && kok) ? *(const uint4*)(A + a_off[i] + koff) : make_uint4(0, 0, 0, 0);
      rb[i] = (w_ok[i] && kok) ? *(const uint4*)(W + w_off[i] + kv) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    char* sA = smem + buf * (BM + BN) * LDSB;
    char* sB = sA + BM * LDSB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(sA + (lr + 32 * i) * LDSB + c16 * 16) = ra[i];
      *(uint4*)(sB + (lr + 32 * i) * LDSB + c16 * 16) = rb[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BKE - 1) / BKE;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 16;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BKE);   // in flight under the MFMAs below
    const char* sA = smem + cur * (BM + BN) * LDSB + (wr * 64 + frow) * LDSB + fk;
    const char* sB = smem + cur * (BM + BN) * LDSB + BM * LDSB + (wc * 64 + frow) * LDSB + fk;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      uint4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *(const uint4*)(sA + i * 16 * LDSB + ch * 64);
        b[i] = *(const uint4*)(sB + i * 16 * LDSB + ch * 64);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma16<T>::run(a[i], b[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  OutT* __restrict__ C = (OutT*)p.C;
  const int crow = (lane >> 4) * 4;
  const int ccol = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wc * 64 + j * 16 + ccol;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + i * 16 + crow + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r] + bv;
        v = act_apply(v, p.act) * p.alpha;
        if (p.res) v += p.res[(size_t)row * p.ldres + col];
        if (p.rowadd && col >= p.rowadd_col0 && col < p.rowadd_col0 + p.rowadd_cols)      // row-periodic bf16 addend (kernels.h)
          v += bf16_to_f32(((const bf16_t*)p.rowadd)[(size_t)(row % p.rowadd_rows) * p.rowadd_ld + (col - p.rowadd_col0)]);
        C[(size_t)row * p.ldc + col] = Cvt<OutT>::from_f32(v);
      }
    }
  }
}

template <typename T, typename OutT, bool CONV>
static int launch(hipStream_t s, const GemmArgs& p) {
  static bool attr_set = false;
  auto kern = gemm_kernel<T, OutT, CONV>;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS));
    attr_set = true;
  }
  const int tiles = cdiv(p.M, BM) * cdiv(p.N, BN);
  if (tiles <= 0) return OK;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), GEMM_LDS, s, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int g_gemm_variant = 0;

int gemm(hipStream_t s, int dtype, const GemmArgs& p) {
  const int ve = dtype == DT_BF16 ? 8 : 4;
  if (p.M < 0 || p.N <= 0 || p.K <= 0 || (p.K % ve) || (p.lda % ve) || (p.ldw % ve) ||
      (p.conv && (p.cC % ve))) {
    set_error("gemm: K, lda, ldw (and conv channels) must be multiples of the 16-byte vector width");
    return E_ARG;
  }
  if (p.M == 0) return OK;
  if (p.rowadd && (dtype != DT_BF16 || p.out_f32 || p.in_fp8 || p.out_fp8 || p.rowadd_rows <= 0)) {
    set_error("gemm: the row-periodic addend exists for bf16 operands and bf16 output only");
    return E_UNSUPPORTED;
  }
  if (p.act == ACT_GLU) {     // pairs of columns gated in the epilogue: gemm2's phase-interleaved bf16 kernel only
    if (g_gemm_variant == 1 || !gemm_glu_supported(dtype, p)) { set_error("gemm: ACT_GLU needs the bf16 LDS-DMA kernel (see gemm_glu_supported)"); return E_UNSUPPORTED; }
    return gemm2(s, dtype, p);
  }
  if (p.in_fp8) {       // fp8 operands exist only on the LDS-DMA kernel
    if (!gemm2_applicable(dtype, p)) { set_error("gemm: fp8 operands need the bf16 engine, K % 128 == 0 and per-channel weight scales"); return E_ARG; }
    return gemm2(s, dtype, p);
  }
  // auto: the 256x256 LDS-DMA kernel for bf16 (1.5-2x); f32 stays on the 128x128 kernel, which already
  // runs at ~70 % of the 157 TFLOP/s f32 MFMA peak (profiles/r01_gemm_microbench.md)
  if (g_gemm_variant != 1 && (dtype == DT_BF16 || g_gemm_variant == 2) && gemm2_applicable(dtype, p))
    return gemm2(s, dtype, p);
  if (dtype == DT_BF16) {
    if (p.out_f32) return p.conv ? launch<bf16_t, float, true>(s, p) : launch<bf16_t, float, false>(s, p);
    return p.conv ? launch<bf16_t, bf16_t, true>(s, p) : launch<bf16_t, bf16_t, false>(s, p);
  }
  return p.conv ? launch<float, float, true>(s, p) : launch<float, float, false>(s, p);
}

}  // namespace rvb
