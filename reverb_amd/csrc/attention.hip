// Fused multi-head attention for gfx950 (flash-style, scores never leave the CU).
//
// One kernel covers the three attention forms on the Reverb-ASR hot path (asr/wenet/transformer/):
//   * encoder RelPositionMultiHeadedAttention.forward, attention.py:317-399 (rel_shift removed):
//         scores = ((q+u).k^T + (q+v).p^T) / sqrt(dk), key-padding mask, softmax, .v
//     -> HAS_POS: the two products are one contraction over [k | p] with the two biased copies
//        of q as A operands.
//   * decoder self attention (causal & ragged hypotheses)  attention.py:129-197, decoder.py:150-156
//   * decoder cross attention over the chunk's encoder memory (key length = valid frames)
//   forward_attention, attention.py:81-127: masked_fill(-inf) -> softmax -> masked_fill(0).
//
// Workgroup = 4 waves = 64 query rows of one (sequence, head); each wave owns 16 rows.
// Per 64-key tile: K (and P) rows and V^T are staged in LDS, S = Q.K^T on MFMA 16x16 fragments,
// online softmax with 16-lane shuffle reductions, probabilities go through a per-wave LDS patch to
// become the A operand of P.V.  T = bf16 (v_mfma_f32_16x16x32_bf16) or f32 (v_mfma_f32_16x16x4_f32).
#include "common.h"
#include "kernels.h"

namespace rvb {

static constexpr int KT = 64;  // keys per tile

template <typename T, int DKP>
struct AttnLds {
  static constexpr int ROW_K = DKP * (int)sizeof(T) + 16;   // K / P rows (bytes)
  static constexpr int ROW_V = KT * (int)sizeof(T) + 16;    // V^T rows and P-patch rows (bytes)
  static constexpr int OFF_K = 0;
  static constexpr int OFF_P = OFF_K + KT * ROW_K;
  static constexpr int OFF_V = OFF_P + KT * ROW_K;
  static constexpr int OFF_W = OFF_V + DKP * ROW_V;
  static constexpr int TOTAL = OFF_W + 4 * 16 * ROW_V;
};

template <typename T, int DKP, bool HAS_POS>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = AttnLds<T, DKP>;
  constexpr int VE = Mma16<T>::VE;
  constexpr int KC = Mma16<T>::KC;
  constexpr int NCH = DKP / KC;     // K-chunks of the q.k contraction
  constexpr int NKC = KT / KC;      // K-chunks of the p.v contraction
  constexpr int NOF = DKP / 16;     // output fragments along dk
  constexpr int VPR = DKP / VE;     // 16-byte vectors per staged row

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int qlen = a.q_len[seq];
  const int q0 = blockIdx.x * 64;
  if (q0 >= qlen) return;                       // block-uniform
  const int qs = a.q_start[seq], ks = a.kv_start[seq], kvlen = a.kv_len[seq];
  const int dk = a.dk;
  const T* Q = (const T*)a.q;
  const T* K = (const T*)a.k;
  const T* V = (const T*)a.v;
  const T* P = (const T*)a.p;

  const int lrow = lane & 15;          // A/B operand row inside a 16-row fragment
  const int lgrp = lane >> 4;          // which 16-byte vector of the 64-byte chunk
  const int crow = lgrp * 4;           // C layout: rows crow..crow+3, column lrow

  // ---- Q fragments (A operands): this wave's 16 rows, biased copies ----
  uint4 qu[NCH], qv[HAS_POS ? NCH : 1];
  {
    const int qr = q0 + wave * 16 + lrow;
    const bool rok = qr < qlen;
    const T* qp = Q + (size_t)(qs + (rok ? qr : 0)) * a.q_stride + head * dk;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int e0 = ch * KC + lgrp * VE;
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (rok && e0 < dk) raw = *(const uint4*)(qp + e0);
      if (a.bias_u != nullptr) {
        T tmp[VE], ou[VE], ov[VE];
        *(uint4*)tmp = raw;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const bool ok = rok && (e0 + e) < dk;
          const float qf = Cvt<T>::to_f32(tmp[e]);
          ou[e] = Cvt<T>::from_f32(ok ? qf + a.bias_u[head * dk + e0 + e] : 0.f);
          ov[e] = Cvt<T>::from_f32(ok ? qf + a.bias_v[head * dk + e0 + e] : 0.f);
        }
        qu[ch] = *(uint4*)ou;
        if constexpr (HAS_POS) qv[ch] = *(uint4*)ov;
      } else {
        qu[ch] = raw;
        if constexpr (HAS_POS) qv[ch] = raw;
      }
    }
  }

  f32x4_t o[NOF];
#pragma unroll
  for (int f = 0; f < NOF; ++f) o[f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

  char* sK = smem + L::OFF_K;
  char* sP = smem + L::OFF_P;
  char* sV = smem + L::OFF_V;
  char* sW = smem + L::OFF_W + wave * 16 * L::ROW_V;

  const float inv_sqrt_dk = 1.0f / a.sqrt_dk;   // bf16 mode multiplies; f32 mode divides like the reference
  int kend = kvlen;
  if (a.causal) kend = min(kvlen, q0 + 64);
  // Staging: tile t+1 is fetched HBM -> VGPR while tile t is being multiplied (issue early, write to
  // LDS late); K and P rows are stored as they come, V is stored transposed (consecutive lanes take
  // consecutive keys so the 2-/4-byte transposed writes of a wave are conflict-free).
  constexpr int NV = (KT * VPR + 255) / 256;
  constexpr bool PREFETCH = false;   // measured: holding tile t+1 in VGPRs costs a wave of occupancy and is slower (33.2 vs 29.8 ms)
  uint4 rk[NV], rp[HAS_POS ? NV : 1], rv[NV];
  auto gload = [&](int kt0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + 256 * n;
      rk[n] = make_uint4(0, 0, 0, 0); rv[n] = make_uint4(0, 0, 0, 0);
      if constexpr (HAS_POS) rp[n] = make_uint4(0, 0, 0, 0);
      if (i < KT * VPR) {
        const int r = i / VPR, c = i - r * VPR;
        const int key = kt0 + r;
        if (key < kvlen && c * VE < dk) {
          rk[n] = *(const uint4*)(K + (size_t)(ks + key) * a.k_stride + head * dk + c * VE);
          if constexpr (HAS_POS) rp[n] = *(const uint4*)(P + (size_t)key * a.p_stride + head * dk + c * VE);
        }
        const int c2 = i / KT, r2 = i - c2 * KT;
        const int key2 = kt0 + r2;
        if (key2 < kvlen && c2 * VE < dk)
          rv[n] = *(const uint4*)(V + (size_t)(ks + key2) * a.v_stride + head * dk + c2 * VE);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + 256 * n;
      if (i < KT * VPR) {
        const int r = i / VPR, c = i - r * VPR;
        *(uint4*)(sK + r * L::ROW_K + c * 16) = rk[n];
        if constexpr (HAS_POS) *(uint4*)(sP + r * L::ROW_K + c * 16) = rp[n];
        const int c2 = i / KT, r2 = i - c2 * KT;
        T tv[VE];
        *(uint4*)tv = rv[n];
#pragma unroll
        for (int e = 0; e < VE; ++e) *(T*)(sV + (c2 * VE + e) * L::ROW_V + r2 * sizeof(T)) = tv[e];
      }
    }
  };
  if (PREFETCH && kend > 0) gload(0);
  for (int kt0 = 0; kt0 < kend; kt0 += KT) {
    if (!PREFETCH) gload(kt0);
    lstore();
    __syncthreads();
    if (PREFETCH && kt0 + KT < kend) gload(kt0 + KT);   // lands under the MFMAs / softmax below

    // ---- S = Qu.K^T (+ Qv.P^T) : 4 fragments of 16 keys ----
    f32x4_t s[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      s[nf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const uint4 bk = *(const uint4*)(sK + (nf * 16 + lrow) * L::ROW_K + ch * 64 + lgrp * 16);
        Mma16<T>::run(qu[ch], bk, s[nf]);
        if constexpr (HAS_POS) {
          const uint4 bp = *(const uint4*)(sP + (nf * 16 + lrow) * L::ROW_K + ch * 64 + lgrp * 16);
          Mma16<T>::run(qv[ch], bp, s[nf]);
        }
      }
    }
    // ---- mask, online softmax ----
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = q0 + wave * 16 + crow + r;
      float mx = -INFINITY;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int key = kt0 + nf * 16 + lrow;
        float v = sizeof(T) == 2 ? s[nf][r] * inv_sqrt_dk : s[nf][r] / a.sqrt_dk;
        if (key >= kvlen || (a.causal && key > qrow)) v = -INFINITY;
        s[nf][r] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float m_new = fmaxf(m_run[r], mx);
      float al = 1.f, ps = 0.f;
      if (m_new == -INFINITY) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) s[nf][r] = 0.f;
      } else {
        al = (m_run[r] == -INFINITY) ? 0.f : (sizeof(T) == 2 ? __expf(m_run[r] - m_new) : expf(m_run[r] - m_new));
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const float pv = sizeof(T) == 2 ? __expf(s[nf][r] - m_new) : expf(s[nf][r] - m_new);   // exp(-inf) = 0 for masked keys
          s[nf][r] = pv;
          ps += pv;
        }
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) ps += __shfl_xor(ps, off, 64);
      l_run[r] = l_run[r] * al + ps;
      m_run[r] = m_new;
      alpha[r] = al;
    }
#pragma unroll
    for (int f = 0; f < NOF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[f][r] *= alpha[r];
    // ---- probabilities -> per-wave LDS patch (row-major [16][64]) ----
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *(T*)(sW + (crow + r) * L::ROW_V + (nf * 16 + lrow) * sizeof(T)) = Cvt<T>::from_f32(s[nf][r]);
    __builtin_amdgcn_wave_barrier();   // the patch is private to this wave: LDS ops of one wave execute in order
    // ---- O += P.V ----
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      const uint4 pa = *(const uint4*)(sW + lrow * L::ROW_V + kc * 64 + lgrp * 16);
#pragma unroll
      for (int f = 0; f < NOF; ++f) {
        const uint4 vb = *(const uint4*)(sV + (f * 16 + lrow) * L::ROW_V + kc * 64 + lgrp * 16);
        Mma16<T>::run(pa, vb, o[f]);
      }
    }
    __syncthreads();
  }

  // ---- normalise and store ----
  T* O = (T*)a.out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = q0 + wave * 16 + crow + r;
    if (qrow >= qlen) continue;
    const float inv = l_run[r] > 0.f ? 1.0f / l_run[r] : 0.f;
    T* orow = O + (size_t)(qs + qrow) * a.o_stride + head * dk;
#pragma unroll
    for (int f = 0; f < NOF; ++f) {
      const int col = f * 16 + lrow;
      if (col < dk) orow[col] = Cvt<T>::from_f32(o[f][r] * inv);
    }
  }
}

template <typename T, int DKP, bool HAS_POS>
static int launch_attn(hipStream_t s, const AttnArgs& a) {
  using L = AttnLds<T, DKP>;
  static bool attr_set = false;
  auto kern = attn_kernel<T, DKP, HAS_POS>;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL));
    attr_set = true;
  }
  dim3 grid(cdiv(a.max_q, 64), a.heads, a.nseq);
  hipLaunchKernelGGL(kern, grid, dim3(256), L::TOTAL, s, a);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

template <typename T>
static int dispatch_attn(hipStream_t s, const AttnArgs& a) {
  const bool pos = a.p != nullptr;
  const int dk = a.dk;
#define RVB_ATTN_CASE(D)                                                           \
  if (dk <= D) return pos ? launch_attn<T, D, true>(s, a) : launch_attn<T, D, false>(s, a);
  RVB_ATTN_CASE(32)
  RVB_ATTN_CASE(64)
  RVB_ATTN_CASE(96)
  RVB_ATTN_CASE(128)
#undef RVB_ATTN_CASE
  set_error("attention: head dim > 128 unsupported");
  return E_UNSUPPORTED;
}

int attention(hipStream_t s, int dtype, const AttnArgs& a) {
  if (a.nseq <= 0 || a.max_q <= 0) return OK;
  const int ve = dtype == DT_BF16 ? 8 : 4;
  if (a.dk % ve || a.q_stride % ve || a.k_stride % ve || a.v_stride % ve || (a.p && a.p_stride % ve)) {
    set_error("attention: dk and row strides must be multiples of the 16-byte vector width");
    return E_ARG;
  }
  if ((a.bias_u == nullptr) != (a.bias_v == nullptr)) { set_error("attention: bias_u/bias_v must come together"); return E_ARG; }
  return dtype == DT_BF16 ? dispatch_attn<bf16_t>(s, a) : dispatch_attn<float>(s, a);
}

}  // namespace rvb
