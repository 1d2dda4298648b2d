// Fused multi-head attention for gfx950 (flash-style, scores never leave the CU).
//
// One kernel covers the three attention forms on the Reverb-ASR hot path (asr/wenet/transformer/):
//   * encoder RelPositionMultiHeadedAttention.forward, attention.py:317-399 (rel_shift removed):
//         scores = ((q+u).k^T + (q+v).p^T) / sqrt(dk), key-padding mask, softmax, .v
//     -> HAS_POS: the two products are one accumulation chain over [k | p] with the two biased
//        copies of q as operands.
//   * decoder self attention (causal & ragged hypotheses)  attention.py:129-197, decoder.py:150-156
//   * decoder cross attention over the chunk's encoder memory (key length = valid frames)
//   forward_attention, attention.py:81-127: masked_fill(-inf) -> softmax -> masked_fill(0).
//
// Workgroup = 8 waves = 128 query rows of one (sequence, head); each wave owns 16 queries.
// Per 64-key tile K (and P) rows and V^T are staged in LDS once for all 8 waves.  The scores are
// computed TRANSPOSED, S^T = K.Q^T (K rows are the MFMA A operand, the wave's queries the B operand),
// so in the 16x16 accumulator layout a lane holds 16 keys of ONE query: the softmax row statistics
// are lane-local plus two cross-lane steps, the rescale factor is one scalar per lane, and the
// probabilities a lane holds are exactly its B-operand share of O^T = V^T.P^T (the k-slot <-> key
// assignment of an MFMA is free as long as both operands use the same one) -- P never touches LDS.
// T = bf16 (v_mfma_f32_16x16x32_bf16, exp2 with the scale folded into q) or f32
// (v_mfma_f32_16x16x4_f32, division by sqrt(dk) and expf as the reference computes them).
#include "common.h"
#include "kernels.h"

namespace rvb {

static constexpr int KT = 64;     // keys per tile

// FOLD (round 4, bf16 encoder form): (q + u).k + (q + v).p = (q + u).(k + p) + (v - u).p -- the second product of every
// key tile becomes a per-key constant c[head][position] = (v - u).p (input independent like P itself: a table built when the
// weights are packed), and K' = k + p is formed once per staged vector on its way into LDS.  Per 64-key tile and wave that is
// 8 MFMAs and 8 KiB of LDS fragment reads for the scores instead of 16 and 16 KiB, and no P tile in LDS.
// PADK: bytes of padding per K / P row.  A wave's ds_read_b128 is served in four groups of 16 lanes --
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- NOT in runs of 16 consecutive lanes:
// with the 16-byte pad rounds 1-3 used (row pitch 144 B = 9 bank quads) two lanes of every group meet in one quad (2-way
// conflicts on every S-phase fragment read); a pitch of 2 mod 4 quads (pad 32: 96 / 160 / 224 / 288 B rows for dk 32 / 64 /
// 96 / 128 in bf16) is conflict-free for that grouping.
template <typename T, int DKP, bool HAS_POS, int FOLD = 0, int PADK = 32>
struct AttnLds {
  static constexpr bool BF = sizeof(T) == 2;
  static constexpr int ROW_K = DKP * (int)sizeof(T) + PADK;   // K / P rows (bytes)
  // f32: V^T rows [dim][key] written transposed.  bf16: V rows [key][dim] as they come (one 16-byte store per staged
  // vector) and the transposition happens in the LDS read (ds_read_b64_tr_b16); the 32-byte pad makes the eight rows a
  // 32-lane half touches start 40 banks apart (DKP = 64): conflict-free.
  static constexpr int ROW_V = BF ? DKP * 2 + 32 : KT * (int)sizeof(T) + 16;
  static constexpr int OFF_K = 0;
  static constexpr int OFF_P = OFF_K + KT * ROW_K;
  static constexpr int OFF_V = (HAS_POS && !FOLD) ? OFF_P + KT * ROW_K : OFF_P;     // no positional keys (or folded into K): no P tile
  static constexpr int OFF_C = OFF_V + (BF ? KT : DKP) * ROW_V;          // FOLD: the (sequence, head)'s per-key constants (fp32), all of them: fold_kv_cap floats
  static constexpr int TOTAL = OFF_C;                                     // (+ 4 * fold_kv_cap bytes at launch)
};

// NW = waves per workgroup (16 queries each): 8 for the encoder and the cross attention (128-query blocks share a
// staged key tile), 1 for the decoder's self attention over a hypothesis trie (a hypothesis owns a handful of rows).
// FOLD: 0 = two products; 1 = folded, K' = k + p formed while staging; 2 (round 6) = folded, `k` already holds K' (the qkv GEMM
// wrote k + p, GemmArgs::rowadd): no positional rows are loaded, no additions -- the fold without the VALU work that made form 1 lose.
// OCC: minimum waves per SIMD asked of the register allocator (1 = whatever the kernel needs).  The folded encoder form needs 94
// VGPRs = two 8-wave workgroups per CU; OCC = 6 (lab switch RVB_ATTN_OCC=3: three workgroups per CU) squeezes it into 80 with 5 dwords
// of scratch per lane.
// MF: 16-query fragments per wave (round 6).  With MF = 2 a wave owns 32 queries: every K / V fragment it reads from LDS feeds two MFMAs,
// a tile is staged and two barriers are passed for twice the work, and the second fragment's MFMAs are independent of the first
// fragment's softmax -- the in-order wave has matrix work to issue beside its own VALU chain.
template <typename T, int DKP, bool HAS_POS, int NW, int FOLD = 0, int PADK = 32, int OCC = 1, int MF = 1>
__global__ __launch_bounds__(64 * NW, OCC) void attn_kernel(AttnArgs a) {
  static_assert(!FOLD || (HAS_POS && sizeof(T) == 2), "the folded positional term is built for the bf16 encoder form");
  static_assert(MF == 1 || FOLD == 2, "two fragments per wave: built for the prefolded encoder form");
  constexpr int QT = 16 * NW * MF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = AttnLds<T, DKP, HAS_POS, FOLD, PADK>;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int VE = Mma16<T>::VE;
  constexpr int KC = Mma16<T>::KC;
  constexpr int NCH = DKP / KC;     // K-chunks of the q.k contraction
  constexpr int NKC = KT / KC;      // K-chunks (of keys) of the p.v contraction
  constexpr int NOF = DKP / 16;     // output fragments along dk
  constexpr int VPR = DKP / VE;     // 16-byte vectors per staged row
  constexpr int NT = 64 * NW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // block -> (sequence, first query): the grid's (x, z), or one entry of a host-built work list (ragged batches of
  // short sequences: no empty blocks)
  // Grid form: the query blocks of one (sequence, head) read the same K / P / V rows, and workgroups go to the 8 XCDs round
  // robin in launch order (x fastest) -- left alone, the four 128-query blocks of an encoder chunk land on four XCDs and
  // each L2 fetches the chunk's keys and values for itself (round 3: 32 GB fetched per hour of audio against 10 GB
  // algorithmic).  Remapped so that all gridDim.x blocks of a (sequence, head) run on ONE XCD, back to back: launch-order id
  // L -> XCD L % 8, slot j = L / 8 on it -> group (j / gx) * 8 + xcd, query block j % gx; the ids past the last whole round of
  // 8 groups keep the plain order (a bijection either way).
  int head = blockIdx.y, seq, q0;
  if (a.work) {
    seq = a.work[2 * blockIdx.x];
    q0 = a.work[2 * blockIdx.x + 1];
  } else {
    const int gx = gridDim.x, gy = gridDim.y;
    const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int total = gx * gy * (int)gridDim.z, full = total / (8 * gx) * (8 * gx);
    int group, qb;
    if (L < full && !a.plain_order) {
      const int xcd = L & 7, j = L >> 3;
      group = (j / gx) * 8 + xcd;
      qb = j - (j / gx) * gx;
    } else {
      group = L / gx;
      qb = L - group * gx;
    }
    seq = group / gy;
    head = group - seq * gy;
    q0 = qb * QT;
  }
  const int qlen = a.q_len[seq];
  if (q0 >= qlen) return;                       // block-uniform
  const int qs = a.q_start[seq], ks = a.kv_start[seq], kvlen = a.kv_len[seq];
  // position of the sequence's first query among its keys (causal mask): 0 when queries and keys are the same rows;
  // a hypothesis that shares its first d tokens with an earlier one only computes rows d.. (rescoring trie)
  const int pos0 = a.q_pos0 ? a.q_pos0[seq] : 0;
  const int* __restrict__ kvi = (MF == 1 && a.kv_index) ? a.kv_index + ks : nullptr;   // key j lives in row kvi[j] instead of ks + j (MF = 2: the encoder's form, no index list)
  const int dk = a.dk;
  const T* Q = (const T*)a.q;
  const T* K = (const T*)a.k;
  const T* V = (const T*)a.v;
  const T* P = (const T*)a.p;

  const int lrow = lane & 15;          // operand row inside a fragment; in C layout: the column
  const int lgrp = lane >> 4;          // 16-byte vector of a 64-byte chunk; in C layout: rows 4*lgrp..+3
  int my_q[MF];                                   // the queries this lane's accumulator columns belong to
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) my_q[mf] = q0 + (wave * MF + mf) * 16 + lrow;

  // bf16: exp(x/sqrt(dk)) = exp2(x * log2(e)/sqrt(dk)); the factor is folded into the q operands
  const float qscale = BF ? 1.44269504f / a.sqrt_dk : 1.0f;

  // ---- query operands: this wave's 16 queries, biased copies (rows >= qlen are zero) ----
  uint4 qu[MF][NCH], qv[HAS_POS ? NCH : 1];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const bool rok = my_q[mf] < qlen;
    const T* qp = Q + (size_t)(qs + (rok ? my_q[mf] : 0)) * a.q_stride + head * dk;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int e0 = ch * KC + lgrp * VE;
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (rok && e0 < dk) raw = *(const uint4*)(qp + e0);
      if (a.bias_u != nullptr || BF) {
        T tmp[VE], ou[VE], ov[VE];
        *(uint4*)tmp = raw;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const bool ok = rok && (e0 + e) < dk;
          const float qf = Cvt<T>::to_f32(tmp[e]);
          const float bu = a.bias_u ? a.bias_u[head * dk + (ok ? e0 + e : 0)] : 0.f;
          const float bvv = a.bias_v ? a.bias_v[head * dk + (ok ? e0 + e : 0)] : 0.f;
          ou[e] = Cvt<T>::from_f32(ok ? (qf + bu) * qscale : 0.f);
          ov[e] = Cvt<T>::from_f32(ok ? (qf + bvv) * qscale : 0.f);
        }
        qu[mf][ch] = *(uint4*)ou;
        if constexpr (HAS_POS && !FOLD) qv[ch] = *(uint4*)ov;
      } else {
        qu[mf][ch] = raw;
        if constexpr (HAS_POS && !FOLD) qv[ch] = raw;
      }
    }
  }

  f32x4_t o[MF][NOF];                  // O^T: lane holds dims 16f + 4*lgrp + r of its query
  float m_run[MF], l_run[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
    for (int f = 0; f < NOF; ++f) o[mf][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    m_run[mf] = -INFINITY; l_run[mf] = 0.f;
  }

  char* sK = smem + L::OFF_K;
  char* sP = smem + L::OFF_P;
  char* sV = smem + L::OFF_V;

  int kend = kvlen, kbeg = 0;
  if (a.causal) kend = min(kvlen, pos0 + q0 + QT);
  // streaming-style chunk mask (utils/mask.py:86-123 subsequent_chunk_mask): query i sees keys
  // [max((i/cs - left)*cs, 0), (i/cs + 1)*cs); the tile loop covers the union over this block's queries
  const int cs = a.chunk;
  if (cs > 0) {
    const int qlast = min(q0 + QT, qlen) - 1;
    kend = min(kend, (qlast / cs + 1) * cs);
    if (a.left >= 0) kbeg = (max((q0 / cs - a.left) * cs, 0) / KT) * KT;
  }
  // Staging: every thread fetches one 16-byte vector of K, P and V per tile (coalesced: 8 lanes per
  // 128-byte row), holds tile t+1 in VGPRs while tile t is multiplied, and writes it to LDS at the
  // top of the next iteration.  K and P rows are stored as they come.  V is stored TRANSPOSED
  // (V^T[dim][key]); the 8 lanes that share a key write to 8 rows 8 dims apart, which would be one
  // bank, so whole groups of 4 keys are XOR-swizzled by the row's 8-dim block: key' = key ^ (((dim/VE)&7)<<2).
  constexpr int NV = (KT * VPR + NT - 1) / NT;
  uint4 rk[NV], rp[HAS_POS ? NV : 1], rv[NV];
  // FOLD: the head's per-key constants (v - u).p_j, ALL fold_kv_cap of them, go into LDS once per workgroup (round 6).  Until then
  // 16 threads fetched the tile's 64 constants tile by tile -- `key + r < kvlen ? Cb[key + r] : 0`, which hipcc compiled into four
  // exec-masked branches whose loads reused one address pair: `s_waitcnt vmcnt(0)` between them, four serialised memory round trips
  // in wave 0 of EVERY tile with the whole workgroup waiting at the barrier behind it (that, not the arithmetic of the fold, is why
  // round 4 measured the fold slower).  Keys past kvlen are masked to -inf below whatever constant they carry.
  char* sC = smem + L::OFF_C;
  if constexpr (FOLD) {
    const float* __restrict__ Cb = a.pos_bias + (size_t)head * a.pos_bias_stride;
    const int last = a.pos_bias_stride - 1;
    for (int j = tid; j < a.fold_kv_cap; j += NT) ((float*)sC)[j] = Cb[min(j, last)];
  }
  auto gload = [&](int kt0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + NT * n;
      const int r = i / VPR, c = i - r * VPR;
      const int key = kt0 + r;
      const bool ok = i < KT * VPR && key < kvlen && c * VE < dk;
      rk[n] = make_uint4(0, 0, 0, 0); rv[n] = make_uint4(0, 0, 0, 0);
      if constexpr (HAS_POS && FOLD != 2) rp[n] = make_uint4(0, 0, 0, 0);
      if (ok) {
        const size_t krow = kvi ? (size_t)kvi[key] : (size_t)(ks + key);
        rk[n] = *(const uint4*)(K + krow * a.k_stride + head * dk + c * VE);
        rv[n] = *(const uint4*)(V + krow * a.v_stride + head * dk + c * VE);
        if constexpr (HAS_POS && FOLD != 2) rp[n] = *(const uint4*)(P + (size_t)key * a.p_stride + head * dk + c * VE);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + NT * n;
      if (i < KT * VPR) {
        const int r = i / VPR, c = i - r * VPR;
        if constexpr (FOLD == 1) {
          // K' = k + p: eight bf16 sums in fp32, rounded once
          const bf16_t* kk = (const bf16_t*)&rk[n];
          const bf16_t* pp = (const bf16_t*)&rp[n];
          uint4 o;
          o.x = pack2_bf16(bf16_to_f32(kk[0]) + bf16_to_f32(pp[0]), bf16_to_f32(kk[1]) + bf16_to_f32(pp[1]));
          o.y = pack2_bf16(bf16_to_f32(kk[2]) + bf16_to_f32(pp[2]), bf16_to_f32(kk[3]) + bf16_to_f32(pp[3]));
          o.z = pack2_bf16(bf16_to_f32(kk[4]) + bf16_to_f32(pp[4]), bf16_to_f32(kk[5]) + bf16_to_f32(pp[5]));
          o.w = pack2_bf16(bf16_to_f32(kk[6]) + bf16_to_f32(pp[6]), bf16_to_f32(kk[7]) + bf16_to_f32(pp[7]));
          *(uint4*)(sK + r * L::ROW_K + c * 16) = o;
        } else {
          *(uint4*)(sK + r * L::ROW_K + c * 16) = rk[n];
          if constexpr (HAS_POS && !FOLD) *(uint4*)(sP + r * L::ROW_K + c * 16) = rp[n];
        }
        if constexpr (BF) {
          *(uint4*)(sV + r * L::ROW_V + c * 16) = rv[n];
        } else {
          T tv[VE];
          *(uint4*)tv = rv[n];
          const int kcol = r ^ ((c & 7) << 2);
#pragma unroll
          for (int e = 0; e < VE; ++e) *(T*)(sV + (c * VE + e) * L::ROW_V + kcol * sizeof(T)) = tv[e];
        }
      }
    }
  };
  if (kend > kbeg) gload(kbeg);
  for (int kt0 = kbeg; kt0 < kend; kt0 += KT) {
    lstore();
    __syncthreads();
    if (kt0 + KT < kend) gload(kt0 + KT);       // in flight under the MFMAs / softmax below

    // ---- S^T = K.Qu^T (+ P.Qv^T): fragment nf holds keys 16nf + 4*lgrp + r of query lrow ----
    f32x4_t s[MF][4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if constexpr (FOLD) {         // the accumulation starts from the keys' constants: fragment nf = keys 16 nf + 4 lgrp + r
          const float4 c4 = *(const float4*)(sC + (kt0 + nf * 16 + lgrp * 4) * 4);
          s[mf][nf] = (f32x4_t){c4.x, c4.y, c4.z, c4.w};
        } else {
          s[mf][nf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const uint4 ak = *(const uint4*)(sK + (nf * 16 + lrow) * L::ROW_K + ch * 64 + lgrp * 16);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) Mma16<T>::run(ak, qu[mf][ch], s[mf][nf]);
        if constexpr (HAS_POS && !FOLD) {
          const uint4 ap = *(const uint4*)(sP + (nf * 16 + lrow) * L::ROW_K + ch * 64 + lgrp * 16);
          Mma16<T>::run(ap, qv[ch], s[0][nf]);
        }
      }
    }
    // ---- scale (f32 mode), mask (only tiles that need it), online softmax of this lane's queries ----
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if constexpr (!BF) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[mf][nf][r] = s[mf][nf][r] / a.sqrt_dk;
      }
      if (a.causal || cs > 0 || kt0 + KT > kvlen) {          // block-uniform
        int lo = 0, hi = kvlen;
        if (cs > 0) {
          const int ci = my_q[mf] / cs;
          hi = min(kvlen, (ci + 1) * cs);
          if (a.left >= 0) lo = max((ci - a.left) * cs, 0);
        }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt0 + nf * 16 + lgrp * 4 + r;
            if (key >= hi || key < lo || (a.causal && key > pos0 + my_q[mf])) s[mf][nf][r] = -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[mf][nf][r]);
      mx = rows_max(mx);
      const float m_new = fmaxf(m_run[mf], mx);
      // Branch-free since round 4: while every key so far is masked for this query (m_new = -inf) the reference point is 0, so that
      // exp(-inf - 0) = 0 gives the zeros the old "all masked" branch wrote by hand -- hipcc had turned that branch into 18
      // register clears + an exec-masked block that EVERY tile executed.  al = exp(-inf) = 0 instead of 1 there, on o = l = 0.
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      float al, ps;
      if constexpr (BF) {
        // pairs: the subtraction and the running sum are packed (v_pk_add_f32); the sum's order differs from the scalar loop's
        // in the last bit at most (bf16 engine only: its probabilities are rounded to bf16 right below)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        al = __builtin_amdgcn_exp2f(m_run[mf] - m_use);
        const f32x2_t mm = {m_use, m_use};
        f32x2_t acc = {0.f, 0.f};
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2_t dd = (f32x2_t){s[mf][nf][r], s[mf][nf][r + 1]} - mm;
            const f32x2_t pp = {__builtin_amdgcn_exp2f(dd.x), __builtin_amdgcn_exp2f(dd.y)};
            s[mf][nf][r] = pp.x; s[mf][nf][r + 1] = pp.y;
            acc += pp;
          }
        ps = acc.x + acc.y;
      } else {
        al = expf(m_run[mf] - m_use);
        ps = 0.f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float pv = expf(s[mf][nf][r] - m_use); s[mf][nf][r] = pv; ps += pv; }
      }
      ps = rows_sum(ps);
      l_run[mf] = l_run[mf] * al + ps;
      m_run[mf] = m_new;
#pragma unroll
      for (int f = 0; f < NOF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[mf][f][r] *= al;
    }

    // ---- O^T += V^T.P^T : the lane's own probabilities are its B operand; the k slot (lgrp, e) of a
    //      key chunk is assigned key 16nf + 4*lgrp + r, and V^T is read with the same assignment ----
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      uint4 pb[MF];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if constexpr (BF) {
          const f32x4_t& x0 = s[mf][2 * kc];
          const f32x4_t& x1 = s[mf][2 * kc + 1];
          pb[mf].x = pack2_bf16(x0[0], x0[1]);
          pb[mf].y = pack2_bf16(x0[2], x0[3]);
          pb[mf].z = pack2_bf16(x1[0], x1[1]);
          pb[mf].w = pack2_bf16(x1[2], x1[3]);
        } else {
          const f32x4_t& x0 = s[mf][kc];
          pb[mf] = make_uint4(__float_as_uint(x0[0]), __float_as_uint(x0[1]), __float_as_uint(x0[2]), __float_as_uint(x0[3]));
        }
      }
#pragma unroll
      for (int f = 0; f < NOF; ++f) {
        const int dim = f * 16 + lrow;
        const char* vrow = sV + dim * L::ROW_V;
        const int swz = ((dim / VE) & 7) << 2;          // same key-group swizzle as the transposed store
        uint4 va;
        if constexpr (BF) {
          // transpose read: the 16 lanes of a group pass the addresses of the 8-byte pieces of a [4 keys][16 dims] block
          // (lane i: key i>>2, dims 4*(i&3)..+3) and lane i receives column i = dim 16f + lrow of the four keys -- its
          // A-operand share for k slots (lgrp, 0..3); a second read 16 keys on fills slots (lgrp, 4..7)
          typedef __attribute__((ext_vector_type(4))) short s16x4_t;
          const char* vb = sV + (32 * kc + 4 * lgrp + (lrow >> 2)) * L::ROW_V + (f * 16 + 4 * (lrow & 3)) * 2;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)vb);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + 16 * L::ROW_V));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          va = make_uint4(l2.x, l2.y, h2.x, h2.y);
        } else {
          va = *(const uint4*)(vrow + ((16 * kc + 4 * lgrp) ^ swz) * 4);
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) Mma16<T>::run(va, pb[mf], o[mf][f]);
      }
    }
    __syncthreads();
  }

  // ---- normalise and store: 4 consecutive dims per fragment ----
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    if (my_q[mf] >= qlen) continue;
    const float inv = l_run[mf] > 0.f ? 1.0f / l_run[mf] : 0.f;
    T* orow = (T*)a.out + (size_t)(qs + my_q[mf]) * a.o_stride + head * dk;
#pragma unroll
    for (int f = 0; f < NOF; ++f) {
      const int c0 = f * 16 + lgrp * 4;
      if (c0 + 4 <= dk && ((a.o_stride * (int)sizeof(T)) % (BF ? 8 : 16)) == 0) {
        if constexpr (BF) {
          uint2 pk;
          pk.x = pack2_bf16(o[mf][f][0] * inv, o[mf][f][1] * inv);
          pk.y = pack2_bf16(o[mf][f][2] * inv, o[mf][f][3] * inv);
          *(uint2*)(orow + c0) = pk;
        } else {
          *(float4*)(orow + c0) = make_float4(o[mf][f][0] * inv, o[mf][f][1] * inv, o[mf][f][2] * inv, o[mf][f][3] * inv);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c0 + r < dk) orow[c0 + r] = Cvt<T>::from_f32(o[mf][f][r] * inv);
      }
    }
  }
}

template <typename T, int DKP, bool HAS_POS, int NW, int FOLD = 0, int PADK = 32, int OCC = 1, int MF = 1>
static int launch_attn(hipStream_t s, const AttnArgs& a) {
  using L = AttnLds<T, DKP, HAS_POS, FOLD, PADK>;
  static bool attr_set = false;
  auto kern = attn_kernel<T, DKP, HAS_POS, NW, FOLD, PADK, OCC, MF>;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL + (FOLD ? 64 * 1024 : 0)));
    attr_set = true;
  }
  dim3 grid(cdiv(a.max_q, 16 * NW * MF), a.heads, a.nseq);
  if (a.work) grid = dim3(a.n_work, a.heads, 1);
  if (grid.x == 0) return OK;
  const int fold_bytes = FOLD ? a.fold_kv_cap * 4 : 0;
  if (FOLD && (a.fold_kv_cap <= 0 || a.fold_kv_cap % 64 || fold_bytes > 64 * 1024)) { set_error("attention: folded form needs fold_kv_cap = a multiple of 64 keys, at most 16384"); return E_ARG; }
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), L::TOTAL + fold_bytes, s, a);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

template <typename T>
static int dispatch_attn(hipStream_t s, const AttnArgs& a) {
  const bool pos = a.p != nullptr;
  const int dk = a.dk;
  if constexpr (sizeof(T) == 2) {      // the folded positional term: bf16, dk <= 64, 128-query workgroups (the encoder's form)
    if (pos && a.pos_bias != nullptr && a.fold_kv_cap > 0 && dk <= 64 && dk > 32 && (a.q_block == 0 || a.q_block == 128))
    {
      static const int occ = lab_env("RVB_ATTN_OCC") ? atoi(lab_env("RVB_ATTN_OCC")) : 2;      // workgroups per CU asked for (lab A/B)
      if (a.k_prefolded && occ == 3) return launch_attn<T, 64, true, 8, 2, 32, 6>(s, a);
      // Two 16-query fragments per wave, 256 queries per workgroup, held to 128 VGPRs = two workgroups per CU (round 6; lab switch
      // RVB_ATTN_MF=0 = one fragment per wave): 8.13 -> 7.25 ms per hour of audio.  Measured beside it and dropped
      // (profiles/r06_call11_*, r06_call13_*): the same with 4 waves / 128 queries (149 VGPRs: 10.2 ms), with 8 waves at the
      // allocator's own 133 VGPRs (one workgroup per CU: 9.2 ms), with 16 waves / 512 queries (8.17 ms), and softmax + P.V fragment by
      // fragment in one basic block (8.55 ms: hipcc does not interleave the two chains and the V fragments are read twice).
      static const int mfv = lab_env("RVB_ATTN_MF") ? atoi(lab_env("RVB_ATTN_MF")) : 1;
      if (a.k_prefolded && a.work == nullptr && a.kv_index == nullptr && a.q_block == 0 && a.max_q > 128 && mfv != 0) return launch_attn<T, 64, true, 8, 2, 32, 4, 2>(s, a);
      return a.k_prefolded ? launch_attn<T, 64, true, 8, 2>(s, a) : launch_attn<T, 64, true, 8, 1>(s, a);
    }
    // A/B switch for the encoder's form (dk 33..64, 128-query workgroups, positional keys): RVB_ATTN_PADK=16 = the 144-byte row
    // pitch of rounds 1-3 (2-way bank conflicts on every S-phase fragment read: 10.41 vs 9.99 ms per hour, SQ_LDS_BANK_CONFLICT
    // 2.1e8 vs 0 on a quarter hour, profiles/r04_call7_attention_padk_pmc.txt); every other form uses the 32-byte pad
    static const int padk = lab_env("RVB_ATTN_PADK") ? atoi(lab_env("RVB_ATTN_PADK")) : 32;
    if (pos && padk == 16 && dk <= 64 && dk > 32 && (a.q_block == 0 || a.q_block == 128)) return launch_attn<T, 64, true, 8, 0, 16>(s, a);
  }
#define RVB_ATTN_CASE(D)                                                           \
  if (dk <= D) {                                                                   \
    if (a.q_block == 16) return launch_attn<T, D, false, 1>(s, a);                 \
    if (a.q_block == 64) return pos ? launch_attn<T, D, true, 4>(s, a) : launch_attn<T, D, false, 4>(s, a); \
    if (a.q_block == 256) {                                                        \
      if constexpr (sizeof(T) == 2 && D <= 64) return pos ? launch_attn<T, D, true, 16>(s, a) : launch_attn<T, D, false, 16>(s, a); \
      set_error("attention: 256-query workgroups are built for bf16 with dk <= 64 (128 VGPRs per wave)"); \
      return E_UNSUPPORTED;                                                        \
    }                                                                              \
    return pos ? launch_attn<T, D, true, 8>(s, a) : launch_attn<T, D, false, 8>(s, a); \
  }
  RVB_ATTN_CASE(32)
  RVB_ATTN_CASE(64)
  RVB_ATTN_CASE(96)
  RVB_ATTN_CASE(128)
#undef RVB_ATTN_CASE
  set_error("attention: head dim > 128 unsupported");
  return E_UNSUPPORTED;
}

// c[h][j] = sum_e (v[h][e] - u[h][e]) * P[j][h*dk + e] * scale, P in bf16 as the kernel reads it; one thread per (h, j)
__global__ __launch_bounds__(256) void pos_bias_kernel(const bf16_t* __restrict__ P, int rows, int p_stride, const float* __restrict__ bu,
                                                       const float* __restrict__ bv, int heads, int dk, float scale, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= heads * rows) return;
  const int h = idx / rows, j = idx - h * rows;
  const bf16_t* pr = P + (size_t)j * p_stride + h * dk;
  float acc = 0.f;
  for (int e = 0; e < dk; ++e) acc += (bv[h * dk + e] - bu[h * dk + e]) * bf16_to_f32(pr[e]);
  out[idx] = acc * scale;
}
int attention_pos_bias(hipStream_t s, const void* P, int rows, int p_stride, const float* bias_u, const float* bias_v, int heads, int dk,
                       float scale, float* out) {
  if (rows <= 0 || heads <= 0) return OK;
  hipLaunchKernelGGL(pos_bias_kernel, dim3(cdiv(heads * rows, 256)), dim3(256), 0, s, (const bf16_t*)P, rows, p_stride, bias_u, bias_v, heads, dk,
                     scale, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int attention(hipStream_t s, int dtype, const AttnArgs& a0) {
  if (a0.nseq <= 0 || a0.max_q <= 0) return OK;
  static const int plain = lab_env("RVB_ATTN_PLAIN") ? atoi(lab_env("RVB_ATTN_PLAIN")) : 0;      // tuning: A/B of the block order
  AttnArgs a = a0;
  if (plain) a.plain_order = 1;
  const int ve = dtype == DT_BF16 ? 8 : 4;
  if (a.dk % ve || a.q_stride % ve || a.k_stride % ve || a.v_stride % ve || (a.p && a.p_stride % ve)) {
    set_error("attention: dk and row strides must be multiples of the 16-byte vector width");
    return E_ARG;
  }
  if (a.k_prefolded && !(dtype == DT_BF16 && a.p && a.pos_bias && a.fold_kv_cap > 0 && a.dk <= 64 && a.dk > 32 && (a.q_block == 0 || a.q_block == 128))) {
    set_error("attention: pre-folded keys (k + p) are read by the folded bf16 encoder form only (33 <= dk <= 64, 128-query blocks, pos_bias table)");
    return E_ARG;
  }
  if ((a.bias_u == nullptr) != (a.bias_v == nullptr)) { set_error("attention: bias_u/bias_v must come together"); return E_ARG; }
  if (a.q_block != 0 && a.q_block != 16 && a.q_block != 64 && a.q_block != 128 && a.q_block != 256) { set_error("attention: q_block must be 0 (= 128), 16, 64, 128 or 256"); return E_ARG; }
  if (a.q_block == 16 && a.p) { set_error("attention: 16-query blocks are built for the decoder forms (no positional keys)"); return E_ARG; }
  return dtype == DT_BF16 ? dispatch_attn<bf16_t>(s, a) : dispatch_attn<float>(s, a);
}

}  // namespace rvb
