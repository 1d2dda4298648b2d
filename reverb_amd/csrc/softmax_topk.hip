// Row log-softmax kernels over fp32 logits (HBM-bound): CTC log-softmax + per-frame top-k, and
// log-softmax + gather of one target per row for attention rescoring.
//   logsoftmax_topk   asr/wenet/transformer/ctc.py:106-114, asr_model.py:318-329 (blank penalty),
//                     search.py:111,155 (torch.topk per frame)
//   lse_gather        asr_model.py:969 + search.py:417-437 (only the needed log-probs)
// One wave64 per row, 16-byte loads, four vectors per lane in flight.
//   pass A : online max / sum-exp per lane (merged across the wave at the end) and the maximum of
//            each lane's slice; the k-th largest of the 64 lane maxima is a lower bound t of the
//            row's k-th largest logit (they are k distinct elements >= t).
//   pass B : the row is read again (40 KB, L2 resident) and the few elements >= t are compacted into
//            a per-wave LDS list with ballot/popcount.
//   pass C : k rounds of wave arg-max over the candidates; order = value descending, ties by lower
//            index (a stable descending order, as torch.topk on CPU returns it).
// Rows with more than 256 candidates (massive ties) fall back to k exclusion scans.
#include "common.h"
#include "kernels.h"

namespace rvb {

static constexpr int TOPK_MAX = 64;      // one lane maximum per kept element bounds the threshold search (pass A): k <= 64
static constexpr int UNR = 4;        // float4 vectors per lane per batch
static constexpr int CAND = 256;     // candidate slots per row

struct RowStat {
  float m = -INFINITY, s = 0.f;
  __device__ inline void add_batch(const float* x, int n) {   // n <= 16 values, any may be -inf
    float bm = x[0];
#pragma unroll
    for (int i = 1; i < 4 * UNR; ++i) if (i < n) bm = fmaxf(bm, x[i]);
    if (bm > m) { s *= expf(m - bm); m = bm; }     // first time m = -inf: s = 0 * exp(-inf) = 0
    if (m == -INFINITY) return;
#pragma unroll
    for (int i = 0; i < 4 * UNR; ++i) if (i < n) s += expf(x[i] - m);
  }
  __device__ inline float wave_lse() const {
    const float M = wave_max(m);
    const float part = (m == -INFINITY) ? 0.f : s * expf(m - M);
    return M + logf(wave_sum(part));
  }
};

// (value desc, index asc) ordering
__device__ inline bool better(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

__device__ inline void wave_argbest(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (better(ov, oi, v, i)) { v = ov; i = oi; }
  }
}

struct RowReader {
  const float* x; int V, nvec, blank; float pen;
  // loads batch `b` (1024 logits of the row) into e[16]; returns number of valid leading... all 16
  // slots are filled, out-of-range ones with -inf; idx(slot) gives the logit index of a slot
  __device__ inline void load(int v0, int lane, float* e) const {
    float4 q[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int vi = v0 + u * 64 + lane;
      q[u] = vi < nvec ? ((const float4*)x)[vi] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { e[4 * u] = q[u].x; e[4 * u + 1] = q[u].y; e[4 * u + 2] = q[u].z; e[4 * u + 3] = q[u].w; }
    if (pen != 0.f) {
#pragma unroll
      for (int s = 0; s < 4 * UNR; ++s) if (idx(v0, lane, s) == blank) e[s] -= pen;
    }
  }
  __device__ inline int idx(int v0, int lane, int slot) const { return (v0 + (slot >> 2) * 64 + lane) * 4 + (slot & 3); }
  __device__ inline float tail(int i) const { float v = x[i]; if (i == blank) v -= pen; return v; }
};

template <bool TOPK>
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ logits, int M, int V, int ld, int k,
                                                      float pen, int blank, float* __restrict__ tv,
                                                      int* __restrict__ ti, float* __restrict__ lp,
                                                      const int* __restrict__ target, float* __restrict__ gathered) {
  __shared__ float s_cv[4][TOPK ? CAND : 1];
  __shared__ int s_ci[4][TOPK ? CAND : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= M) return;
  RowReader rd;
  rd.x = logits + (size_t)row * ld; rd.V = V; rd.blank = blank; rd.pen = pen;
  const bool vec = ((ld & 3) == 0) && (((size_t)logits & 15) == 0);
  rd.nvec = vec ? (V >> 2) : 0;
  const int tail0 = rd.nvec * 4;

  // ---- pass A ----
  RowStat st;
  float lane_max = -INFINITY;
  for (int v0 = 0; v0 < rd.nvec; v0 += 64 * UNR) {
    float e[4 * UNR];
    rd.load(v0, lane, e);
    st.add_batch(e, 4 * UNR);
    if constexpr (TOPK) {
#pragma unroll
      for (int s = 0; s < 4 * UNR; ++s) lane_max = fmaxf(lane_max, e[s]);
    }
  }
  for (int i = tail0 + lane; i < V; i += 64) {
    float one[4 * UNR];
    one[0] = rd.tail(i);
    st.add_batch(one, 1);
    lane_max = fmaxf(lane_max, one[0]);
  }
  const float lse = st.wave_lse();
  if constexpr (!TOPK) {
    if (k < 0) {      // CSR form: row r owns targets [ti[r], ti[r+1]) of `target`, results land at the same positions
      for (int p = ti[row] + lane; p < ti[row + 1]; p += 64) gathered[p] = rd.tail(target[p]) - lse;
    } else if (lane == 0) {
      gathered[row] = rd.tail(target[row]) - lse;       // (with a blank penalty: the penalised logit, as ctc_logprobs forms it)
    }
    return;
  } else {
    if (lp) {
      float* o = lp + (size_t)row * V;
      for (int i = lane; i < V; i += 64) o[i] = rd.tail(i) - lse;
    }
    // threshold: k-th largest lane maximum
    float thr;
    {
      float h = lane_max;
      float cur = 0.f;
      for (int r = 0; r < k; ++r) {
        cur = wave_max(h);
        const unsigned long long mk = __ballot(h == cur);
        if (lane == (int)__builtin_ctzll(mk)) h = -INFINITY;   // retire one holder of the maximum
      }
      thr = cur;
    }
    // ---- pass B: compact candidates (>= thr) ----
    int count = 0;   // wave-uniform
    auto push = [&](bool pred, float v, int i) {
      const unsigned long long mk = __ballot(pred);
      if (mk == 0) return;
      const int pos = count + (int)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
      if (pred && pos < CAND) { s_cv[w][pos] = v; s_ci[w][pos] = i; }
      count += (int)__builtin_popcountll(mk);
    };
    for (int v0 = 0; v0 < rd.nvec; v0 += 64 * UNR) {
      float e[4 * UNR];
      rd.load(v0, lane, e);
#pragma unroll
      for (int s = 0; s < 4 * UNR; ++s)
        push((v0 + (s >> 2) * 64 + lane) < rd.nvec && e[s] >= thr, e[s], rd.idx(v0, lane, s));
    }
    for (int i0 = tail0; i0 < V; i0 += 64) {
      const int i = i0 + lane;
      const float v = i < V ? rd.tail(i) : -INFINITY;
      push(i < V && v >= thr, v, i);
    }
    __builtin_amdgcn_wave_barrier();
    if (count <= CAND) {
      // ---- pass C: k arg-max rounds over <= 256 candidates (4 per lane) ----
      float cv[4]; int ci[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = lane + 64 * j;
        cv[j] = p < count ? s_cv[w][p] : -INFINITY;
        ci[j] = p < count ? s_ci[w][p] : 0x7fffffff;
      }
      for (int r = 0; r < k; ++r) {
        float bv = cv[0]; int bi = ci[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) if (better(cv[j], ci[j], bv, bi)) { bv = cv[j]; bi = ci[j]; }
        wave_argbest(bv, bi);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (ci[j] == bi) { cv[j] = -INFINITY; ci[j] = 0x7fffffff; }
        if (lane == 0) { tv[(size_t)row * k + r] = bv - lse; ti[(size_t)row * k + r] = bi; }
      }
    } else {
      // ---- fallback (massive ties): k exclusion scans of the whole row ----
      float pv = INFINITY; int pi = -1;
      for (int r = 0; r < k; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < V; i += 64) {
          const float v = rd.tail(i);
          const bool after_prev = v < pv || (v == pv && i > pi);
          if (after_prev && better(v, i, bv, bi)) { bv = v; bi = i; }
        }
        wave_argbest(bv, bi);
        pv = bv; pi = bi;
        if (lane == 0) { tv[(size_t)row * k + r] = bv - lse; ti[(size_t)row * k + r] = bi; }
      }
    }
  }
}

int logsoftmax_topk(hipStream_t s, const float* logits, int M, int V, int ld, int k, float blank_penalty,
                    int blank_id, float* topk_val, int* topk_idx, float* logp_out) {
  if (M <= 0) return OK;
  if (k < 1 || k > TOPK_MAX || k > V) { set_error("logsoftmax_topk: beam must be in [1,64] and <= vocab"); return E_ARG; }
  hipLaunchKernelGGL(row_lse_kernel<true>, dim3(cdiv(M, 4)), dim3(256), 0, s, logits, M, V, ld, k, blank_penalty,
                     blank_id, topk_val, topk_idx, logp_out, (const int*)nullptr, (float*)nullptr);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int lse_gather(hipStream_t s, const float* logits, int R, int V, int ld, const int* target, float* out, float blank_penalty, int blank_id) {
  if (R <= 0) return OK;
  hipLaunchKernelGGL(row_lse_kernel<false>, dim3(cdiv(R, 4)), dim3(256), 0, s, logits, R, V, ld, 0, blank_penalty, blank_penalty != 0.f ? blank_id : -1,
                     (float*)nullptr, (int*)nullptr, (float*)nullptr, target, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// rows shared by several hypotheses (rescoring over a prefix trie): row r has targets target[ptr[r] .. ptr[r+1]) and
// out[p] = logits[r][target[p]] - logsumexp(logits[r][:V]) for each of them
int lse_gather_multi(hipStream_t s, const float* logits, int R, int V, int ld, const int* ptr, const int* target, float* out) {
  if (R <= 0) return OK;
  hipLaunchKernelGGL(row_lse_kernel<false>, dim3(cdiv(R, 4)), dim3(256), 0, s, logits, R, V, ld, -1, 0.f, -1,
                     (float*)nullptr, const_cast<int*>(ptr), (float*)nullptr, target, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// out[i] = table[row[i]][col[i]] (fp32 table with `ld` entries per row): the attention log-probs joint_decoding asks for,
// read out of the stored log-softmax rows of the decoded prefixes
__global__ __launch_bounds__(256) void gather_pairs_kernel(const float* __restrict__ table, size_t ld, const int* __restrict__ row,
                                                           const int* __restrict__ col, int n, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = table[(size_t)row[i] * ld + col[i]];
}

int gather_pairs(hipStream_t s, const float* table, size_t ld, const int* row, const int* col, int n, float* out) {
  if (n <= 0) return OK;
  hipLaunchKernelGGL(gather_pairs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, table, ld, row, col, n, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
