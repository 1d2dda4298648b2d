// Row log-softmax kernels over fp32 logits (HBM-bound): CTC log-softmax + per-frame top-k, and
// log-softmax + gather of one target per row for attention rescoring.
//   logsoftmax_topk   asr/wenet/transformer/ctc.py:106-114, asr_model.py:318-329 (blank penalty),
//                     search.py:111,155 (torch.topk per frame)
//   lse_gather        asr_model.py:969 + search.py:417-437 (only the needed log-probs)
// One wave64 per row.  The row is read ONCE: 16-byte vectors, four per lane in flight per batch
// (1024 logits per wave per batch), online max/sum-exp per lane merged across the wave at the end.
// Each lane keeps a sorted top-16 of its slice in registers (static indices only); the wave then
// pops the global maximum k times (ties -> lower index, matching a stable descending order).
#include "common.h"
#include "kernels.h"

namespace rvb {

static constexpr int TOPK_MAX = 16;
static constexpr int UNR = 4;      // float4 vectors per lane per batch

struct RowStat {
  float m = -INFINITY, s = 0.f;
  __device__ inline void add_batch(const float* x, int n) {   // n values, any may be -inf
    float bm = x[0];
#pragma unroll
    for (int i = 1; i < 4 * UNR; ++i) if (i < n) bm = fmaxf(bm, x[i]);
    if (bm > m) { s *= expf(m - bm); m = bm; }     // m = -inf first time: s = 0 * exp(-inf) = 0
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

template <bool TOPK>
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ logits, int M, int V, int ld, int k,
                                                      float pen, int blank, float* __restrict__ tv,
                                                      int* __restrict__ ti, float* __restrict__ lp,
                                                      const int* __restrict__ target, float* __restrict__ gathered) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* x = logits + (size_t)row * ld;
  float bv[TOPK ? TOPK_MAX : 1];
  int bi[TOPK ? TOPK_MAX : 1];
  if constexpr (TOPK) {
#pragma unroll
    for (int i = 0; i < TOPK_MAX; ++i) { bv[i] = -INFINITY; bi[i] = 0x7fffffff; }
  }
  RowStat st;
  const bool vec = ((ld & 3) == 0) && (((size_t)logits & 15) == 0);
  const int nvec = vec ? (V >> 2) : 0;
  for (int v0 = 0; v0 < nvec; v0 += 64 * UNR) {
    float4 q[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int vi = v0 + u * 64 + lane;
      q[u] = vi < nvec ? ((const float4*)x)[vi] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    float e[4 * UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { e[4 * u] = q[u].x; e[4 * u + 1] = q[u].y; e[4 * u + 2] = q[u].z; e[4 * u + 3] = q[u].w; }
    if (pen != 0.f) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) if ((v0 + u * 64 + lane) * 4 + c == blank) e[4 * u + c] -= pen;
    }
    st.add_batch(e, 4 * UNR);
    if constexpr (TOPK) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = e[4 * u + c];
          if (v > bv[TOPK_MAX - 1]) {
            int vi = (v0 + u * 64 + lane) * 4 + c;
#pragma unroll
            for (int j = 0; j < TOPK_MAX; ++j) {
              // strict '>' keeps an earlier equal value ahead; a lane visits its indices in
              // increasing order, so on ties the lower index stays first
              if (v > bv[j]) { const float tvv = bv[j]; const int tii = bi[j]; bv[j] = v; bi[j] = vi; v = tvv; vi = tii; }
            }
          }
        }
    }
  }
  for (int i = nvec * 4 + lane; i < V; i += 64) {      // unaligned rows / the last V % 4 logits
    float v = x[i];
    if (i == blank) v -= pen;
    float one[4 * UNR];
    one[0] = v;
    st.add_batch(one, 1);
    if constexpr (TOPK) {
      if (v > bv[TOPK_MAX - 1]) {
        int vi = i;
#pragma unroll
        for (int j = 0; j < TOPK_MAX; ++j)
          if (v > bv[j] || (v == bv[j] && vi < bi[j])) { const float tvv = bv[j]; const int tii = bi[j]; bv[j] = v; bi[j] = vi; v = tvv; vi = tii; }
      }
    }
  }
  const float lse = st.wave_lse();
  if constexpr (!TOPK) {
    if (lane == 0) gathered[row] = x[target[row]] - lse;
    return;
  } else {
    if (lp) {
      float* o = lp + (size_t)row * V;
      for (int i = lane; i < V; i += 64) o[i] = (i == blank ? x[i] - pen : x[i]) - lse;
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
      if (bi[0] == hi) {   // the unique winner pops its head
#pragma unroll
        for (int j = 0; j < TOPK_MAX - 1; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
        bv[TOPK_MAX - 1] = -INFINITY; bi[TOPK_MAX - 1] = 0x7fffffff;
      }
      if (lane == 0) { tv[(size_t)row * k + r] = hv - lse; ti[(size_t)row * k + r] = hi; }
    }
  }
}

int logsoftmax_topk(hipStream_t s, const float* logits, int M, int V, int ld, int k, float blank_penalty,
                    int blank_id, float* topk_val, int* topk_idx, float* logp_out) {
  if (M <= 0) return OK;
  if (k < 1 || k > TOPK_MAX || k > V) { set_error("logsoftmax_topk: beam must be in [1,16] and <= vocab"); return E_ARG; }
  hipLaunchKernelGGL(row_lse_kernel<true>, dim3(cdiv(M, 4)), dim3(256), 0, s, logits, M, V, ld, k, blank_penalty,
                     blank_id, topk_val, topk_idx, logp_out, (const int*)nullptr, (float*)nullptr);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int lse_gather(hipStream_t s, const float* logits, int R, int V, int ld, const int* target, float* out) {
  if (R <= 0) return OK;
  hipLaunchKernelGGL(row_lse_kernel<false>, dim3(cdiv(R, 4)), dim3(256), 0, s, logits, R, V, ld, 0, 0.f, -1,
                     (float*)nullptr, (int*)nullptr, (float*)nullptr, target, out);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
