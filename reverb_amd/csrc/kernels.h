// Internal launcher interface of librvb's HIP kernels (one translation unit per kernel family).
// All pointers are device pointers; all launches are asynchronous on the given stream.
#pragma once
#include "common.h"

namespace rvb {

enum { DT_F32 = 0, DT_BF16 = 1 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

inline size_t dt_size(int dt) { return dt == DT_BF16 ? 2 : 4; }

// ---------------------------------------------------------------- gemm.hip
struct GemmArgs {
  const void* A;      // T [M, lda]   (or NHWC activation when conv != 0)
  const void* W;      // T [N, ldw]   K-contiguous ("weight[out][in]" as torch stores it)
  const float* bias;  // [N] or null
  const float* res;   // fp32 [M, ldres] or null
  void* C;            // T or fp32 [M, ldc]
  int M, N, K;
  int lda, ldw, ldc, ldres;
  float alpha;        // C = res + alpha * act(A.W^T + bias)
  int act;
  int out_f32;        // bf16 mode only: write fp32 instead of bf16
  int conv;           // implicit 3x3 stride-2 conv gather on A
  int cT1, cF1, cT2, cF2, cC;
};
int gemm(hipStream_t s, int dtype, const GemmArgs& a);
// gemm2.hip: 256x256 LDS-DMA kernel for large shapes (K multiple of the 128-byte step)
bool gemm2_applicable(int dtype, const GemmArgs& a);
int gemm2(hipStream_t s, int dtype, const GemmArgs& a);
extern int g_gemm_variant;   // 0 = auto, 1 = always gemm.hip kernel, 2 = gemm2.hip whenever applicable

// ---------------------------------------------------------------- fbank.hip
struct FbankTables {
  const float* window;   // [400] povey
  const float* twiddle;  // [256][2] cos,sin(-2*pi*k/512)
  const float* mel_w;    // [80][257] dense triangular weights
  const int* mel_lo;     // [80] first non-zero fft bin
  const int* mel_hi;     // [80] one past last non-zero bin
};
// pcm: int16 mono (device).  feats: fp32 [n_frames, 80] raw log-mel.
int fbank(hipStream_t s, const int16_t* pcm, int64_t n_frames, float* feats, const FbankTables& t);

// ---------------------------------------------------------------- elementwise.hip
// CMVN + Conv2d(1,d,3,stride 2) + ReLU; feats fp32 [B,T0,F0] -> out T [B,T1,F1,d] (NHWC)
int subsample_conv1(hipStream_t s, int dtype, const float* feats, const float* mean, const float* istd,
                    const float* w /*[d][9]*/, const float* b /*[d]*/, void* out, int B, int T0, int F0,
                    int d);

enum { NORM_LN = 0, NORM_AFFINE = 1 };
struct NormArgs {
  const float* x;      // fp32 [M, d]
  const float* gamma;  // [d]
  const float* beta;   // [d]
  float eps;
  int mode;            // NORM_LN: (x-mean)*rstd*gamma+beta ; NORM_AFFINE: x*gamma+beta
  int silu;            // apply SiLU after the affine
  const void* add;     // T [M, d] added after everything (nullable)
  void* out;           // T or fp32 [M, d]
  int out_f32;
  int M, d;
};
int rownorm(hipStream_t s, int dtype, const NormArgs& a);

// GLU over channel halves of G [M,2d] (T), then depthwise Conv1d (kernel K, same padding) along time
// inside each chunk of T rows; rows t >= lens[b] see glu(pw1 bias) (convolution.py:107-118).
struct GluDwArgs {
  const void* G;          // T [B*T, 2d]
  const float* pw1_bias;  // [2d]
  const float* dw_w;      // [d][K]
  const float* dw_b;      // [d]
  const int* lens;        // [B] valid rows per chunk
  float* out;             // fp32 [B*T, d]
  int B, T, d, K;
};
int glu_dwconv(hipStream_t s, int dtype, const GluDwArgs& a);

// x[row] = E[tok[row]] * scale + PE[pos[row]]   (fp32)
int embed_tokens(hipStream_t s, const float* E, const float* pe, const int* tok, const int* pos, float* out,
                 int rows, int d, float scale);

// per row of fp32 logits [M, ld] (first V entries valid): optional blank penalty, log-softmax,
// top-k (descending, ties -> lower index).  logp_out nullable: full log-probs [M, V] (ld = V).
int logsoftmax_topk(hipStream_t s, const float* logits, int M, int V, int ld, int k, float blank_penalty,
                    int blank_id, float* topk_val, int* topk_idx, float* logp_out);

// out[r] = logits[r][target[r]] - logsumexp(logits[r][:V])
int lse_gather(hipStream_t s, const float* logits, int R, int V, int ld, const int* target, float* out);

// fp32 -> T conversion copy (weight packing), n elements
int convert_f32(hipStream_t s, int dtype, const float* src, void* dst, size_t n);

// ---------------------------------------------------------------- attention.hip
struct AttnArgs {
  const void* q; const void* k; const void* v;   // T, row-major, head h at column offset h*dk
  const void* p;                                 // T positional keys [Tpos, p_stride] or null
  int q_stride, k_stride, v_stride, p_stride, o_stride;  // elements per row
  const float* bias_u; const float* bias_v;      // [heads*dk] or null
  void* out;                                     // T [rows, o_stride]
  const int* q_start; const int* q_len;          // per sequence
  const int* kv_start; const int* kv_len;        // per sequence
  int nseq, heads, dk, max_q;
  int causal;
  float sqrt_dk;   // scores are divided by this (attention.py:384,395: `/ math.sqrt(self.d_k)`)
};
int attention(hipStream_t s, int dtype, const AttnArgs& a);

}  // namespace rvb
