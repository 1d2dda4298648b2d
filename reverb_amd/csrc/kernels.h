// Internal launcher interface of librvb's HIP kernels (one translation unit per kernel family).
// All pointers are device pointers; all launches are asynchronous on the given stream.
#pragma once
#include <vector>
#include "common.h"

namespace rvb {

enum { DT_F32 = 0, DT_BF16 = 1 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_LRELU = 3 /* leaky_relu, slope 0.01 (gemm.hip kernel only) */ };
// ACT_GLU (gemm2's phase-interleaved bf16 kernel only, round 6): the N output columns are (a, b) PAIRS -- column 2c = a_c, column
// 2c + 1 = b_c (the caller interleaves the weight rows) -- and what is stored is the gated value a_c * sigmoid(b_c), N / 2 columns
// per row at leading dimension ldc: the convolution module's pointwise_conv1 + GLU (convolution.py:107-111) in one kernel, half the
// output bytes.  gemm_glu_supported() says whether a problem can run this way.
enum { ACT_GLU = 4 };

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
  int group_m, prio;  // gemm2 tuning (filled in by gemm2(): tile order, wave priority); leave 0
  int res_epilogue;   // gemm2p: add the residual in the epilogue (prefetched) instead of preloading the accumulators; leave 0
  int fast_epilogue;  // gemm2p (filled in by gemm2(); RVB_GEMM2_FLAGS bit 10 turns it off): full tiles store through inline asm; leave 0
  int stagger_ticks, stagger_first;   // gemm2p (filled in by gemm2()): start delay of every second first-round workgroup, in 10-ns ticks; leave 0
  int k_serp;         // gemm2p tuning (RVB_GEMM2_FLAGS bit 7): odd waves of tiles walk K downwards (L2 reuse across waves); leave 0
  // fp8 (OCP e4m3) operands, gemm2 only: A and W are bytes, the accumulator is multiplied by a_scale * w_scale[n];
  // out_fp8: C is written as fp8 of value * out_inv_scale (saturating)
  int in_fp8, out_fp8;
  float a_scale, out_inv_scale;
  const float* w_scale;   // [N]
  unsigned* sat;          // out_fp8: += the values clipped at +-448 (device counter, nullable)
  long long* dbg;         // test hook (rvb_test_gemm_timeline): per workgroup {start, stage 0 landed, main loop done, end} in 10 ns ticks + HW ids
  // Row-periodic addend (bf16 engine, bf16 output; round 6): C[m][n] += rowadd[m % rowadd_rows][n - rowadd_col0] for the columns
  // rowadd_col0 <= n < rowadd_col0 + rowadd_cols, added in fp32 after bias / activation / alpha, before the one rounding to bf16.
  // The qkv GEMM of an encoder block uses it to write K' = k + p (p = the layer's positional keys, one row per frame of a chunk),
  // which is what the folded form of the rel-pos attention reads (attention.hip FOLD 2).  null = none.
  const void* rowadd;     // bf16 [rowadd_rows][rowadd_ld]
  int rowadd_rows, rowadd_ld, rowadd_col0, rowadd_cols;
};
int gemm(hipStream_t s, int dtype, const GemmArgs& a);
// gemm2.hip: 256x256 LDS-DMA kernel for large shapes (K multiple of the 128-byte step)
bool gemm2_applicable(int dtype, const GemmArgs& a);
bool gemm_glu_supported(int dtype, const GemmArgs& a);
int gemm2(hipStream_t s, int dtype, const GemmArgs& a);
extern int g_gemm_variant;   // 0 = auto, 1 = always gemm.hip kernel, 2 = gemm2.hip whenever applicable
extern int g_gemm2_flags, g_gemm2_group_m;   // gemm2.hip tuning switches (-1 = read RVB_GEMM2_FLAGS / RVB_GEMM2_GROUP_M or the defaults)

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
int fbank_f32(hipStream_t s, const float* wave, int64_t n_frames, float* feats, const FbankTables& t);
// torchaudio.functional.resample's kernel (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) for sample_rate -> target,
// built in fp64 on the host: ker [new][K], K = 2 * width + orig, with orig / new the rates divided by their gcd
void resample_taps(int sample_rate, int target, std::vector<float>* ker, int* orig, int* new_, int* width, int* K);
// float [n] (int16 scale) -> int16, round to nearest even, saturating
int round_to_i16(hipStream_t s, const float* x, int64_t n, int16_t* out);
// polyphase sinc resampler: ker fp32 [new][K] (K = 2*width + orig), out fp32 [n_out] at int16 scale
int resample(hipStream_t s, const int16_t* pcm, int64_t n_in, const float* ker, int orig, int new_, int width, int K, float* out,
             int64_t n_out);
int resample_f32(hipStream_t s, const float* wave, int64_t n_in, const float* ker, int orig, int new_, int width, int K, float* out,
             int64_t n_out);

// ---------------------------------------------------------------- elementwise.hip
// CMVN + Conv2d(1,d,3,stride 2) + ReLU; feats fp32 [B,T0,F0] -> out T [B,T1,F1,d] (NHWC)
int subsample_conv1(hipStream_t s, int dtype, const float* feats, const float* mean, const float* istd,
                    const float* w /*[9][d]: tap-major, conv.0.weight transposed at load*/, const float* b /*[d]*/, void* out, int B, int T0, int F0,
                    int d, float out_fp8_scale = 0.f /*> 0: out is e4m3 of value / scale*/, unsigned* amax = nullptr /*running max, float bits*/,
                    unsigned* sat = nullptr /*fp8 output: += clipped values*/);

enum { NORM_LN = 0, NORM_AFFINE = 1 };
struct NormArgs {
  const float* x;      // fp32 [M, d] (bf16 when x_bf16)
  int x_bf16 = 0;
  const float* gamma;  // [d]
  const float* beta;   // [d]
  float eps;
  int mode;            // NORM_LN: (x-mean)*rstd*gamma+beta ; NORM_AFFINE: x*gamma+beta
  int silu;            // apply SiLU after the affine
  const void* add;     // T [M, d] added after everything (nullable)
  void* out;           // T or fp32 [M, d]
  int out_f32;
  int M, d;
  // optional second stage (NORM_LN only): out2 = LayerNorm(result; gamma2, beta2, eps2) as T [M, d]
  const float* gamma2 = nullptr;
  const float* beta2 = nullptr;
  float eps2 = 0.f;
  void* out2 = nullptr;
  // fp8 (e4m3) outputs for the bf16 engine's fp8 GEMMs: bytes of value * inv_scale, saturating
  int out_fp8 = 0, out2_fp8 = 0;
  float out_inv_scale = 1.f, out2_inv_scale = 1.f;
  unsigned* sat = nullptr;     // fp8 outputs: += the values that were clipped at +-448 (device counters, nullable)
  unsigned* sat2 = nullptr;
};
int rownorm(hipStream_t s, int dtype, const NormArgs& a);

// GLU over channel halves of G [M,2d] (T), then depthwise Conv1d (kernel K) along time inside each chunk of T
// rows; rows t >= lens[b] see glu(pw1 bias) (convolution.py:107-118).  causal = 0: "same" padding, zeros outside
// [0,T) (Conv1d padding, convolution.py:58-62).  causal = 1 (lorder = K-1, convolution.py:55-57,113-121): output t
// reads frames t-K+1 .. t; frames before the chunk are the module's left context: the last `hist_rows` of them
// come from `hist` (the pointwise-conv1 outputs of the frames that preceded this chunk in the stream = the
// reference's cnn_cache pushed through pointwise_conv1, which is per frame), older ones are the reference's zero
// padding, which pointwise_conv1 + GLU turn into glu(pw1 bias).
struct GluDwArgs {
  const void* G;          // T [B*T, 2d]
  const float* pw1_bias;  // [2d]
  const float* dw_w;      // [K][d]: tap-major (depthwise_conv.weight [d,1,K] transposed at load)
  const float* dw_b;      // [d]
  const int* lens;        // [B] valid rows per chunk
  float* out;             // fp32 [B*T, d] (or bf16 when out_bf16)
  int B, T, d, K;
  int out_bf16 = 0;       // bf16 engine: the convolution-module norm that follows reads bf16 (NormArgs::x_bf16)
  int gated = 0;          // G is [B*T, d], already gated (the pointwise GEMM ran with ACT_GLU): no gate arithmetic here
  int causal = 0;
  const void* hist = nullptr;   // T [K-1][2d], row K-2 = the frame just before this chunk (causal, B = 1)
  int hist_rows = 0;            // real frames in hist (its last hist_rows rows)
};
int glu_dwconv(hipStream_t s, int dtype, const GluDwArgs& a);

// x[row] = E[tok[row]] * scale + PE[pos[row]]   (fp32)
int embed_tokens(hipStream_t s, const float* E, const float* pe, const int* tok, const int* pos, float* out,
                 int rows, int d, float scale);

// per row of fp32 logits [M, ld] (first V entries valid): optional blank penalty, log-softmax,
// top-k (descending, ties -> lower index).  logp_out nullable: full log-probs [M, V] (ld = V).
int logsoftmax_topk(hipStream_t s, const float* logits, int M, int V, int ld, int k, float blank_penalty,
                    int blank_id, float* topk_val, int* topk_idx, float* logp_out);

// out[r] = logits[r][target[r]] - logsumexp(logits[r][:V]); blank_penalty != 0: of the row with logits[blank_id] -= blank_penalty
// (ctc_logprobs, asr_model.py:318-329)
int lse_gather(hipStream_t s, const float* logits, int R, int V, int ld, const int* target, float* out, float blank_penalty = 0.f,
               int blank_id = -1);
// CSR form for rows with several targets: out[p] = logits[r][target[p]] - lse(r) for p in [ptr[r], ptr[r+1])
int lse_gather_multi(hipStream_t s, const float* logits, int R, int V, int ld, const int* ptr, const int* target, float* out);

// out[i] = table[row[i]][col[i]], fp32 table with ld entries per row
int gather_pairs(hipStream_t s, const float* table, size_t ld, const int* row, const int* col, int n, float* out);

// dst[r][t][:] = src[parent[r]][t][:], t < rows: caches [R][L][row_bytes]
int gather_cache(hipStream_t s, const void* src, void* dst, const int* parent, int R, int L, int rows, int row_bytes);

// max |x| over n elements of T (or fp32 when dtype says so) folded into *slot with atomicMax on the float bits (slot >= 0)
int amax_abs(hipStream_t s, int dtype, const void* x, size_t n, float* slot);

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
  int chunk, left;   // chunk > 0: streaming chunk mask (utils/mask.py:86-123), left < 0 = all left chunks
  float sqrt_dk;   // scores are divided by this (attention.py:384,395: `/ math.sqrt(self.d_k)`)
  // ---- ragged / shared-prefix batches (all nullable / 0) ----
  const int* kv_index;   // key j of sequence s is row kv_index[kv_start[s] + j] of k / v (instead of kv_start[s] + j)
  const int* q_pos0;     // per sequence: position of its first query among its keys (causal mask), default 0
  const int* work;       // [n_work][2] = {sequence, first query} per block instead of the (max_q / q_block, nseq) grid
  int n_work;
  int q_block;           // queries per workgroup: 0 / 128 (8 waves) or 16 (1 wave; decoder forms only)
  int plain_order;       // tuning (RVB_ATTN_PLAIN=1): keep the launch order instead of the XCD-aware (sequence, head) grouping
  // bf16 encoder form with the positional term folded (attention.hip FOLD): fp32 [heads][pos_bias_stride], entry j =
  // (pos_bias_v - pos_bias_u)[head] . p[j][head] * log2(e) / sqrt(dk) for positional row j of `p` (same row offset as `p`); null = two products
  const float* pos_bias;
  int pos_bias_stride;
  int fold_kv_cap;       // with pos_bias: keys whose constants the kernel keeps in LDS (a multiple of 64, >= every sequence's kv_len; <= 16384); 0 = do not fold
  int k_prefolded;       // with pos_bias: `k` already holds K' = k + p (written by the qkv GEMM, GemmArgs::rowadd): nothing to add while staging
};
// builds that table for positional keys P [rows, p_stride] (bf16): out fp32 [heads][rows]
int attention_pos_bias(hipStream_t s, const void* P, int rows, int p_stride, const float* bias_u, const float* bias_v, int heads, int dk,
                       float scale, float* out);
int attention(hipStream_t s, int dtype, const AttnArgs& a);

}  // namespace rvb

// ================================================================ diar.hip (pyannote segmentation path)
namespace rvb {

// int16 PCM -> float (/32768), zero-filled up to n_pad samples
int pcm_to_float(hipStream_t s, const int16_t* pcm, int64_t n, float* out, int64_t n_pad);

// Sinc band-pass bank on the raw waveform, shared by every window that overlaps a frame:
// craw[t][f] = sum_k wave[stride*t + k] * filt[f][k]        T [n_frames][nf] (fp32 accumulation; the bf16 engine stores bf16: the
// tensor is read twice by every window that covers a frame -- 10 windows -- in pool_norm), nf <= 80, ksize <= 251
int sinc_conv(hipStream_t s, int dtype, const float* wave, const float* filt, void* craw, int64_t n_frames, int nf,
              int ksize, int stride);

// per window (start = (first+w)*step samples, `len` samples): stats[w] = {mean, 1/sqrt(var+eps)} (biased var)
int window_stats(hipStream_t s, const float* wave, int64_t first, int nwin, int64_t step, int len, float eps,
                 float* stats);

// MaxPool1d(3,3) + InstanceNorm1d(affine) + LeakyReLU over the frames of each window (SincNet block tail).
struct PoolNormArgs {
  const void* x;        // T [W*rows_in, ld_in] conv output (bias applied);  first block: null
  int rows_in, ld_in;   // rows per window in x (>= frames_in)
  int frames_in;        // valid conv frames per window; pooled frames = frames_in / 3
  int C, ld_out;        // channels; output row stride (pad channels written as 0)
  const float* gamma; const float* beta; float eps;
  void* out;            // T [W*frames_out, ld_out]
  int W;
  // first block (x == null): value = | a*(craw[frame0 + t][c] - mean*fsum[c]) + b*fsum[c] |, a = wn_gamma*rstd,
  // i.e. the sinc conv of the instance-normalised window, from the shared raw conv
  const void* craw /* T */; int64_t craw_frame0; int craw_frames_per_step;
  const float* stats; const float* fsum; float wn_gamma, wn_beta;
};
int pool_norm(hipStream_t s, int dtype, const PoolNormArgs& a);
// SincNet conv layers 2 / 3 (Conv1d kernel 5, 64 padded filters) straight from the [rows + 8][cin] activation tensor (bf16; cin 80 or 64):
// out[m][0..63] = bias + W[64][5 cin] . A[m cin .. (m + 5) cin); W as pack_conv1d lays it out
int conv1d5(hipStream_t s, int dtype, const void* A, int cin, const void* W, const float* bias, void* out, int64_t M);

// One bidirectional LSTM layer's recurrence (hidden 128).  xproj T [W*T, 8H] = x.Wih^T + b_ih + b_hh, per direction (forward | reverse)
// 4H columns in the kernel's read order: gate q (i, f, g, o) of hidden unit 32 v + 16 hf + c is column 128 v + 8 c + 2 q + hf
// (diar_engine.hip permutes W_ih / the biases accordingly); whh T [2][4H][H]; out T [W*T, 2H].
int lstm_recurrence(hipStream_t s, int dtype, const void* xproj, const void* whh, void* out, int W, int T);

// logp[r][:] = log_softmax(x[r].Wc^T + bc)   x T [M, ldx], Wc fp32 [C][in], C <= 16; cls[r] = argmax (nullable)
int classifier_logsoftmax(hipStream_t s, int dtype, const void* x, int ldx, const float* w, const float* b,
                          float* logp, uint8_t* cls, int64_t M, int in, int C);

// ---------------------------------------------------------------- resnet.hip (speaker-embedding ResNet34)
// mean[b][bin] over the nfr fbank frames of window win[b] (frames start at win*frames_per_step)
int emb_window_mean(hipStream_t s, const float* fb, const int64_t* win, int B, int frames_per_step, int nfr, float* mean);
// stem conv (1 -> C channels, 3x3 pad 1, folded BN, ReLU) from the shared fbank; out T [B][F+2][NT+2][C]
int emb_conv1(hipStream_t s, int dtype, const float* fb, const int64_t* win, const float* mean, const float* w,
              const float* bias, void* out, int B, int F, int NT, int frames_per_step, int C);

// NHWC convolution with a one-pixel zero border on input and output:
// out[b][fo+1][to+1][:] = relu?( sum_taps in[b][s*fo+kh][s*to+kw][:] . w[tap] + bias + res[b][fo+1][to+1][:] )
struct ConvArgs {
  const void* in;     // T [B][Fi+2][Ti+2][Cin]
  const void* w;      // T [taps][Cin/CK][Cout][CK], CK = 64 bytes of input channels (BN folded)
  const float* bias;  // [Cout]
  const void* res;    // T [B][Fo+2][To+2][Cout] or null
  void* out;          // T [B][Fo+2][To+2][Cout]
  int B, Fi, Ti, Cin, Fo, To, Cout;
  int stride;         // 1 | 2
  int taps;           // 9 (3x3, pad 1) | 1 (1x1)
  int relu;
  const void* w_ig;   // bf16 [Cout][9][Cin] (BN folded) for conv_gemm.hip, or null (last: ConvArgs a{} leaves it null)
  // conv_gemm.hip, fused projection shortcut (round 4, opt-in): out = relu(conv3x3(in) + conv1x1_stride(in2) + bias) -- the second
  // convolution of a ResNet block that changes shape takes the block's 1x1 / stride-2 shortcut into its own K loop: w_ig rows are
  // then [9 Cin + Cin2] long (the shortcut's weights appended), bias is the sum of both, res must be null
  const void* in2;    // bf16 [B][Fi2+2][Ti2+2][Cin2] or null
  int Cin2, Fi2, Ti2, stride2;
  // conv_gemm.hip, fp8 form (round 4 candidate: conv_igemm8_kernel; stages 3-4 of the ResNet34 trunk):
  // the A operand is the e4m3 copy of the bordered NHWC activation (value = fp8 * a_scale), the weights e4m3 with one scale
  // per output channel; outputs: bf16 (out, nullable) and / or e4m3 at out8_inv_scale (out8, nullable: the next convolution's
  // operand); amax8 (nullable) receives the running maximum of the output as float bits (ReLU output: >= 0)
  const void* in8;        // fp8 [B][Fi+2][Ti+2][Cin] or null (null: the bf16 kernels)
  const void* w8;         // fp8 [Cout][9][Cin]
  const float* w8_scale;  // [Cout]
  float a_scale;
  void* out8;             // fp8 [B][Fo+2][To+2][Cout] or null
  float out8_inv_scale;
  unsigned* amax8;
  unsigned* sat8;         // += values of the fp8 output clipped at 448
};
int conv2d(hipStream_t s, int dtype, const ConvArgs& a);
// conv_block.hip: a whole stride-1 BasicBlock of 32 channels in one kernel (bf16): out = relu(conv_b(relu(conv_a(in) + ba)) + bb + in),
// BN folded; the intermediate tensor lives in LDS, the residual comes out of the input patch.  Same tensor layouts as conv2d.
struct ConvBlockArgs {
  const void* in;      // bf16 [B][F+2][T+2][32]
  const void* wa;      // bf16 [9][1][32][32] (conv2d's layout)
  const float* ba;     // [32]
  const void* wb;
  const float* bb;
  void* out;           // bf16 [B][F+2][T+2][32]
  int B, F, T;
};
bool conv_block32_applicable(int dtype, int cin, int cmid, int cout, int stride_a, int stride_b, int taps_a, int taps_b, int F, int T);
int conv_block32(hipStream_t s, const ConvBlockArgs& a);
// conv_s2.hip: the block that opens the 64-channel stage in one kernel (bf16): out = relu(conv3x3_s2(in) + bias), sc = conv1x1_s2(in) + bsc
struct ConvS2Args {
  const void* in;      // bf16 [B][Fi+2][Ti+2][32]
  const void* w;       // bf16 [9][1][64][32] (conv2d's layout)
  const float* bias;   // [64]
  const void* wsc;     // bf16 [1][1][64][32]
  const float* bsc;    // [64]
  void* out;           // bf16 [B][Fo+2][To+2][64]
  void* sc;            // bf16 [B][Fo+2][To+2][64]
  int B, Fi, Ti, Fo, To;
};
bool conv_s2sc_applicable(int dtype, int cin, int cout, int stride, int taps, int sc_cin, int sc_cout, int sc_stride, int sc_taps,
                          int Fi, int Ti, int Fo, int To);
int conv_s2sc(hipStream_t s, const ConvS2Args& a);
// conv_gemm.hip: 3x3 stride-1 convolution as an implicit GEMM on the LDS-DMA loop (bf16, Cin % 64 == 0, Cout % 128 == 0);
// conv2d() routes to it when a.w_ig is set
bool conv_igemm_applicable(int dtype, const ConvArgs& a);
bool conv_igemm_wide(const ConvArgs& a);
bool conv_igemm8_applicable(int dtype, const ConvArgs& a);      // fp8 operands (a.in8 != null)
int conv_igemm8(hipStream_t s, const ConvArgs& a);
// resnet.hip (round 4 candidate): e4m3 copy of a bf16 activation tensor at one scale (value = fp8 * scale), and the running
// maximum of |x| as float bits (calibration)
int act_quant_fp8(hipStream_t s, const void* in_bf16, void* out_fp8, size_t n, float scale, unsigned* sat);
int act_amax_bf16(hipStream_t s, const void* in_bf16, size_t n, unsigned* amax);
// conv_stream.hip: the stride-1 3x3 convolutions of the 32- and 64-channel stages (bf16) as a stream of tiles per workgroup
// (weights resident in LDS, patches by LDS-DMA ahead of the MFMAs); bit-identical with conv2d's direct kernel.
// RVD_CONV_STREAM=0 turns it off, n >= 1 splits the time axis of a row of tiles over n workgroups (default 1).  The 64-channel
// stage takes it only with RVD_CONV_STREAM64=1: measured equal to the direct kernel there (52.4 vs 51.5 ms per hour of audio;
// the 32-channel stage 53.3-55.2 vs 65.7-66.3 ms, profiles/r04_call16_fast_epilogue.txt).
#define CONV_STREAM_DEFAULT 1
// conv_row64.hip: the stride-1 3x3 convolutions of the 64-channel stage (bf16): weights in registers, accumulators = 8 consecutive
// channels of a pixel per lane, swizzled patch, two workgroups per CU
bool conv_row64_applicable(int dtype, const ConvArgs& a);
int conv_row64(hipStream_t s, const ConvArgs& a);
bool conv_stream_applicable(int dtype, const ConvArgs& a);
int conv_stream(hipStream_t s, const ConvArgs& a);
int conv_igemm(hipStream_t s, const ConvArgs& a);

// weighted mean/std over time of the trunk output x T [B][F+2][TT+2][C] for each item (item_b = batch row,
// mask [n_items][mask_len] resampled to TT by nearest); stats T [n_items][2*C*F]
int tstp_pool(hipStream_t s, int dtype, const void* x, const int* item_b, const float* mask, int mask_len, int n_items,
              int F, int TT, int C, void* stats);

// ---------------------------------------------------------------- linkage.hip
// scipy.cluster.hierarchy.linkage(X, "centroid", "euclidean") for X fp64 [n][d] (device): Z fp64 [n-1][4] in merge
// order.  Scratch (device): D [n*n] doubles, size [n] uint16 (initialised to 1), cluster_id [n] (arange) and
// neighbor [n] ints, min_dist [n] doubles, scratch >= 4 KiB (control block of the multi-workgroup merge loop; null = one
// workgroup only).  From 3 000 points on the merge loop runs on 16 workgroups of one XCD (RVD_LINKAGE_MB) and the call then
// returns after the loop has finished (it checks the loop's status and falls back to the one-workgroup loop if needed).
// `workgroups`: 0 = default (16), 1 = the one-workgroup loop, 2 / 4 / 8 / 16 = that many workgroups (fewer disturb a concurrent
// kernel less: the joint pipeline clusters underneath the ASR encoder).
int centroid_linkage(hipStream_t s, const double* X, int n, int d, double* D, uint16_t* size, int* cluster_id, int* neighbor,
                     double* min_dist, double* Z, void* scratch, int workgroups = 0);

}  // namespace rvb
