/* test_api.h -- entry points of librvb_test.so: raw kernel / host-search hooks for the unit tests and the tuning scripts
 * (host buffers in, host buffers out).  NOT part of the product: librvb.so exports include/rvb.h and include/rvd.h only;
 * reverb_amd/build.py links the same objects plus csrc/test_api.hip (and engine.hip compiled with -DRVB_TEST_API) into
 * reverb_amd/librvb_test.so, which tests/ and scripts/ load through reverb_amd._lib.load_test(). */
#ifndef RVB_TEST_API_H_
#define RVB_TEST_API_H_
#include "../../include/rvb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- raw kernel entry points for unit tests (host buffers in, host buffers out) ---- */
int rvb_test_gemm(int dtype, const float* A, const float* W, const float* bias, const float* res, float* C,
                  int M, int N, int K, float alpha, int act, int out_f32,
                  int conv, int cT1, int cF1, int cC, int cB);
int rvb_test_rownorm(int dtype, const float* x, const float* gamma, const float* beta, float eps, int mode,
                     int silu, const float* add, float* out, int out_f32, int M, int d);
/* conv_block.hip (a whole 32-channel BasicBlock per launch) on host floats, unbordered NHWC in / out, torch weight layout */
int rvb_test_conv_block32(const float* x, const float* wa, const float* ba, const float* wb, const float* bb, float* out, int B, int F, int T);
// conv_s2.hip on a plane of its own: x [B][Fi][Ti][32], w [64][32][3][3], wsc [64][32] -> out = relu(conv3x3 stride 2) and sc = conv1x1 stride 2, [B][Fo][To][64]
int rvb_test_conv_s2sc(const float* x, const float* w, const float* b, const float* wsc, const float* bsc, float* out, float* sc, int B, int Fi, int Ti);
int rvb_test_conv1(int dtype, const float* feats, const float* mean, const float* istd, const float* w,
                   const float* b, float* out, int B, int T0, int F0, int d);
/* round-4 candidate: the fp8 implicit-GEMM convolution (csrc/conv_gemm.hip conv_igemm8_kernel) on host floats, see test_api.hip */
int rvb_test_conv_igemm_fp8(const float* x, const float* w, const float* bias, const float* res, float* out, float* out8, int B,
                            int Fi, int Ti, int Cin, int Cout, int stride, int relu, float a_scale, float out8_scale, float* x_deq,
                            float* w_deq, float* amax);
/* host only: the `joint_decoding` state machine of one chunk (csrc/search.cpp JointSearch: transformer/search.py:450-496,
 * espnet/beam_search_timesync.py), driven frame by frame; the caller supplies the attention log-probs it asks for */
void* rvb_test_joint_new(int beam, int pre_beam, int blank, int sos, double w_ctc, double w_dec, double bonus);
void rvb_test_joint_free(void* h);
int rvb_test_joint_begin(void* h, int t, const float* tv, const int32_t* ti, int K, float p_tok0, float p_blank, int32_t* decode,
                         int32_t* n_decode, int32_t* pair_node, int32_t* pair_tok, int32_t* n_pairs, int cap);
int rvb_test_joint_finish(void* h, const float* vals);
int rvb_test_joint_prefix(void* h, int node, int32_t* toks, int32_t* n);
int rvb_test_joint_result(void* h, int32_t* tokens, int32_t* times, int32_t* end_times, double* conf, int32_t* n, double* score);
int rvb_test_glu_dwconv(int dtype, const float* G, const float* pw1_bias, const float* dw_w, const float* dw_b,
                        const int32_t* lens, float* out, int B, int T, int d, int K, int causal,
                        const float* hist /* nullable [K-1][2d] */, int hist_rows);
int rvb_test_attention(int dtype, const float* q, const float* k, const float* v, const float* p,
                       const float* bias_u, const float* bias_v, float* out, int q_rows, int kv_rows, int p_rows,
                       int heads, int dk, const int32_t* q_start, const int32_t* q_len, const int32_t* kv_start,
                       const int32_t* kv_len, int nseq, int causal);
/* decoder self attention over shared prefixes: sequence s has kv_len[s] keys in rows kv_index[kv_start[s] ..] and
 * q_len[s] queries in rows q_start[s] .. at key positions q_pos0[s] .. (causal); q_block 16 = one-wave blocks + work list */
int rvb_test_attention_trie(int dtype, const float* q, const float* k, const float* v, float* out, int rows, int heads, int dk,
                            const int32_t* q_start, const int32_t* q_len, const int32_t* q_pos0, const int32_t* kv_start,
                            const int32_t* kv_len, const int32_t* kv_index, int n_index, int nseq, int q_block);
int rvb_test_logsoftmax_topk(const float* logits, int M, int V, int k, float blank_penalty, int blank_id,
                             float* topk_val, int32_t* topk_idx, float* logp);
int rvb_test_lse_gather(const float* logits, int R, int V, const int32_t* target, float* out);
/* fp8 (e4m3) GEMM / LayerNorm-to-fp8 of the RVB_FP8 mode on host floats (operands quantised as the engine does) */
/* bf16 GEMM with bf16 output and the row-periodic addend of GemmArgs::rowadd (round 6): C[m][n] = A.W^T + bias (+ add[m % add_rows][n - add_col0]
 * for add_col0 <= n < add_col0 + add_cols); add is fp32 on the host, rounded to bf16 on the way up (what the engine's positional keys are) */
int rvb_test_gemm_glu(const float* A, const float* W, const float* bias, float* C, int M, int N, int K);
int rvb_test_gemm_rowadd(const float* A, const float* W, const float* bias, const float* add, float* C, int M, int N, int K,
                         int add_rows, int add_col0, int add_cols);
/* csrc/mp3.cpp (round 6).  decode: as rvb_audio_decode_f32 + the stream facts (info9: version, channels, rate, audio frames, samples per
 * frame, info frame, start skip, samples, kbit/s) and what the pass saw (stats12: granule-channels, Huffman data ending exactly on /
 * before / past part2_3_length, CRCs checked / failed, frames without their reservoir bytes, short / mixed / M-S / intensity granules,
 * largest main_data_begin).  hybrid / polyphase: one granule of the two synthesis stages on caller state.  window: D[512].
 * huffman: table t of the standard (32 / 33 = count1 A / B) -> number of entries, (code, length) per symbol, linbits per table_select. */
int64_t rvb_test_mp3_decode(const void* data, int64_t nbytes, int channel, float* out, int64_t capacity, int64_t* info9, int64_t* stats12, int threads);
int rvb_test_mp3_hybrid(float* xr576, float* overlap576, int block_type, int mixed, float* out576);
int rvb_test_mp3_polyphase(const float* sb576, float* vbuf1024, int* voff, float* pcm576);
int rvb_test_mp3_window(float* out512);
int rvb_test_mp3_huffman(int t, uint16_t* codes, uint8_t* lens, int32_t* linbits32);
int rvb_test_gemm_fp8(const float* A, const float* W, const float* bias, const float* res, float* C, int M, int N, int K,
                      float a_scale, float alpha, int act, int out_kind, float out_scale, float* a_deq, float* w_deq);
int rvb_test_rownorm_fp8(const float* x, const float* gamma, const float* beta, float eps, int silu, int M, int d, float scale,
                         float* out, const float* gamma2, const float* beta2, float eps2, float scale2, float* out1_f32, float* out2);
int rvb_test_lse_gather_multi(const float* logits, int R, int V, const int32_t* ptr /* [R+1] */, const int32_t* target,
                              int P, float* out /* [P] */);
/* host only: the trie of distinct hypothesis prefixes attention rescoring computes decoder rows for (engine.hip build_trie) */
int rvb_test_build_trie(const int32_t* tokens, const int32_t* lens, const int32_t* chunk_of, int n_hyps, int n_chunks, int sos, int eos,
                        int reversed, int32_t* n_rows, int32_t* tok, int32_t* pos, int32_t* path, int32_t* hq_start, int32_t* hq_len,
                        int32_t* hq_pos0, int32_t* tgt_ptr, int32_t* tgt, int32_t* pair_slot, int32_t* n_work);
/* host only: the worker pool of the CTC search / trie building (engine.h HostPool) runs `rounds` jobs on up to n_threads threads;
 * fails unless every work item of every job was executed exactly once */
int rvb_test_host_pool(int n_threads, int items, int rounds);
int rvb_test_fbank(const int16_t* pcm, int64_t n_samples, float* feats /* [frames,80] */);
/* native prefix beam search on host arrays: top-k log-probs/indices [T,beam] of one utterance */
int rvb_test_prefix_beam(const float* topk_val, const int32_t* topk_idx, int T, int beam, int blank,
                         int32_t* n_hyps, int32_t* tokens /* [beam][T] */, int32_t* lens, int32_t* times,
                         int32_t* times_lens, double* scores);

/* GEMM kernel selection / micro-benchmark hooks (tests and tuning only) */
int rvb_test_set_gemm_variant(int variant /* 0 auto, 1 gemm.hip 128x128, 2 gemm2.hip 256x256 LDS-DMA */);
/* gemm2.hip tuning switches: flags bit 0 = 32x32x16 MFMAs, bit 1 = s_setprio for the later-dispatched waves;
 * group_m = tile order (0/1 row-major inside an XCD's run, n = n row tiles down then the next column); -1 = defaults */
int rvb_test_set_gemm2_opts(int flags, int group_m);
/* per-workgroup phase timestamps of one bf16 gemm2 launch (measurement aid, scripts/gemm_timeline.py) */
int rvb_test_gemm_timeline(int M, int N, int K, int act, int out_f32, int with_res, long long* out, int cap, int* n_wg);
int rvb_test_gemm_bench(int dtype, int M, int N, int K, int variant, int iters, int act, int out_f32, int with_res,
                        double* ms_out, double* max_abs_diff_vs_variant1);

#ifdef __cplusplus
}
#endif
#endif /* RVB_TEST_API_H_ */
