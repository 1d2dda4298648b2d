/* librvb -- C ABI of the MI355X-native Reverb-ASR hot path (gfx950 HIP kernels behind it).
 *
 * The reference (revdotcom/reverb) is pure Python and exposes no FFI; its boundary for the
 * `recognize_wav` hot path is the Python API in asr/wenet/cli/reverb.py and the inner seam
 * `ASRModel.decode` (asr/wenet/transformer/asr_model.py:331-432).  Each entry point below names
 * the reference interface it replaces.  The Python host in reverb_amd/ binds these with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions: plain C, opaque handle, every function returns 0 (RVB_OK) or a negative RVB_E*
 * code; rvb_last_error() returns the calling thread's last error string.  Host buffers are
 * caller-owned; device memory is library-owned.  One engine per device; an engine is not
 * thread-safe, distinct engines are.  All device work runs on the engine's own HIP stream and
 * every function returns after its results are complete on the host.
 */
#ifndef RVB_H_
#define RVB_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVB_OK 0
#define RVB_E_ARG -1
#define RVB_E_HIP -2
#define RVB_E_STATE -3
#define RVB_E_NOMEM -4
#define RVB_E_UNSUPPORTED -5
#define RVB_E_TIMEOUT -7        /* a collective did not complete within rvb_comm_set_timeout (-6: corrupt audio data) */

#define RVB_F32 0  /* parity mode: v_mfma_f32_16x16x4_f32, exact f32 fma chains */
#define RVB_BF16 1 /* throughput mode: v_mfma_f32_16x16x32_bf16, fp32 accumulate/residual */
#define RVB_FP8 2  /* BASELINE configs[4]: the bf16 engine with the encoder's feed-forward GEMMs (rvb_set_fp8_policy: also qkv / pointwise-conv) on
                      v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 operands: weights scaled per output channel, activations per
                      tensor, scales calibrated by the FIRST rvb_encode call, which itself runs in bf16) */

typedef struct rvb_engine rvb_engine;

/* Model dimensions: the values the reference reads from config.yaml
 * (asr/wenet/utils/init_model.py:102-183; cli/reverb.py:62-98). */
typedef struct rvb_model_cfg {
  int32_t struct_size;   /* sizeof(rvb_model_cfg) as the CALLER's binding lays it out: rvb_create refuses any other value
                            (RVB_E_ARG, "ABI mismatch"), so a binding written against an older header -- a field short, as
                            INTEGRATION.md's stub was after round 3 added cnn_causal -- fails by name instead of making the
                            library read past the caller's struct */
  int32_t dtype;         /* RVB_F32 | RVB_BF16 | RVB_FP8 */
  int32_t input_dim;     /* input_dim (80 mel bins) */
  int32_t vocab;         /* output_dim */
  int32_t d_model;       /* encoder_conf.output_size */
  int32_t heads;         /* encoder_conf.attention_heads */
  int32_t ffn_dim;       /* encoder_conf.linear_units */
  int32_t num_blocks;    /* encoder_conf.num_blocks (first and last are language-specific if num_langs>0) */
  int32_t cnn_kernel;    /* encoder_conf.cnn_module_kernel (odd unless cnn_causal) */
  int32_t cnn_norm;      /* 0 layer_norm, 1 batch_norm (encoder_conf.cnn_module_norm) */
  int32_t num_langs;     /* dataset_conf.cat_emb_conf.emb_len when pass_cat_emb, else 0 */
  int32_t dec_heads;     /* decoder_conf.attention_heads */
  int32_t dec_ffn_dim;   /* decoder_conf.linear_units */
  int32_t dec_blocks;    /* decoder_conf.num_blocks   (left decoder) */
  int32_t dec_r_blocks;  /* decoder_conf.r_num_blocks (right decoder, 0 = none) */
  int32_t blank_id;      /* ctc_conf.ctc_blank_id */
  int32_t sos_id;        /* vocab-1 (asr_model.py:79-82) */
  int32_t eos_id;
  int32_t max_chunks;    /* chunks per device batch (workspace sizing) */
  int32_t chunk_frames;  /* max input frames per chunk (2051, cli/reverb.py:188) */
  int32_t cnn_causal;    /* encoder_conf.causal: the convolution module pads cnn_kernel-1 frames on the left only
                            (convolution.py:55-57,113-121) and carries a cnn cache across streamed chunks */
} rvb_model_cfg;

const char* rvb_last_error(void);
const char* rvb_version(void);
/* sizeof(rvb_model_cfg) of THIS build: what a binding's struct_size has to equal (a binding can assert it at load) */
int rvb_model_cfg_size(void);

/* ReverbASR.__init__ / init_model / load_checkpoint (cli/reverb.py:46-98,
 * utils/init_model.py:99-277, utils/checkpoint.py:29-80): create an engine, stream the
 * state-dict tensors in by their reference names (fp32, host memory), then finalize. */
int rvb_create(const rvb_model_cfg* cfg, int device, rvb_engine** out);
void rvb_destroy(rvb_engine* e);
int rvb_load_tensor(rvb_engine* e, const char* name, const float* host, const int64_t* shape, int ndim);
/* Fold the language-specific linear layers with `cat_embs` = [verbatimicity, 1-verbatimicity]
 * (cli/reverb.py:215-217; encoder_layer.py:378-390, decoder_layer.py:319-330), pack weights to
 * the compute dtype, precompute the per-layer positional keys.  May be called again with other
 * cat_embs.  Fails listing the first missing tensor. */
int rvb_finalize(rvb_engine* e, const float* cat_embs, int n_cat);

/* Same for PCM at another sample rate: resampled on the device to 16 kHz the way the reference does
 * (asr/wenet/cli/reverb.py:128-134: `torchaudio.transforms.Resample(sr, 16000)` on the float waveform =
 * sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99); the fbank then reads the un-rounded float result. */
int rvb_upload_pcm_rate(rvb_engine* e, const int16_t* pcm, int64_t n_samples, int sample_rate);
/* the waveform the fbank will read (after resampling, int16 scale); out may be NULL to query the length */
int rvb_get_waveform(rvb_engine* e, float* out, int64_t* n_samples);

/* ReverbASR.compute_feats (cli/reverb.py:119-146): Kaldi fbank of 16 kHz mono int16 PCM.
 * rvb_upload_pcm copies the samples to HBM; rvb_fbank computes [n_frames,80] log-mel on the
 * device (kept resident, zero-padded to whole chunks) and optionally copies it to feats_out.  With feats_out == NULL the
 * call only enqueues the kernel (rvb_encode is ordered behind it on the engine's stream); with a host buffer it returns
 * when the copy has landed. */
int64_t rvb_num_frames(int64_t n_samples);
/* ReverbASR.compute_feats with front-end settings OTHER than the model's 80 bins / 25 ms / 10 ms (cli/reverb.py:119-146 passes
 * num_mel_bins, frame_length, frame_shift straight to kaldi.fbank; its own default is 23 bins): stand-alone, not bound to an engine.
 * wave: mono waveform at 16 kHz on the int16 scale (what `.to(torch.float)` of the loaded file holds), host memory; feats_out
 * [n_frames][num_mel_bins] (NULL: only *n_frames is set).  Supported: frame_length 16.1 .. 32 ms (Kaldi pads the window to the
 * next power of two: the 512-point FFT of the hot path), any frame_shift, 1 .. 128 bins, dither 0; RVB_E_UNSUPPORTED otherwise. */
int rvb_compute_feats(int device, const float* wave, int64_t n_samples, int num_mel_bins, double frame_length_ms,
                      double frame_shift_ms, float* feats_out, int64_t* n_frames);
int rvb_upload_pcm(rvb_engine* e, const int16_t* pcm, int64_t n_samples);
/* Page-locked host memory for the audio reader (what `torchaudio.load` fills in the reference, cli/reverb.py:128):
 * PCM decoded straight into such a buffer reaches HBM at the PCIe rate (115 MB per hour of audio in ~2 ms);
 * pageable memory works with rvb_upload_pcm too, several times slower.  Free with rvb_host_free. */
/* Double-buffered upload for back-to-back recordings (a long-form service decodes file i while file i+1 arrives): the copy
 * is enqueued on a copy stream of the engine's and the call returns at once; from page-locked memory it runs on a DMA engine
 * underneath the decoding in progress.  The samples become the engine's audio at the NEXT rvb_fbank, which orders itself
 * behind the copy on the device; `pcm` must stay valid and unchanged until that rvb_fbank has been called and one more
 * stream synchronisation (any blocking call) has passed.  One upload may be pending at a time (RVB_E_STATE otherwise).
 * 16 kHz int16 only.  Replaces: the `torchaudio.load` + `.to(device)` of the NEXT file in cli/reverb.py:128-146, which the
 * reference runs serially. */
int rvb_upload_pcm_async(rvb_engine* e, const int16_t* pcm, int64_t n_samples);
int rvb_host_alloc(void** out, int64_t bytes);
int rvb_host_free(void* p);
int rvb_fbank(rvb_engine* e, float* feats_out /* nullable [n_frames*80] */, int64_t* n_frames);

/* The audio reader itself -- `torchaudio.load(audio_file, normalize=False)` followed by `.to(torch.float)`
 * (asr/wenet/cli/reverb.py:128-130) -- for the lossless containers, decoded on the host from a file image in memory
 * (csrc/audio.cpp).  normalize=False keeps the decoder's native sample format, so the VALUES depend on the container:
 *   RIFF/WAVE  PCM 16 (int16), PCM 8 (uint8, 0..255 as stored), PCM 24 / 32 (int32; 24-bit left-justified = value << 8),
 *              IEEE float 32 / 64 (as stored), A-law / mu-law (expanded to int16); WAVE_FORMAT_EXTENSIBLE and RF64 of the same
 *   AIFF / AIFF-C  NONE / twos / sowt PCM 8 (signed in the file, handed on as uint8 = value + 128) / 16 / 24 / 32, fl32 / fl64, alaw / ulaw
 *   FLAC       (RFC 9639) <= 16 bits per sample: int16, left-justified; wider: int32, left-justified; frame CRC-8 / CRC-16
 *              and the STREAMINFO MD5 of the decoded PCM are verified (md5_checked = 1 when the file carries a signature)
 *   MPEG audio Layer III (MP3; MPEG-1, MPEG-2 and 2.5 sampling rates, mono / stereo / joint stereo, CRC checked when present): float32,
 *              full scale 1.0; a leading Xing / Info / VBRI frame is skipped and the encoder delay / padding of a LAME-style tag are trimmed
 *              the way FFmpeg (torchaudio's loader) trims them; free format and Layers I / II are refused by name
 * Ogg is refused by name (RVB_E_UNSUPPORTED); corrupt data is -6.
 * rvb_audio_probe fills `info` only.  The decode calls write channel `channel` (or, with channel = -1, all channels
 * planar [channels][frames]) into `out` of `capacity` samples and return the frames per channel (< 0: error code).
 * flags: RVB_AUDIO_NO_MD5 skips the FLAC signature check (the loader of the reference does not check it either); bits 8..15 =
 * host threads for FLAC (0 = one per 512 KiB of stream, at most 16: frames decode independently, a 1 h file in ~0.1 s).
 * rvb_audio_decode_i16 serves the files whose native format is int16 (sample_format == RVB_SAMPLE_I16): the PCM can go
 * straight into a page-locked buffer for rvb_upload_pcm_rate; everything else goes through rvb_audio_decode_f32 and
 * rvb_upload_wave_f32. */
enum { RVB_AUDIO_WAVE = 1, RVB_AUDIO_FLAC = 2, RVB_AUDIO_AIFF = 3, RVB_AUDIO_MP3 = 4 /* MPEG-1 / 2 / 2.5 audio Layer III: float32 */ };
enum { RVB_AUDIO_NO_MD5 = 1 };
enum { RVB_SAMPLE_U8 = 1, RVB_SAMPLE_I16 = 2, RVB_SAMPLE_I32 = 3, RVB_SAMPLE_F32 = 4, RVB_SAMPLE_F64 = 5 };
typedef struct rvb_audio_info {
  int32_t container;        /* RVB_AUDIO_* */
  int32_t sample_format;    /* RVB_SAMPLE_*: the dtype of the tensor torchaudio would return */
  int32_t channels;
  int32_t sample_rate;
  int32_t bits_per_sample;  /* as the file declares it */
  int32_t md5_checked;
  int64_t frames;           /* samples per channel */
  int32_t decode_threads;   /* host threads the decode used (FLAC: runs of frames) */
  int32_t reserved;
} rvb_audio_info;
int rvb_audio_probe(const void* data, int64_t nbytes, rvb_audio_info* info);
int64_t rvb_audio_decode_f32(const void* data, int64_t nbytes, int channel, float* out, int64_t capacity, int flags, rvb_audio_info* info);
int64_t rvb_audio_decode_i16(const void* data, int64_t nbytes, int channel, int16_t* out, int64_t capacity, int flags, rvb_audio_info* info);
/* rvb_upload_pcm_rate for a float waveform (`waveform.to(torch.float)` of a non-int16 source, cli/reverb.py:130-134):
 * resampled on the device when sample_rate != 16000; the fbank reads the float values as they are. */
int rvb_upload_wave_f32(rvb_engine* e, const float* wave, int64_t n_samples, int sample_rate);

/* Chunk-masked encoder attention for the following rvb_encode calls: query frame i attends the encoder frames
 * [max((i/chunk - left)*chunk, 0), (i/chunk + 1)*chunk) -- what `decoding_chunk_size` / `num_decoding_left_chunks`
 * select in BaseEncoder.forward via add_optional_chunk_mask (encoder.py:140-145, utils/mask.py:86-197) for models
 * trained with use_dynamic_chunk / static_chunk_size.  chunk_size <= 0: full context (default); left < 0: all. */
int rvb_set_decoding_chunk(rvb_engine* e, int chunk_size, int num_left_chunks);

/* RVB_FP8 engines: which GEMM groups of which conformer blocks run on the fp8 MFMA path -- `groups` is a bit mask (1 macaron
 * feed-forward, 2 qkv projection, 4 pointwise conv 1, 8 pointwise conv 2, 16 feed-forward) applied to the blocks
 * first_block .. last_block (last_block < 0: the last one), plus 32 = the subsampling's second convolution (not per block;
 * rvb_get_fp8_subsample); everything else stays bf16.  groups < 0 restores the default: the
 * feed-forward modules of every block (17), the policy whose token error rate against the reference stays at the level of the
 * reference's own bf16 autocast (tests/test_fp8_gpu.py).  Takes effect with the next rvb_encode; call after rvb_finalize. */
int rvb_set_fp8_policy(rvb_engine* e, int groups, int first_block, int last_block);

/* ASRModel._forward_encoder + ctc_logprobs + per-frame top-`beam` (asr_model.py:288-329,378-389,
 * search.py:155): encode a batch of B chunks of T0 frames.  feats: host fp32 [B,T0,80], or NULL
 * to read chunk rows [first_chunk, first_chunk+B) of the device-resident rvb_fbank output.
 * lens: valid input frames per chunk (feats_batcher, cli/reverb.py:148-180). */
int rvb_encode(rvb_engine* e, const float* feats, int64_t first_chunk, const int32_t* lens, int B, int T0,
               int beam, float blank_penalty);
/* Streaming encoder: BaseEncoder.forward_chunk / forward_chunk_by_chunk (asr/wenet/transformer/encoder.py:231-402) for
 * one stream.  rvb_stream_begin empties the attention cache; rvb_stream_chunk encodes one chunk of `n_frames` input frames
 * ((chunk-1)*4 + 7 for a full chunk, encoder.py:378-381) against the cached keys / values of earlier chunks, positional keys
 * at absolute frame positions, and keeps the last `required_cache_size` frames (< 0: all, 0: none) as forward_chunk does;
 * `out` (nullable) receives the chunk's encoder output [n_out, d] fp32.  The caches live in the engine (the reference
 * returns att_cache / cnn_cache tensors): per block the key|value rows of the cached frames and, for a causal convolution
 * module (cnn_causal), the pointwise-conv1 outputs of the last cnn_kernel-1 frames -- the reference's cnn_cache holds the
 * module inputs of those frames and pushes them through pointwise_conv1 again with every chunk (convolution.py:113-125).
 * rvb_stream_finish runs the CTC head + top-k over all frames produced and makes them the current batch (one "chunk"), so
 * that rvb_ctc_greedy / rvb_ctc_prefix_beam / rvb_attention_rescore / rvb_get_encoder_out work on the streamed output --
 * ASRModel.decode(simulate_streaming=True) (asr_model.py:301-306). */
int rvb_stream_begin(rvb_engine* e);
int rvb_stream_chunk(rvb_engine* e, const float* feats, int n_frames, int required_cache_size, float* out, int32_t* n_out);
int rvb_stream_state(rvb_engine* e, int32_t* offset, int32_t* cache_frames);
int rvb_stream_finish(rvb_engine* e, int beam, float blank_penalty);
int rvb_encoder_frames(rvb_engine* e, int32_t* T_out);                 /* encoder frames per chunk (512) */
int rvb_get_encoder_lens(rvb_engine* e, int32_t* lens /* [B] */);      /* encoder_mask.sum (asr_model.py:387) */
int rvb_get_encoder_out(rvb_engine* e, float* out /* [B,T,d] */);      /* parity tap */
int rvb_get_ctc_logprobs(rvb_engine* e, int chunk, float* out /* [T,V] */); /* parity tap */
int rvb_get_ctc_topk(rvb_engine* e, float* vals, int32_t* idx /* [B,T,beam] each */);

/* ctc_greedy_search (search.py:106-121): tokens [B,T] (first ntok[b] valid) and the frame of each
 * token's first emission. */
int rvb_ctc_greedy(rvb_engine* e, int32_t* tokens, int32_t* ntok, int32_t* frames);

/* ctc_prefix_beam_search (search.py:124-248), float64 host arithmetic, one host thread per chunk.
 * Results are kept in the engine; read them with rvb_get_nbest. */
int rvb_ctc_prefix_beam(rvb_engine* e, int beam);
int rvb_get_nbest_count(rvb_engine* e, int chunk, int32_t* n_hyps, int32_t* max_len);
/* tokens/times: [n_hyps][max_len] padded with -1; lens/times_lens/scores: [n_hyps] */
int rvb_get_nbest(rvb_engine* e, int chunk, int32_t* tokens, int32_t* lens, int32_t* times, int32_t* times_lens,
                  double* scores);

/* attention_rescoring (search.py:363-448) over the stored n-best of every chunk of the batch.
 * Per chunk: index of the winning hypothesis, its score (fp32 accumulation as the reference),
 * confidence, and per-token confidences [max_len] of the winner. */
/* Optional, between rvb_encode and rvb_ctc_prefix_beam when rvb_attention_rescore will follow: enqueues the decoders' memory
 * key / value projections (decoder_layer.py:112-119 `src_attn(x, memory, memory)`), which depend on the encoder output alone,
 * so that the device computes them while the host searches the last slice.  rvb_attention_rescore does it itself otherwise. */
int rvb_prepare_rescoring(rvb_engine* e, int right_to_left);
int rvb_attention_rescore(rvb_engine* e, double ctc_weight, double reverse_weight);
/* `attention` mode (asr/wenet/transformer/search.py:251-360): autoregressive beam search with the left decoder on
 * the chunks of the last rvb_encode; per-hypothesis K/V caches on the device, beam bookkeeping on the host in float32
 * as the reference does.  Runs until every beam ended with <eos> or for encoder-frames steps. */
int rvb_attention_decode(rvb_engine* e, int beam, float length_penalty);
/* best hypothesis of one chunk without <sos>/<eos>; `tokens` needs room for rvb_encoder_frames() entries */
int rvb_get_attention_result(rvb_engine* e, int chunk, int32_t* tokens, int32_t* n_tokens, float* score);

/* `joint_decoding` mode (asr/wenet/transformer/search.py:450-496 -> espnet/beam_search_timesync.py:86-508): time-synchronous
 * joint CTC / attention beam search on the chunks of the last rvb_encode, which must have kept at least
 * int(pre_beam_ratio * beam) log-probs per frame.  Every chunk advances one encoder frame per iteration; the prefixes of all
 * chunks that need the attention decoder form one batched decoder step.  The decoder's memory is the chunk's
 * valid frames as a (1, len, d) tensor -- the shape BeamSearchTimeSync.reset expects; the reference's own call passes a 2-D
 * tensor and raises there.  length_bonus is what ASRModel.decode passes as `length_penalty` (asr_model.py:427-431).
 * Result per chunk: winner without <sos>, start / end frame and confidence (max of CTC and attention) per token, joint score. */
int rvb_joint_decode(rvb_engine* e, int beam, double ctc_weight, double pre_beam_ratio, double length_bonus);
int rvb_get_joint_result(rvb_engine* e, int chunk, int32_t* tokens, int32_t* times, int32_t* end_times, double* tokens_confidence,
                         int32_t* n_tokens, double* score);
int rvb_get_joint_stats(rvb_engine* e, int64_t* decoder_rows, int64_t* steps);

int rvb_get_rescored(rvb_engine* e, int chunk, int32_t* best_index, float* score, double* confidence,
                     double* tokens_confidence /* [len of best] */);
/* all chunks of the batch at once: the winning hypothesis of each chunk, arrays padded to [B][T]
 * (T = rvb_encoder_frames) with -1 / 0 */
int rvb_get_rescored_batch(rvb_engine* e, int32_t* lens, int32_t* tokens, int32_t* times_lens, int32_t* times,
                           float* scores, double* confidences, double* tokens_confidence);
/* fp8 engines: forget the activation scales; the next rvb_encode calibrates again (in bf16) on its batch */
int rvb_fp8_recalibrate(rvb_engine* e);
/* fp8 engines: the calibrated per-tensor activation scales, 7 per conformer block (inputs of macaron FFN 1 / its hidden /
 * qkv / pointwise conv 1 / pointwise conv 2 / FFN 1 / its hidden; powers of two, value = fp8 * scale), followed by ONE more entry:
 * the scale of conv1's fp8 output (policy bit 5; 0 = not measured, conv2 then runs in bf16) -- n = 7 * blocks + 1 (round 5: until
 * then the subsampling scale was not part of the vector and ranks of a sharded run could not agree on it).  get: *n = 0 while the
 * engine is not calibrated; `scales` may be NULL to query n.  set: n = 7 * blocks (subsampling scale untouched) or 7 * blocks + 1;
 * installs scales and ends calibration, so that the very next
 * rvb_encode already runs in fp8 -- how the ranks of a sharded run agree on one set (element-wise maximum of what each rank
 * calibrated on its own slice, reverb_amd/dist.py), and how a deployment pins scales measured once. */
int rvb_get_fp8_scales(rvb_engine* e, float* scales, int32_t* n);
/* How many values did NOT fit e4m3 at the installed scales since they were calibrated / installed / last reset: the kernels that
 * write an fp8 operand (the LayerNorms in front of the fp8 GEMMs, the feed-forward's fp8 epilogue) clip to +-448 and count what they
 * clip, per block and activation slot in the order of rvb_get_fp8_scales (n = 7 * blocks; counts may be NULL to query n).
 * A calibrated engine decoding audio like its calibration batch reports zeros (2x headroom); anything else says the scales do
 * not cover this input -- recalibrate (rvb_fp8_recalibrate) or install wider scales.  reset != 0 zeroes the counters. */
int rvb_get_fp8_saturation(rvb_engine* e, uint32_t* counts, int32_t* n, int reset);
/* Round 4: rvb_set_fp8_policy's bit 5 puts the subsampling's second convolution (K = 9 d, a quarter of the encoder's FLOPs;
 * subsampling.py:189-190) on the fp8 path: conv1 then writes its ReLU output in e4m3 at a scale calibrated with the others.
 * *scale: that scale (0 while not calibrated -- conv2 then stays bf16; the last entry of rvb_get / rvb_set_fp8_scales); *clipped: values of conv1's output beyond 448 * scale since the last reset.  Either may be NULL. */
int rvb_get_fp8_subsample(rvb_engine* e, float* scale, uint32_t* clipped, int reset);
int rvb_set_fp8_scales(rvb_engine* e, const float* scales, int32_t n);
/* Work of the last rvb_attention_rescore: `pairs` = (hypothesis, position) log-probs served = the rows the reference's
 * padded [N, L] decoder batch computes (search.py:391-412); `decoder_rows` = rows actually computed: one per DISTINCT
 * hypothesis prefix of a chunk (the decoder is causal, so hypotheses of one beam share the rows of their common prefix). */
int rvb_get_rescore_stats(rvb_engine* e, int64_t* decoder_rows, int64_t* pairs);
/* per-hypothesis decoder log-probs of the last rescoring (parity tap): for hyp i of `chunk`,
 * out[j] = log p(w_j | ...) for j < len and out[len] = log p(eos); right=1 for the r2l decoder */
int rvb_get_rescore_logp(rvb_engine* e, int chunk, int hyp, int right, float* out);

/* Collectives of the chunk-sharded path (no data-path collective, ONE all-gather of the packed per-chunk results).
 * RCCL is bound directly (resolved with dlopen on first use; PyTorch is not needed): one rank calls rvb_comm_unique_id and
 * hands the 128 bytes to the others by any side channel, every rank calls rvb_comm_init on its engine (one engine = one GPU =
 * one rank), rvb_allgather_results gathers `bytes` bytes per rank (host buffers; recv = world * bytes) in rank order on the
 * engine's stream.  rvb_comm_create / rvb_comm_allgather / rvb_comm_free are the same collective on a stand-alone
 * communicator (own stream; one per process = per GPU), which the ASR results and the diarization shard's classes / embeddings
 * both travel on: reverb_amd/dist.py (`RvbComm`, default_comm) uses it whenever the ranks run on GPUs; torch.distributed only
 * carries the 128-byte id once (or a file does: RvbComm.from_file) and serves the gloo CPU tests. */
typedef struct rvb_comm rvb_comm;
int rvb_comm_unique_id(void* id128 /* out: 128 bytes */);
int rvb_comm_create(int device, int world, int rank, const void* id128, rvb_comm** out);
int rvb_comm_allgather(rvb_comm* c, const void* send, int64_t bytes, void* recv);
/* The collective alone (device to device, HIP events on the communicator's stream, checked): average milliseconds of one
 * all-gather of `bytes` bytes per rank -- the xGMI datapoint of SURVEY 8(e) for the top-beam posterior exchange. */
int rvb_comm_time_allgather(rvb_comm* c, int64_t bytes, int iters, double* avg_ms);
/* Failure handling (the reference has none: it is single-process).  With a timeout set, a collective that
 * does not complete in `seconds` -- a peer died or hangs -- returns RVB_E_TIMEOUT instead of blocking for ever; the
 * communicator is aborted (ncclCommAbort) and every later call on it returns RVB_E_STATE.  The host then exchanges through
 * its rendezvous channel and re-queues the missing rank's chunk range (reverb_amd/dist.py: decode_sharded).  0 = wait for ever. */
int rvb_comm_set_timeout(rvb_comm* c, double seconds);
/* What a launcher needs of a process group besides the gather: a barrier and the maximum of a double over the ranks
 * (bench.py's step timing) -- 8-byte all-gathers on the same communicator, so a rank holds ONE RCCL communicator. */
int rvb_comm_barrier(rvb_comm* c);
int rvb_comm_max_f64(rvb_comm* c, double* value /* in: this rank's, out: the maximum */);
/* The alternative exchange of SURVEY.md 8(e) -- top-beam CTC posteriors instead of decoded results -- device to device: the
 * [B, T, k] fp32 log-probs and int32 ids of the engine's last rvb_encode are gathered straight out of HBM into the
 * communicator's device buffer ([world][vals] then [world][ids]); host_out (nullable, 2 * world * B*T*k*4 bytes) receives a
 * copy; *bytes_per_rank = B*T*k*8.  Every rank must hold the same batch shape. */
int rvb_comm_allgather_topk(rvb_comm* c, rvb_engine* e, void* host_out, int64_t* bytes_per_rank);
int rvb_comm_free(rvb_comm* c);
int rvb_comm_init(rvb_engine* e, int world, int rank, const void* id128);
int rvb_allgather_results(rvb_engine* e, const void* send, int64_t bytes, void* recv);
int rvb_comm_destroy(rvb_engine* e);

/* Stage timing (HIP events on the engine stream).  level 1: every kernel family is bracketed;
 * names: "fbank","subsample","gemm","attention","rownorm","glu_dwconv","ctc_topk","embed",
 * "lse_gather","search_host".  level 2: only the GEMM launches (the dominant kernel; half the
 * events, ~1 % less perturbation of the step).  level 0: off.
 * flops: algorithmic FLOPs launched (gemm/attention only). */
int rvb_set_profiling(rvb_engine* e, int level);
int rvb_reset_timings(rvb_engine* e);
int rvb_get_timing(rvb_engine* e, const char* name, double* ms, double* flops, int64_t* launches);
/* ALGORITHMIC HBM bytes launched under `name` since the last reset ("gemm" / "gemm_fp8" only): every operand of a GEMM once --
 * A (the NHWC activation once for the implicit convolution), W, bias, C, the fp32 residual.  What `roofline.traffic` (PMC) is
 * compared with. */
int rvb_get_timing_bytes(rvb_engine* e, const char* name, double* bytes);

/* Word error counts of a hypothesis against a reference, both as word-id sequences (host code, no GPU): minimal Levenshtein
 * alignment, counts = {errors, substitutions, deletions, insertions} -- the numbers `fstalign wer` logs as bestWER and
 * asr/wer_evaluation/aggregate_scoring.py:37-44 sums (reverb_amd/wer_evaluation/align.py is the caller). */
int rvb_wer_counts(const int32_t* ref, int64_t n_ref, const int32_t* hyp, int64_t n_hyp, int64_t* counts /* [4] */);

#ifdef __cplusplus
}
#endif
#endif /* RVB_H_ */
