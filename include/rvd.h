/* rvd.h -- C ABI of the MI355X-native diarization hot path (same shared library as rvb.h: librvb.so).
 *
 * What it replaces.  /root/reference/diarization/infer_pyannote3.0.py:33-42 builds
 * `pyannote.audio.Pipeline.from_pretrained("Revai/reverb-diarization-v1")`, moves it to the GPU and
 * calls `pipeline(audio)`.  Inside that call the two neural networks are the hot path:
 *
 *   segmentation   PyanNet (SincNet + 4 x BiLSTM(128) + 2 x Linear(128) + 7-class powerset head) on every
 *                  10 s window, hop 1 s        (pyannote.audio Inference.slide -> model.forward)
 *   embedding      WeSpeaker ResNet34 + masked statistics pooling on every (window, local speaker) pair
 *
 * The glue between them (powerset decoding, speaker counting, clustering, reconstruction, RTTM) is host
 * code in reverb_amd/diarization.py.  A maintainer binds this header with ctypes exactly as
 * reverb_amd/_lib.py does (INTEGRATION.md shows the pyannote-side stub).
 *
 * Conventions: every function returns 0 on success or a negative RVB_E_* code (rvb.h); the message is in
 * rvd_last_error().  Host pointers are plain C arrays.  All tensors are loaded as fp32 in the layout and
 * under the names of the pyannote checkpoints' state dicts, prefixed "segmentation." / "embedding.".
 * There is no CPU fallback: rvd_create fails when no HIP device is present.
 */
#ifndef RVD_H
#define RVD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvd_engine rvd_engine;

typedef struct rvd_model_cfg {
  int32_t struct_size;      /* sizeof(rvd_model_cfg) in the caller's binding; rvd_create refuses any other value (rvb.h) */
  int32_t dtype;            /* RVB_F32 (0) exact-parity mode, RVB_BF16 (1) bf16 MFMA inputs, fp32 accumulate, RVB_FP8 (2) = the
                               bf16 engine with the 3x3 convolutions of the embedding trunk's stages 3-4 on e4m3 operands
                               (BASELINE configs[4]; scales calibrated by the first rvd_embed call, see rvd_get_emb_fp8) */
  int32_t sample_rate;      /* 16000 */
  int32_t window_samples;   /* 160000 (10 s) */
  int32_t step_samples;     /* 16000  (segmentation_step 0.1 x window) */
  int32_t sinc_filters;     /* 80 (40 cos + 40 sin), kernel 251, stride 10 */
  int32_t sinc_channels;    /* 60: Conv1d(80,60,5), Conv1d(60,60,5) */
  int32_t lstm_hidden;      /* 128 (bidirectional) */
  int32_t lstm_layers;      /* 4 */
  int32_t linear_dim;       /* 128 */
  int32_t linear_layers;    /* 2 */
  int32_t num_classes;      /* 7 powerset classes (3 speakers, <= 2 simultaneously) */
  int32_t emb_channels;     /* 32: ResNet34 m_channels; 0 = no embedding model loaded */
  int32_t emb_dim;          /* 256 */
} rvd_model_cfg;

const char* rvd_last_error(void);
int rvd_model_cfg_size(void);   /* sizeof(rvd_model_cfg) of this build */
/* distinct windows one pass of the embedding trunk takes (768 unless RVD_EMB_BATCH says otherwise; halved by rvd_embed when the
 * activations of that many windows -- 35 MB each in bf16 -- cannot be allocated): a host that prepares pooling masks underneath
 * the first pass (reverb_amd/diarization.py) needs the masks of exactly this many windows up front */
int rvd_emb_windows_per_pass(rvd_engine* e);

int rvd_create(const rvd_model_cfg* cfg, int device, rvd_engine** out);
void rvd_destroy(rvd_engine* e);
/* stage one fp32 tensor of a checkpoint, e.g. "segmentation.sincnet.conv1d.1.weight" */
int rvd_load_tensor(rvd_engine* e, const char* name, const float* host, const int64_t* shape, int ndim);
/* pack weights for the device (sinc filters from low_hz_/band_hz_, conv weights as GEMM operands,
 * BatchNorm folded into the ResNet convolutions) and free the staged copies */
int rvd_finalize(rvd_engine* e);

/* windows pyannote's Inference.slide makes of n samples: the full ones plus one zero-padded tail */
int64_t rvd_num_windows(const rvd_engine* e, int64_t n_samples);
/* frames the segmentation model emits per window (589 for 10 s) */
int rvd_frames_per_window(const rvd_engine* e);

/* 16-bit mono PCM at cfg.sample_rate -> HBM; evaluates the sinc filter bank once for all windows */
int rvd_upload_pcm(rvd_engine* e, const int16_t* pcm, int64_t n_samples);
/* the recording of the last rvd_upload_pcm is still resident in HBM: run its front end again (everything rvd_upload_pcm does
 * behind the copy -- waveform, sinc filter bank, embedding fbank).  For callers that keep a recording on the device and for
 * bench_diar.py, whose timed step starts from HBM-resident samples as the ASR bench's does (the reference has no counterpart:
 * pyannote re-reads the file, diarization/infer_pyannote3.0.py:37). */
int rvd_rerun_resident(rvd_engine* e);
/* pyannote's `Audio` resamples every file to the model's rate (torchaudio.functional.resample with its defaults = the kernel of
 * the ASR front end, rvb_upload_pcm_rate): int16 mono PCM at `sample_rate` -> int16 at cfg.sample_rate, on the device.
 * out == NULL: only *n_out is written; otherwise *n_out holds the capacity of `out` on entry, the length on return. */
int rvd_resample_pcm(rvd_engine* e, const int16_t* pcm, int64_t n_samples, int sample_rate, int16_t* out, int64_t* n_out);

/* segmentation model on windows [first, first+n): logp_out host fp32 [n][frames][num_classes] (NULL: keep the
 * result on the device only) */
int rvd_segment(rvd_engine* e, int64_t first_window, int n_windows, float* logp_out);
/* argmax powerset class of every frame of the last rvd_segment (what Powerset.to_multilabel(soft=False) needs):
 * out host uint8 [n][frames] */
int rvd_get_classes(rvd_engine* e, uint8_t* out);
/* debug taps of the last rvd_segment: "sincnet" [n][frames][sinc_channels], "lstm" [n][frames][2*hidden] */
int rvd_get_tap(rvd_engine* e, const char* name, float* out);

/* embedding model: for each of n items, window index win[i] and a frame mask[i][frames] (weights in [0,1],
 * resampled to the ResNet's time axis as pyannote does); emb_out host fp32 [n][emb_dim] */
int rvd_embed(rvd_engine* e, const int64_t* win, const float* mask, int n, float* emb_out);
/* debug: the 80-bin log-mel features of one window as fed to the ResNet, [frames][80]; returns frames in *n */
int rvd_get_emb_fbank(rvd_engine* e, int64_t window, float* out, int32_t* n_frames);

/* clustering step of the pipeline: the dendrogram scipy.cluster.hierarchy.linkage(X, method="centroid",
 * metric="euclidean") returns for X host fp64 [n][d] -- Z host fp64 [n-1][4] (id_a, id_b, distance, size),
 * rows in merge order, cluster n+k created by row k.  fp64 throughout. */
int rvd_centroid_linkage(rvd_engine* e, const double* X, int n, int d, double* Z);
/* How many workgroups the merge loop of rvd_centroid_linkage may take (it is persistent: its workgroups keep their CUs for the
 * whole clustering): 0 = default (16 workgroups of one XCD from 3 000 points on), 1 = one workgroup, 2 / 4 / 8 / 16.  The
 * dendrogram does not depend on it.  A pipeline that clusters UNDERNEATH another engine's kernels (transcribe_diarize: the ASR
 * encoder) asks for 4: measured on one hour, joint step 472 ms with 4, 477 with 1, 486 with 8, 493-499 with 16. */
int rvd_set_linkage_workgroups(rvd_engine* e, int workgroups);
/* Round 4 CANDIDATE (compiled, not yet run on a GPU): with RVD_EMB_FP8=1 in the environment at rvd_finalize, stages 3-4 of the ResNet34
 * trunk (the MFMA-bound half of the embedding network) run on e4m3 operands (csrc/conv_gemm.hip conv_igemm8_kernel): weights scaled
 * per output channel at load, activations per tensor at scales calibrated by the FIRST trunk pass of the engine (which runs in bf16).
 * *state: 0 = off or not calibrated, 1 = calibrating, 2 = active; scales (nullable; *n in: capacity, out: 32) = the activation
 * scales, index ((stage - 2) * 8 + block) * 2 + {0: first convolution's output, 1: block output}; *clipped (nullable) = values that
 * did not fit e4m3 at those scales so far. */
int rvd_get_emb_fp8(rvd_engine* e, int32_t* state, float* scales, int32_t* n, uint32_t* clipped);
/* Install the 32 activation scales (layout as rvd_get_emb_fp8) and make the fp8 path active at once: every later trunk pass runs on
 * e4m3 operands at these scales, no bf16 calibration pass.  How the ranks of a sharded run agree on ONE quantisation of the
 * recording (reverb_amd/dist.py diarize_sharded: each rank calibrates on its windows, the element-wise maximum is installed
 * everywhere and the windows are embedded again), as rvb_set_fp8_scales does on the ASR side.  The clip counter restarts. */
int rvd_set_emb_fp8_scales(rvd_engine* e, const float* scales, int32_t n);

int rvd_set_profiling(rvd_engine* e, int enabled);
int rvd_reset_timings(rvd_engine* e);
int rvd_get_timing(rvd_engine* e, const char* name, double* ms, double* flops, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
