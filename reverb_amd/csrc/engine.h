// Engine: weights, workspace and orchestration of the Reverb-ASR hot path on one MI355X.
#pragma once
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rvb.h"
#include "kernels.h"
#include "search.h"

namespace rvb {

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t n);   // grow-only
  void release();
  template <typename T> T* as() const { return (T*)p; }
};

struct Linear {      // y = x.W^T + b, W packed to the compute dtype [out][in]
  DevBuf w, b;
  int out = 0, in = 0;
  DevBuf w8, wscale;   // fp8 mode: e4m3 copy [out][in] and its per-output-channel scales fp32 [out]
};
// per-tensor scales of the activations an fp8 GEMM reads (value = fp8 * scale), one set per conformer block
struct F8Scales { float in_ffm1 = 1, h_ffm = 1, in_qkv = 1, in_pw1 = 1, in_pw2 = 1, in_ff1 = 1, h_ff = 1; };
struct LNorm { DevBuf g, b; float eps = 1e-5f; };

struct EncLayer {
  Linear ffm1, ffm2, ff1, ff2, qkv, att_out, pw1, pw2, lsl;
  Linear pw1_glu;             // bf16 engine: pw1 with its output rows interleaved (a_0, b_0, a_1, b_1, ...) for the GEMM's ACT_GLU epilogue
  DevBuf pos_keys;            // T [Tpos, d] = linear_pos(pe[:Tpos])
  DevBuf pos_bias;            // bf16 engine: fp32 [heads][Tpos], (pos_bias_v - pos_bias_u) . pos_keys * log2(e)/sqrt(dk) (attention.hip FOLD)
  DevBuf bias_u, bias_v;      // fp32 [h*dk]
  LNorm n_ffm, n_mha, n_conv, n_ff, n_final, n_cnn;
  DevBuf dw_w, dw_b;          // fp32 [K][d] (tap-major), [d]
  bool is_lsl = false;
};
struct DecLayer {
  Linear self_qkv, self_out, src_q, src_kv, src_out, ff1, ff2, lsl;
  LNorm n1, n2, n3;
  bool is_lsl = false;
  DevBuf kvmem;               // T [B*T2][2d]: this layer's keys / values of the encoder output (rescoring)
};
struct Decoder {
  DevBuf embed;               // fp32 [V][d]
  Linear out;
  LNorm after;
  std::vector<DecLayer> layers;
  bool present = false;
  bool kv_ready = false;      // the layers' kvmem hold the current batch (rvb_prepare_rescoring or the first decoder pass)
};

struct ProfEntry { double ms = 0, flops = 0, bytes = 0; int64_t launches = 0; };

struct RescoreResult {
  int best = 0;
  float score = 0.f;
  double confidence = 0.0;
  std::vector<double> tok_conf;
  std::vector<std::vector<float>> logp, rlogp;   // per hyp: len+1 decoder log-probs
};


// Host worker threads kept alive across calls (prefix beam search per slice, trie building): starting 31 threads costs
// about as much as the work of the last, un-overlapped slice.  run(n, fn): fn is executed by n threads, the caller among them.
class HostPool {
 public:
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(unsigned n, const std::function<void()>& fn) {
    if (n <= 1) { fn(); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      while (th_.size() + 1 < n) th_.emplace_back([this] { loop(); });
      job_ = &fn; want_ = n - 1; started_ = 0; finished_ = 0; ++gen_;
    }
    cv_.notify_all();
    fn();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return finished_ == want_; });
    job_ = nullptr;
  }

 private:
  void loop() {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || (gen_ != seen && started_ < want_); });
      if (stop_) return;
      seen = gen_;
      ++started_;
      const std::function<void()>* job = job_;
      lk.unlock();
      (*job)();
      lk.lock();
      if (++finished_ == want_) done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void()>* job_ = nullptr;
  unsigned want_ = 0, started_ = 0, finished_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// ---- rescoring: every DISTINCT prefix of a chunk's n-best list is one decoder row (engine.hip, "rescoring")
struct HypRef { int chunk, idx, len, row0; };      // row0: first of the hypothesis' len+1 (hyp, j) pairs

struct TrieBatch {
  int R = 0, P = 0, max_chunk_rows = 0;            // unique rows, (hyp, j) pairs, most rows of one chunk
  std::vector<int32_t> tok, pos;                   // per row: input token, position
  std::vector<int32_t> path;                       // per hypothesis: the rows of its prefixes 0..len (flat)
  std::vector<int32_t> hq_start, hq_len, hq_pos0, hkv_start, hkv_len;   // per hypothesis: owned rows / path
  std::vector<int32_t> crow_start, crow_len;       // per chunk: its rows (contiguous)
  std::vector<int32_t> tgt_ptr, tgt;               // CSR over rows: the targets asked of a row
  std::vector<int32_t> pair_slot;                  // (hyp, j) pair -> position in tgt / in the gathered log-probs
  std::vector<int32_t> work;                       // self-attention blocks: {hypothesis, first owned query}
};

}  // namespace rvb

struct rvb_engine {
  rvb_model_cfg cfg;
  int device = 0;
  int dtype = 0;
  bool fp8 = false;              // RVB_FP8: dtype stays DT_BF16, the encoder's feed-forward / qkv / pointwise GEMMs run in fp8
  int f8_state = 0;              // 0 not calibrated, 1 calibrating (bf16 pass collecting max |.|), 2 fp8 GEMMs active
  std::vector<rvb::F8Scales> f8;
  std::vector<unsigned> f8_groups;   // per block: which GEMM groups run in fp8 (bit 0 ffm, 1 qkv, 2 pw1, 3 pw2, 4 ff)
  rvb::DevBuf d_amax;            // fp32 [blocks + 1][8]; row `blocks`, slot 0: the subsampling's conv1 output (conv2's fp8 operand)
  bool f8_conv2 = false;         // policy bit 5: conv2 of the subsampling (K = 9 d, 27 % of the encoder's FLOPs) in fp8
  float f8_x1 = 0.f;             // calibrated scale of conv1's output (0: not calibrated -> conv2 stays bf16)
  rvb::DevBuf d_f8sat;           // uint32 [blocks + 1][8]: values clipped at +-448 per activation slot (rvb_get_fp8_saturation)
  hipStream_t stream = nullptr;
  bool finalized = false;

  std::map<std::string, rvb::HostTensor> host;   // staged state dict (fp32)

  // ---- packed weights ----
  rvb::DevBuf cmvn_mean, cmvn_istd, conv1_w, conv1_b;
  rvb::Linear conv2, embed_out, ctc;
  rvb::LNorm enc_after;
  std::vector<rvb::EncLayer> enc;
  rvb::Decoder dec_l, dec_r;
  rvb::DevBuf pe_f32;            // fp32 [pe_rows][d] sinusoid table
  int pe_rows = 0;
  rvb::DevBuf fb_window, fb_twiddle, fb_melw, fb_lo, fb_hi;
  rvb::DevBuf stage;             // fp32 staging for weight packing

  // ---- audio / features ----
  rvb::DevBuf pcm, feats;        // int16 [n], fp32 [chunks*T0pad][80]
  // double-buffered upload (rvb_upload_pcm_async): the NEXT recording's samples travel on `copy_stream` into `pcm_next` while
  // the current one is being decoded; the next rvb_fbank waits for `pcm_ready` on the engine's stream and swaps the buffers
  rvb::DevBuf pcm_next;
  hipStream_t copy_stream = nullptr;
  hipEvent_t pcm_ready = nullptr, pcm_free = nullptr;   // pcm_free: the last fbank that read `pcm` has run
  bool pcm_pending = false;
  int64_t n_samples_next = 0;
  rvb::DevBuf wave_f32, rs_kernel; // the waveform the fbank reads when it is not int16 PCM at 16 kHz: resampled and / or uploaded as float
  rvb::DevBuf wave_in;             // a float waveform at another rate, before resampling (rvb_upload_wave_f32)
  bool pcm_is_float = false;
  int dec_chunk = 0, dec_left = -1;   // encoder chunk mask (decoding_chunk_size / num_decoding_left_chunks), 0 = full context
  int64_t n_samples = 0, n_frames = 0, feat_rows = 0;

  // ---- batch state ----
  int B = 0, T0 = 0, T1 = 0, F1 = 0, T2 = 0, F2 = 0, beam = 0;
  std::vector<int32_t> in_lens, enc_lens;
  rvb::DevBuf d_feats_in, X1, X2, x, xn, y, h, ao, dconv, enc_out, logits, topv, topi;
  rvb::DevBuf d_enc_lens, d_seq_start, d_seq_len, d_aux_i32;
  float* h_topv = nullptr;         // pinned host copies of the per-frame top-k (async D2H per slice)
  int32_t* h_topi = nullptr;
  size_t h_top_cap = 0;
  struct Slice { int c0, nb; hipEvent_t ev; bool done; };
  std::vector<Slice> slices;       // slices of the last rvb_encode, in chunk order
  std::vector<hipEvent_t> slice_event_pool;
  int xattn_max_rows = 0;          // most decoder rows of any chunk (cross-attention query sequence length)
  const int* cur_lens = nullptr;   // device pointer: valid encoder frames of the slice being encoded
  std::vector<rvb::PrefixResult> nbest;
  std::vector<rvb::RescoreResult> rescored;
  std::vector<rvb::TrieBatch> trie_l;   // per chunk, local numbering: built by the prefix-beam workers for the rescoring decoder
  rvb::HostPool pool;                   // host workers of the CTC search and the trie building
  std::vector<rvb::JointResult> joint;           // rvb_joint_decode: winner per chunk
  std::vector<rvb::DevBuf> jkv;                   // joint_decoding: per decoder layer T [rows][2d], key | value of every decoded prefix
  rvb::DevBuf jlogp, jpair_row, jpair_tok, jpair_out;     // fp32 [rows][V] log-softmax after each decoded prefix; pair gather buffers
  int64_t joint_rows = 0, joint_steps = 0;       // decoder rows computed / batched decoder steps of the last rvb_joint_decode
  float last_blank_penalty = 0.f;                // of the last rvb_encode / rvb_stream_finish
  std::vector<std::vector<int>> attn_tokens;     // rvb_attention_decode: best hypothesis per chunk
  std::vector<float> attn_scores;
  rvb::DevBuf atopv, atopi;                       // per-step top-k of the decoder output
  std::vector<rvb::DevBuf> kcache, vcache, kcache2, vcache2, memkv;   // per decoder layer
  // decoder workspace
  rvb::DevBuf dx, dxn, dy, dh, dqkv, dq, dao, kvmem, d_tok, d_pos, d_tgt, d_logp;
  rvb::DevBuf d_hq_start, d_hq_len, d_hkv_start, d_hkv_len;
  rvb::DevBuf d_hq_pos0, d_hpath_start, d_hpath_len, d_path, d_work, d_tgt_ptr;    // rescoring over the hypothesis trie
  int64_t rescore_rows = 0, rescore_pairs = 0;      // decoder rows computed / (hypothesis, position) pairs served, last call

  // ---- streaming encoder (forward_chunk with caches, encoder.py:231-402) ----
  struct StreamState {
    bool active = false;
    int offset = 0;                 // encoder frames produced so far (`offset` of forward_chunk)
    int cache_len = 0;              // frames in the attention cache (cache_t1)
    std::vector<rvb::DevBuf> kv, kv2;   // per layer T [pe_rows][2d]: key | value rows of the cached frames (+ spare for trimming)
    int cnn_rows = 0;               // real frames in the cnn cache (cache_t2 grows to lorder = K-1; older = zero padding)
    std::vector<rvb::DevBuf> cnn, cnn2; // causal conv module, per layer T [K-1][2d]: pointwise-conv1 outputs of the last frames
  } stream_st;
  rvb::DevBuf d_stream_i32;         // {kv_start = 0, kv_len = cache + chunk}

  // ---- RCCL communicator of the C-ABI collectives (comm.hip; optional) ----
  void* comm = nullptr;
  int comm_world = 1, comm_rank = 0;
  rvb::DevBuf comm_send, comm_recv;

  // ---- profiling ----
  int profiling = 0;     // 0 off, 1 every stage, 2 GEMM launches only (what the roofline needs; half the events)
  std::map<std::string, rvb::ProfEntry> prof;
  struct Pending { hipEvent_t a, b; std::string name; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};
