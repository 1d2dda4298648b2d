// Host-side CTC search (see search.cpp).
#pragma once
#include <stdint.h>
#include <vector>

namespace rvb {

struct PrefixResult {
  std::vector<std::vector<int>> nbest;   // best first
  std::vector<double> scores;            // log_add(s, ns) of each prefix
  std::vector<std::vector<int>> times;   // Viterbi peak frame per token
};

// tv/ti: per-frame top-`beam` log-probs / token ids in torch.topk order, row stride `kstride`.
void prefix_beam_search(const float* tv, const int* ti, int T, int kstride, int beam, int blank,
                        PrefixResult* out);

// top1: best token per frame (stride between frames), T_valid frames.
void greedy_collapse(const int* top1, int T_valid, int stride, int blank, std::vector<int>* tokens,
                     std::vector<int>* frames);

// ---- `joint_decoding`: time-synchronous joint CTC / attention beam search (transformer/search.py:450-496,
// espnet/beam_search_timesync.py:86-508) for ONE chunk, as a state machine the engine drives frame by frame: the CTC half of
// a frame (begin_frame) runs on the host and says which prefixes the attention decoder has to be run on and which
// (prefix, next token) log-probabilities it needs; the caller computes them (one batched decoder step on the GPU for all
// chunks) and finish_frame does the joint scoring and the pruning.  A prefix is a node of a per-chunk trie; every float64
// operation, dict insertion order and tie rule of the reference's Python is kept (search.cpp).
struct JointParams {
  int beam = 4, pre_beam = 6, blank = 0, sos = 0;
  double w_ctc = 0.5, w_dec = 0.5, bonus = 0.5, log_thr = 0.0;      // blank_threshold = 1 -> log 1
};
struct JointResult {
  std::vector<int> tokens, times, end_times;
  std::vector<double> tokens_confidence;
  double score = 0.0;
};
class JointSearch {
 public:
  explicit JointSearch(const JointParams& p);
  // CTC half of frame t.  tv / ti: the frame's top-K log-probs / ids (descending, K >= pre_beam); p_tok0: log-prob of token 0
  // (the reference's blank-skip test reads p_ctc[0]); p_blank: log-prob of the blank.  Appends to `decode` the nodes whose
  // decoder row is needed and not there yet, to `pair_node` / `pair_tok` the (decoded-or-about-to-be node, token) pairs whose
  // attention log-prob is needed.  Returns false when the frame is skipped (nothing to finish).
  bool begin_frame(int t, const float* tv, const int* ti, int K, float p_tok0, float p_blank, std::vector<int>* decode,
                   std::vector<int>* pair_node, std::vector<int>* pair_tok);
  // vals[i] = log p(pair_tok[i] | prefix pair_node[i]) for the pairs of the matching begin_frame, in order
  void finish_frame(const float* vals);
  void prefix(int node, std::vector<int>* toks) const;     // tokens of a node's prefix, <sos> first
  int parent(int node) const { return par_[node]; }
  int length(int node) const { return nodes_[node].len; }
  int token(int node) const { return tok_[node]; }
  int n_nodes() const { return (int)nodes_.size(); }
  void result(JointResult* out) const;
  // engine-side tag of a node (device row of its decoder state), -1 = not decoded
  int tag(int node) const { return tag_[node]; }
  void set_tag(int node, int tag) { tag_[node] = tag; }
  // some frame's kept top-K list ended inside a run of exact ties with the pre-beam threshold (see begin_frame)
  bool ties_cut() const { return ties_cut_; }

 private:
  struct Node {
    int parent, tok, len;
    bool has_times = false, has_conf = false, decoded = false, att_known = false;
    int in_hyps = 0;             // 1 while the node is in the beam
    int new_stamp = -1;          // frame in which the node entered new_hyps
    int dp_stamp = -1, nxt_stamp = -1;
    double dp_nb = 0, dp_b = 0, nxt_nb = 0, nxt_b = 0;     // ctc_score_dp / ctc_score_dp_next entries (valid when stamped)
    double att_tok = 0.0;        // log p_att(tok | parent prefix)
    double log_sum = 0.0;        // log p_att(prefix) (set when the node is decoded)
    double score = 0.0;
    // The reference copies a prefix's start / end / confidence LISTS when it extends the prefix (times[h][0] + [t], ...) and
    // later mutates only the last element of a list (end frame, confidences of the prefix's own last token).  So a
    // prefix's list = a snapshot of its parent's list at the moment of the extension + its own entry: the node keeps its
    // own (mutable) entry and the snapshot of the parent's own entry; older elements are the ancestors' snapshots.  O(1) per
    // extension instead of an O(length) copy, same values.
    int st = 0, en = 0, st_inh = 0, en_inh = 0;
    double conf_ctc = 0, conf_att = 0, conf_ctc_inh = 0, conf_att_inh = 0;
  };
  int child(int node, int tok);
  JointParams p_;
  bool ties_cut_ = false;
  std::vector<Node> nodes_;
  std::vector<int> par_, tok_, tag_;     // compact copies for the engine's path walks (a Node is ~150 bytes)
  std::vector<std::vector<std::pair<int, int>>> child_;     // per node: (token, child id)
  std::vector<int> hyps_, new_hyps_, scored_, touched_;
  std::vector<int> pend_decode_, pend_node_, pend_tok_;
  int frame_ = -1, dp_frame_ = 0;
  bool any_scores_ = false;
};

// counts = {errors, substitutions, deletions, insertions} of the minimal word alignment of hyp against ref
void edit_counts(const int32_t* ref, int64_t n, const int32_t* hyp, int64_t m, int64_t counts[4]);

}  // namespace rvb
