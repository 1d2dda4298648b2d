// Native CTC search for librvb (host side, exact float64 semantics of the reference).
//
//   prefix_beam_search  <- ctc_prefix_beam_search, asr/wenet/transformer/search.py:124-248
//                          PrefixScore :61-103, log_add asr/wenet/utils/common.py:355-363
//   greedy_collapse     <- ctc_greedy_search :106-121 + remove_duplicates_and_blank
//                          asr/wenet/utils/ctc_utils.py:22-32
//
// The reference walks Python dicts keyed by token tuples; here a prefix is a node id in a per-utterance
// trie (parent id + last token) and the candidates of a frame need no dictionary at all: the pair
// (beam entry, top-k token) is visited exactly once per frame, so an extension is a fresh candidate
// unless the extended prefix is itself in the beam (checked against the <= beam children that are).
// Trie nodes are only created for the candidates that survive the pruning.  The iteration order
// (top-k tokens outer, current beam inner), the insertion order of new prefixes, the stable descending
// sort and every float64 operation are kept identical -- including the `vs_ns` typo at search.py:178
// that leaves the Viterbi non-blank score of a repeated token un-updated.
#include "search.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rvb {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

static inline double log_add2(double a, double b) {
  // exact shortcuts: with one operand at -inf the reference computes max + log(0 + 1) = max + 0.0;
  // otherwise one of its two exponentials is exp(0.0) == 1.0 exactly
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  return a > b ? a + std::log(1.0 + std::exp(b - a)) : b + std::log(std::exp(a - b) + 1.0);
}

namespace {
// The reference copies Python lists of peak frames on every update; here a list is an index into a
// per-utterance arena of (parent, frame) nodes, so "copy", "copy and append" and "copy and replace
// the last element" are O(1) and share their common prefix.  -1 is the empty list.
struct PS {
  double s = NEG_INF, ns = NEG_INF, v_s = NEG_INF, v_ns = NEG_INF, cur_token_prob = NEG_INF;
  int times_s = -1, times_ns = -1;
  double score() const { return log_add2(s, ns); }
  double viterbi() const { return v_s > v_ns ? v_s : v_ns; }
  int times() const { return v_s > v_ns ? times_s : times_ns; }
};
struct Hyp {
  int id;        // trie node of the prefix, -1 while it is a candidate that has no node yet
  int par, tok;  // ... in which case it is `par` extended by `tok`
  PS ps;
  double score_cache;  // ps.score(), computed once per frame (pure function of ps)
};
struct Kid { int tok, hj; };
}  // namespace

void prefix_beam_search(const float* tv, const int* ti, int T, int kstride, int beam, int blank,
                        PrefixResult* out) {
  // trie of the prefixes that have been in the beam: node 0 = empty prefix; the children of a node
  // form a singly linked sibling list (short: only survivors get nodes)
  std::vector<int> parent(1, -1), last(1, -1), first_child(1, -1), next_sib(1, -1);
  auto extend = [&](int id, int tok) -> int {
    for (int c = first_child[id]; c >= 0; c = next_sib[c])
      if (last[c] == tok) return c;
    const int nid = (int)parent.size();
    parent.push_back(id);
    last.push_back(tok);
    first_child.push_back(-1);
    next_sib.push_back(first_child[id]);
    first_child[id] = nid;
    return nid;
  };

  std::vector<int> tn_parent, tn_val;   // arena of time-list nodes
  tn_parent.reserve(8192); tn_val.reserve(8192);
  auto t_push = [&](int list, int v) -> int {
    tn_parent.push_back(list); tn_val.push_back(v);
    return (int)tn_parent.size() - 1;
  };

  const size_t cap = (size_t)beam * (beam + 1);      // every beam entry itself + one extension per top-k token
  std::vector<Hyp> cur, nxt;
  cur.reserve(beam); nxt.reserve(cap);                // no reallocation: references into nxt stay valid
  std::vector<int> self_slot(beam), order;
  std::vector<std::vector<Kid>> kids(beam);
  order.reserve(cap);
  cur.emplace_back();
  cur[0].id = 0; cur[0].par = -1; cur[0].tok = -1;
  cur[0].ps.s = 0.0; cur[0].ps.ns = NEG_INF; cur[0].ps.v_s = 0.0; cur[0].ps.v_ns = 0.0;

  for (int t = 0; t < T; ++t) {
    const int nc = (int)cur.size();
    nxt.clear();
    for (int hi = 0; hi < nc; ++hi) {
      cur[hi].score_cache = cur[hi].ps.score();
      self_slot[hi] = -1;
      kids[hi].clear();
    }
    for (int hj = 0; hj < nc; ++hj) {                 // which beam entries extend another beam entry by one token
      const int pj = parent[cur[hj].id];
      if (pj < 0) continue;
      for (int hi = 0; hi < nc; ++hi)
        if (cur[hi].id == pj) { kids[hi].push_back({last[cur[hj].id], hj}); break; }
    }
    auto self_of = [&](int hi) -> PS& {               // next_hyps[prefix]
      if (self_slot[hi] < 0) {
        self_slot[hi] = (int)nxt.size();
        nxt.emplace_back();
        nxt.back().id = cur[hi].id;
      }
      return nxt[self_slot[hi]].ps;
    };
    auto child_of = [&](int hi, int u) -> PS& {       // next_hyps[prefix + (u,)]
      for (const Kid& k : kids[hi])
        if (k.tok == u) return self_of(k.hj);
      nxt.emplace_back();
      Hyp& h = nxt.back();
      h.id = -1; h.par = cur[hi].id; h.tok = u;
      return h.ps;
    };

    for (int kk = 0; kk < beam; ++kk) {
      const int u = ti[(size_t)t * kstride + kk];
      const double prob = (double)tv[(size_t)t * kstride + kk];
      for (int hi = 0; hi < nc; ++hi) {
        const PS& ps = cur[hi].ps;
        const double sc = cur[hi].score_cache;
        if (u == blank) {
          PS& n = self_of(hi);
          n.s = log_add2(n.s, sc + prob);
          n.v_s = ps.viterbi() + prob;
          n.times_s = ps.times();
        } else if (u == last[cur[hi].id]) {
          PS& n1 = self_of(hi);
          n1.ns = log_add2(n1.ns, ps.ns + prob);
          if (n1.v_ns < ps.v_ns + prob) {
            // reference assigns a misspelled attribute here (`vs_ns`): v_ns stays as it was
            if (n1.cur_token_prob < prob) {
              n1.cur_token_prob = prob;
              // copy of the list with its last element replaced by t
              n1.times_ns = ps.times_ns >= 0 ? t_push(tn_parent[ps.times_ns], t) : -1;
            }
          }
          PS& n2 = child_of(hi, u);
          n2.ns = log_add2(n2.ns, ps.s + prob);
          if (n2.v_ns < ps.v_s + prob) {
            n2.v_ns = ps.v_s + prob;
            n2.cur_token_prob = prob;
            n2.times_ns = t_push(ps.times_s, t);
          }
        } else {
          PS& n = child_of(hi, u);
          n.ns = log_add2(n.ns, sc + prob);
          const double vit = ps.viterbi() + prob;
          if (n.v_ns < vit) {
            n.v_ns = vit;
            n.cur_token_prob = prob;
            n.times_ns = t_push(ps.times(), t);
          }
        }
      }
    }
    // sorted(..., reverse=True)[:beam] of the reference: descending score, equal scores keep their
    // insertion order (= index in nxt).
    const int nn = (int)nxt.size();
    order.resize(nn);
    for (int i = 0; i < nn; ++i) { nxt[i].score_cache = nxt[i].ps.score(); order[i] = i; }
    const int keep = std::min<int>(beam, nn);
    std::partial_sort(order.begin(), order.begin() + keep, order.end(), [&](int a, int b) {
      const double sa = nxt[a].score_cache, sb = nxt[b].score_cache;
      return sa > sb || (sa == sb && a < b);
    });
    cur.clear();
    for (int k = 0; k < keep; ++k) {
      cur.push_back(nxt[order[k]]);
      Hyp& h = cur.back();
      if (h.id < 0) h.id = extend(h.par, h.tok);      // survivors get (or find again) their trie node
    }
  }

  out->nbest.clear(); out->scores.clear(); out->times.clear();
  for (auto& h : cur) {
    std::vector<int> toks;
    for (int id = h.id; id > 0; id = parent[id]) toks.push_back(last[id]);
    std::reverse(toks.begin(), toks.end());
    out->nbest.push_back(std::move(toks));
    out->scores.push_back(h.ps.score());
    std::vector<int> tm;
    for (int n = h.ps.times(); n >= 0; n = tn_parent[n]) tm.push_back(tn_val[n]);
    std::reverse(tm.begin(), tm.end());
    out->times.push_back(std::move(tm));
  }
}

void greedy_collapse(const int* top1, int T_valid, int stride, int blank, std::vector<int>* tokens,
                     std::vector<int>* frames) {
  tokens->clear();
  frames->clear();
  int prev = -1;
  for (int t = 0; t < T_valid; ++t) {
    const int u = top1[(size_t)t * stride];
    if (u != prev) {
      if (u != blank) { tokens->push_back(u); frames->push_back(t); }
      prev = u;
    }
  }
}

// Word-level Levenshtein alignment of a hypothesis against a reference (what `fstalign wer` reports as bestWER for
// plain token sequences; asr/wer_evaluation/README.md): minimum number of edits, and among minimal alignments the one a
// fixed preference order (match/substitution, then deletion, then insertion) reaches, so that the S / D / I split is
// deterministic.  O(n*m) time, O(m) memory: every cell carries its own (cost, S, D, I).
void edit_counts(const int32_t* ref, int64_t n, const int32_t* hyp, int64_t m, int64_t counts[4]) {
  struct Cell { int64_t c, s, d, i; };
  std::vector<Cell> prev((size_t)m + 1), cur((size_t)m + 1);
  for (int64_t j = 0; j <= m; ++j) prev[j] = {j, 0, 0, j};
  for (int64_t i = 1; i <= n; ++i) {
    cur[0] = {i, 0, i, 0};
    for (int64_t j = 1; j <= m; ++j) {
      const bool eq = ref[i - 1] == hyp[j - 1];
      Cell best = prev[j - 1];
      if (!eq) { best.c += 1; best.s += 1; }
      const Cell& up = prev[j];          // reference word i has no counterpart: deletion
      if (up.c + 1 < best.c) { best = up; best.c += 1; best.d += 1; }
      const Cell& left = cur[j - 1];     // hypothesis word j has no counterpart: insertion
      if (left.c + 1 < best.c) { best = left; best.c += 1; best.i += 1; }
      cur[j] = best;
    }
    prev.swap(cur);
  }
  counts[0] = prev[m].c; counts[1] = prev[m].s; counts[2] = prev[m].d; counts[3] = prev[m].i;
}

// ================================================================================================
// joint_decoding: BeamSearchTimeSync (espnet/beam_search_timesync.py) for one chunk, see search.h
// ================================================================================================
namespace {
// beam_search_timesync.py:29-37, python floats: max + log(sum(exp(a - max)))
inline double lse2(double a, double b) {
  if (a == NEG_INF && b == NEG_INF) return NEG_INF;
  const double m = a > b ? a : b;
  return m + std::log(std::exp(a - m) + std::exp(b - m));
}
}  // namespace

JointSearch::JointSearch(const JointParams& p) : p_(p) {
  Node root;
  root.parent = -1; root.tok = p.sos; root.len = 1;
  root.has_times = root.has_conf = true;
  root.st = 0; root.en = 0;
  root.conf_ctc = NEG_INF; root.conf_att = NEG_INF;
  root.dp_stamp = 0; root.dp_nb = NEG_INF; root.dp_b = 0.0;          // ctc_score_dp[[sos]] = (-inf, 0.0)
  root.in_hyps = 1;
  root.decoded = true; root.log_sum = 0.0;                           // reset() runs the decoder on <sos>
  nodes_.push_back(root);
  par_.push_back(-1); tok_.push_back(p.sos); tag_.push_back(-1);
  child_.emplace_back();
  hyps_.assign(1, 0);
}

int JointSearch::child(int node, int tok) {
  for (const auto& c : child_[node]) if (c.first == tok) return c.second;
  Node n;
  n.parent = node; n.tok = tok; n.len = nodes_[node].len + 1;
  nodes_.push_back(n);
  par_.push_back(node); tok_.push_back(tok); tag_.push_back(-1);
  child_.emplace_back();
  const int id = (int)nodes_.size() - 1;
  child_[node].push_back({tok, id});
  return id;
}

void JointSearch::prefix(int node, std::vector<int>* toks) const {
  toks->assign(nodes_[node].len, 0);
  for (int n = node, i = nodes_[node].len - 1; n >= 0; n = par_[n], --i) (*toks)[i] = tok_[n];
}

bool JointSearch::begin_frame(int t, const float* tv, const int* ti, int K, float p_tok0, float p_blank, std::vector<int>* decode,
                              std::vector<int>* pair_node, std::vector<int>* pair_tok) {
  // :271-274 -- `torch.argmax(p_ctc[0])` is the argmax of a 0-d tensor, i.e. always token 0
  if (0 == p_.blank && (double)p_tok0 >= p_.log_thr) return false;
  frame_ = t;
  const int cur = dp_frame_, nxt = t + 1;                    // stamps: dp entries carry `cur`, next-frame entries `nxt`
  // candidates: everything >= the pre_beam-th largest log-prob, in token order (nonzero())
  const int pb = p_.pre_beam < K ? p_.pre_beam : K;
  const float thr = tv[pb - 1];
  // the reference compares the WHOLE row with the threshold (`ctc_probs >= thr`): log-probs that tie with it exactly are
  // candidates too.  The kept list runs past the pre-beam (the host asks for pre_beam + 8); if even its last entry ties, the
  // run of ties may go on beyond what was kept -- reported, never silently cut (ties_cut_; K == vocabulary keeps everything)
  if (K > pb && tv[K - 1] >= thr) ties_cut_ = true;
  std::vector<std::pair<int, double>> cands;
  for (int k = 0; k < K; ++k) if (tv[k] >= thr) cands.push_back({ti[k], (double)tv[k]});
  std::sort(cands.begin(), cands.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
  new_hyps_.clear();
  auto nxt_get = [&](Node& n, double& nb, double& b) { if (n.nxt_stamp == nxt) { nb = n.nxt_nb; b = n.nxt_b; } else { nb = NEG_INF; b = NEG_INF; } };
  touched_.clear();
  auto nxt_set = [&](Node& n, double nb, double b) {
    if (n.nxt_stamp != nxt) touched_.push_back((int)(&n - nodes_.data()));
    n.nxt_stamp = nxt; n.nxt_nb = nb; n.nxt_b = b;
  };
  auto add_new = [&](int id) { if (nodes_[id].new_stamp != nxt) { nodes_[id].new_stamp = nxt; new_hyps_.push_back(id); } };
  const std::vector<int> hyps = hyps_;
  for (int h : hyps) {
    const double prev = lse2(nodes_[h].dp_nb, nodes_[h].dp_b);         // nodes in the beam always have a dp entry
    for (const auto& cd : cands) {
      const int c = cd.first;
      const double pc = cd.second;
      if (c == p_.blank) {
        double nb, bl;
        nxt_get(nodes_[h], nb, bl);
        nxt_set(nodes_[h], nb, lse2(bl, pc + prev));
        add_new(h);
        continue;
      }
      const int g = child(h, c);                 // may grow nodes_: take references afterwards
      Node& G = nodes_[g];
      Node& H = nodes_[h];
      double nb, bl;
      nxt_get(G, nb, bl);
      if (!G.has_times) {
        G.st_inh = H.st; G.en_inh = H.en;          // times[h][0] + [t], times[h][1] + [t + 1]: h's lists as they are now
        G.st = t; G.en = t + 1;
        G.has_times = true;
      } else {
        G.en = t + 1;
      }
      if (!G.has_conf) {
        G.conf_ctc_inh = H.conf_ctc; G.conf_att_inh = H.conf_att;
        G.conf_ctc = NEG_INF; G.conf_att = NEG_INF;
        G.has_conf = true;
      }
      G.conf_ctc = G.conf_ctc > pc ? G.conf_ctc : pc;
      if (c == H.tok) {                          // repeated token: only through a blank; h itself keeps growing
        const double nb_prev = H.dp_nb, b_prev = H.dp_b;
        nb = lse2(nb, pc + b_prev);
        double hn, hb;
        nxt_get(H, hn, hb);
        nxt_set(H, lse2(hn, pc + nb_prev), hb);
        H.en = t + 1;
        H.conf_ctc = H.conf_ctc > pc ? H.conf_ctc : pc;
      } else {
        nb = lse2(nb, pc + prev);
      }
      if (!G.in_hyps && G.dp_stamp == cur) {     // seen before but pruned: its old mass comes back in
        bl = lse2(bl, (double)p_blank + lse2(G.dp_nb, G.dp_b));
        nb = lse2(nb, pc + G.dp_nb);
      }
      nxt_set(G, nb, bl);
      add_new(g);
    }
  }
  // attention scores needed for the joint score of every candidate (:226-255, cached_score :185-224)
  pend_decode_.clear(); pend_node_.clear(); pend_tok_.clear();
  if (p_.w_dec > 0) {
    for (int h : new_hyps_) {
      Node& H = nodes_[h];
      if (H.len <= 1 || H.att_known) continue;
      Node& R = nodes_[H.parent];
      if (!R.decoded) {
        R.decoded = true;                        // its row exists once the caller has run the decoder step
        R.log_sum = R.parent >= 0 ? nodes_[R.parent].log_sum + R.att_tok : 0.0;
        pend_decode_.push_back(H.parent);
      }
      pend_node_.push_back(H.parent);
      pend_tok_.push_back(H.tok);
    }
  }
  decode->insert(decode->end(), pend_decode_.begin(), pend_decode_.end());
  pair_node->insert(pair_node->end(), pend_node_.begin(), pend_node_.end());
  pair_tok->insert(pair_tok->end(), pend_tok_.begin(), pend_tok_.end());
  return true;
}

void JointSearch::finish_frame(const float* vals) {
  const int nxt = frame_ + 1;
  size_t vi = 0;
  std::vector<std::pair<double, int>> order;     // (score, node) in scoring order
  for (int h : new_hyps_) {
    Node& H = nodes_[h];
    double sc = p_.w_ctc * lse2(H.nxt_nb, H.nxt_b);
    if (H.len > 1 && p_.w_dec > 0) {
      if (!H.att_known) { H.att_tok = (double)vals[vi++]; H.att_known = true; }
      sc += (nodes_[H.parent].log_sum + H.att_tok) * p_.w_dec;
      H.conf_att = H.att_tok;
    }
    sc += p_.bonus * (H.len - 1);
    H.score = sc;
    order.push_back({sc, h});
  }
  // `reverse_dict[score] = hyp` for every scored hypothesis in order (equal scores: the later one wins), descending, top beam
  std::vector<std::pair<double, int>> uniq;
  for (const auto& o : order) {
    bool found = false;
    for (auto& u : uniq) if (u.first == o.first) { u.second = o.second; found = true; break; }
    if (!found) uniq.push_back(o);
  }
  std::stable_sort(uniq.begin(), uniq.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
  for (int h : hyps_) nodes_[h].in_hyps = 0;
  hyps_.clear();
  for (size_t i = 0; i < uniq.size() && (int)i < p_.beam; ++i) { hyps_.push_back(uniq[i].second); nodes_[uniq[i].second].in_hyps = 1; }
  // ctc_score_dp = ctc_score_dp_next.copy(): exactly the entries written this frame (a repeated token also writes the
  // entry of the un-extended hypothesis without putting it among the candidates)
  for (int h : touched_) { Node& H = nodes_[h]; H.dp_stamp = nxt; H.dp_nb = H.nxt_nb; H.dp_b = H.nxt_b; }
  dp_frame_ = nxt;
  scored_ = new_hyps_;
  any_scores_ = true;
}

void JointSearch::result(JointResult* out) const {
  const int best = hyps_[0];
  const Node& B = nodes_[best];
  const int L = B.len;                       // list elements: <sos> + tokens
  std::vector<int> toks, st(L), en(L);
  std::vector<double> cf(L);
  prefix(best, &toks);
  // element L-1 is the node's own entry, element i < L-1 the snapshot kept by the ancestor at depth i+1
  st[L - 1] = B.st; en[L - 1] = B.en;
  cf[L - 1] = B.conf_ctc > B.conf_att ? B.conf_ctc : B.conf_att;
  for (int n = best, i = L - 2; i >= 0; n = nodes_[n].parent, --i) {
    const Node& N = nodes_[n];
    st[i] = N.st_inh; en[i] = N.en_inh;
    cf[i] = N.conf_ctc_inh > N.conf_att_inh ? N.conf_ctc_inh : N.conf_att_inh;
  }
  out->tokens.assign(toks.begin() + 1, toks.end());
  out->times.assign(st.begin() + 1, st.end());
  out->end_times.assign(en.begin() + 1, en.end());
  out->tokens_confidence.clear();
  for (int i = 1; i < L; ++i) out->tokens_confidence.push_back(std::exp(cf[i]));
  out->score = any_scores_ ? B.score : 0.0;
}

}  // namespace rvb
