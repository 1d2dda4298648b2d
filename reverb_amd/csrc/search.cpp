// Native CTC search for librvb (host side, exact float64 semantics of the reference).
//
//   prefix_beam_search  <- ctc_prefix_beam_search, asr/wenet/transformer/search.py:124-248
//                          PrefixScore :61-103, log_add asr/wenet/utils/common.py:355-363
//   greedy_collapse     <- ctc_greedy_search :106-121 + remove_duplicates_and_blank
//                          asr/wenet/utils/ctc_utils.py:22-32
//
// The reference walks Python dicts/tuples; here a prefix is a node id in a per-utterance trie
// (parent id + last token) so dictionary lookups become array indexing, while the iteration
// order (top-k tokens outer, current beam inner), the insertion order of new prefixes, the stable
// descending sort and every float64 operation are kept identical -- including the `vs_ns` typo at
// search.py:178 that leaves the Viterbi non-blank score of a repeated token un-updated.
#include "search.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rvb {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

static inline double log_add2(double a, double b) {
  // exact shortcuts: with one operand at -inf the reference computes max + log(0 + 1) = max + 0.0
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  const double mx = a > b ? a : b;
  return mx + std::log(std::exp(a - mx) + std::exp(b - mx));
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
  int id;
  PS ps;
  double score_cache;  // ps.score(), computed once per frame (pure function of ps)
};
}  // namespace

void prefix_beam_search(const float* tv, const int* ti, int T, int kstride, int beam, int blank,
                        PrefixResult* out) {
  // trie of prefixes: node 0 = empty prefix
  // children of a node form a singly linked sibling list (a prefix has few live extensions)
  std::vector<int> parent(1, -1), last(1, -1), first_child(1, -1), next_sib(1, -1);
  parent.reserve(16384); last.reserve(16384); first_child.reserve(16384); next_sib.reserve(16384);
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
  std::vector<int> sel;
  std::vector<char> taken;
  std::vector<Hyp> cur(1), nxt;
  cur[0].id = 0;
  cur[0].ps.s = 0.0; cur[0].ps.ns = NEG_INF; cur[0].ps.v_s = 0.0; cur[0].ps.v_ns = 0.0;
  std::vector<int> slot;        // node id -> index in nxt for the current frame
  std::vector<int> slot_frame;  // frame stamp validating `slot`
  auto next_of = [&](int id, int t) -> PS& {
    if ((int)slot.size() <= id) { slot.resize(id + 64, -1); slot_frame.resize(id + 64, -1); }
    if (slot_frame[id] != t) {
      slot_frame[id] = t;
      slot[id] = (int)nxt.size();
      nxt.emplace_back();
      nxt.back().id = id;
    }
    return nxt[slot[id]].ps;
  };

  for (int t = 0; t < T; ++t) {
    nxt.clear();
    for (auto& h : cur) h.score_cache = h.ps.score();
    for (int kk = 0; kk < beam; ++kk) {
      const int u = ti[(size_t)t * kstride + kk];
      const double prob = (double)tv[(size_t)t * kstride + kk];
      for (size_t hi = 0; hi < cur.size(); ++hi) {
        const int pid = cur[hi].id;
        const double sc = cur[hi].score_cache;
        if (u == blank) {
          const PS& ps = cur[hi].ps;
          PS& n = next_of(pid, t);
          n.s = log_add2(n.s, sc + prob);
          n.v_s = ps.viterbi() + prob;
          n.times_s = ps.times();
        } else if (u == last[pid]) {
          {
            const PS& ps = cur[hi].ps;
            PS& n1 = next_of(pid, t);
            n1.ns = log_add2(n1.ns, ps.ns + prob);
            if (n1.v_ns < ps.v_ns + prob) {
              // reference assigns a misspelled attribute here (`vs_ns`): v_ns stays as it was
              if (n1.cur_token_prob < prob) {
                n1.cur_token_prob = prob;
                // copy of the list with its last element replaced by t
                n1.times_ns = ps.times_ns >= 0 ? t_push(tn_parent[ps.times_ns], t) : -1;
              }
            }
          }
          const int nid = extend(pid, u);
          PS& n2 = next_of(nid, t);          // may reallocate nxt: re-read ps afterwards
          const PS& ps = cur[hi].ps;
          n2.ns = log_add2(n2.ns, ps.s + prob);
          if (n2.v_ns < ps.v_s + prob) {
            n2.v_ns = ps.v_s + prob;
            n2.cur_token_prob = prob;
            n2.times_ns = t_push(ps.times_s, t);
          }
        } else {
          const int nid = extend(pid, u);
          PS& n = next_of(nid, t);
          const PS& ps = cur[hi].ps;
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
    for (auto& h : nxt) h.score_cache = h.ps.score();
    // sorted(..., reverse=True)[:beam] of the reference: descending score, equal scores keep their
    // insertion order.  Only the first `beam` are needed: repeated selection of the first maximum.
    const int keep = std::min<int>(beam, (int)nxt.size());
    sel.clear();
    taken.assign(nxt.size(), 0);
    for (int k = 0; k < keep; ++k) {
      int best = -1;
      for (int i = 0; i < (int)nxt.size(); ++i)
        if (!taken[i] && (best < 0 || nxt[i].score_cache > nxt[best].score_cache)) best = i;
      if (best < 0) break;
      taken[best] = 1;
      sel.push_back(best);
    }
    cur.clear();
    for (int i : sel) cur.push_back(nxt[i]);
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

}  // namespace rvb
