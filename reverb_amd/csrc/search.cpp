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
#include <unordered_map>

namespace rvb {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

static inline double log_add2(double a, double b) {
  if (a == NEG_INF && b == NEG_INF) return NEG_INF;
  const double mx = a > b ? a : b;
  return mx + std::log(std::exp(a - mx) + std::exp(b - mx));
}

namespace {
struct PS {
  double s = NEG_INF, ns = NEG_INF, v_s = NEG_INF, v_ns = NEG_INF, cur_token_prob = NEG_INF;
  std::vector<int> times_s, times_ns;
  double score() const { return log_add2(s, ns); }
  double viterbi() const { return v_s > v_ns ? v_s : v_ns; }
  const std::vector<int>& times() const { return v_s > v_ns ? times_s : times_ns; }
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
  std::vector<int> parent(1, -1), last(1, -1);
  std::unordered_map<uint64_t, int> child;
  child.reserve(4096);
  auto extend = [&](int id, int tok) -> int {
    const uint64_t key = ((uint64_t)(uint32_t)id << 32) | (uint32_t)tok;
    auto it = child.find(key);
    if (it != child.end()) return it->second;
    const int nid = (int)parent.size();
    parent.push_back(id);
    last.push_back(tok);
    child.emplace(key, nid);
    return nid;
  };

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
                n1.times_ns = ps.times_ns;
                if (!n1.times_ns.empty()) n1.times_ns.back() = t;
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
            n2.times_ns = ps.times_s;
            n2.times_ns.push_back(t);
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
            n.times_ns = ps.times();
            n.times_ns.push_back(t);
          }
        }
      }
    }
    for (auto& h : nxt) h.score_cache = h.ps.score();
    std::stable_sort(nxt.begin(), nxt.end(),
                     [](const Hyp& a, const Hyp& b) { return a.score_cache > b.score_cache; });
    if ((int)nxt.size() > beam) nxt.resize(beam);
    cur.swap(nxt);
  }

  out->nbest.clear(); out->scores.clear(); out->times.clear();
  for (auto& h : cur) {
    std::vector<int> toks;
    for (int id = h.id; id > 0; id = parent[id]) toks.push_back(last[id]);
    std::reverse(toks.begin(), toks.end());
    out->nbest.push_back(std::move(toks));
    out->scores.push_back(h.ps.score());
    out->times.push_back(h.ps.times());
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
