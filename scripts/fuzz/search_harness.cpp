// ASan / UBSan run of the host CTC search code (csrc/search.cpp) on random device-shaped inputs.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../reverb_amd/csrc/search.h"
using namespace rvb;
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 300;
  std::mt19937 rng(7);
  long toks = 0, jt = 0;
  for (int it = 0; it < rounds; ++it) {
    const int V = 20 + rng() % 200, T = rng() % 140, K = 1 + rng() % 16, beam = 1 + rng() % K;
    std::vector<float> tv((size_t)T * K);
    std::vector<int> ti((size_t)T * K), top1(T);
    for (int t = 0; t < T; ++t) {
      // descending log-probs over distinct tokens, blank (0) favoured like a CTC head
      std::vector<int> ids(V);
      for (int v = 0; v < V; ++v) ids[v] = v;
      for (int k = 0; k < K; ++k) { int j = k + rng() % (V - k); std::swap(ids[k], ids[j]); }
      if (rng() % 3) { for (int k = 0; k < K; ++k) if (ids[k] == 0) std::swap(ids[0], ids[k]); if (ids[0] != 0 && K > 0) ids[0] = 0; }
      float lp = -0.01f * (1 + rng() % 50);
      for (int k = 0; k < K; ++k) { tv[(size_t)t * K + k] = lp; ti[(size_t)t * K + k] = ids[k]; lp -= 0.1f * (1 + rng() % 30); }
      top1[t] = ids[0];
    }
    PrefixResult pr;
    prefix_beam_search(tv.data(), ti.data(), T, K, beam, 0, &pr);
    for (auto& h : pr.nbest) toks += (long)h.size();
    std::vector<int> g, fr;
    greedy_collapse(top1.data(), T, 1, 0, &g, &fr);
    // joint search with a made-up attention model
    JointParams jp;
    jp.beam = 1 + rng() % 5; jp.pre_beam = std::min(K, jp.beam + (int)(rng() % 3)); jp.blank = 0; jp.sos = V - 1;
    jp.w_ctc = 0.1 * (1 + rng() % 9); jp.w_dec = 1.0 - jp.w_ctc; jp.bonus = 0.1 * (rng() % 60);
    if (jp.pre_beam < 1) continue;
    JointSearch js(jp);
    std::vector<int> dec, pn, pt;
    for (int t = 0; t < T; ++t) {
      dec.clear(); pn.clear(); pt.clear();
      float p0 = -20.f, pb = -20.f;
      for (int k = 0; k < K; ++k) if (ti[(size_t)t * K + k] == 0) p0 = pb = tv[(size_t)t * K + k];
      if (!js.begin_frame(t, tv.data() + (size_t)t * K, ti.data() + (size_t)t * K, K, p0, pb, &dec, &pn, &pt)) continue;
      for (int n : dec) js.set_tag(n, n);
      std::vector<float> vals(pn.size());
      for (size_t i = 0; i < vals.size(); ++i) vals[i] = -0.05f * (1 + (unsigned)(pn[i] * 31 + pt[i] * 17) % 90);
      js.finish_frame(vals.data());
    }
    JointResult jr;
    js.result(&jr);
    jt += (long)jr.tokens.size();
  }
  printf("prefix tokens %ld, joint tokens %ld\n", toks, jt);
  return 0;
}
