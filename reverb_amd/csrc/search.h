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

// counts = {errors, substitutions, deletions, insertions} of the minimal word alignment of hyp against ref
void edit_counts(const int32_t* ref, int64_t n, const int32_t* hyp, int64_t m, int64_t counts[4]);

}  // namespace rvb
