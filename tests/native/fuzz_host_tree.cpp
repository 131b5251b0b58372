// ASan/UBSan fuzz of the host suffix tree (arena, flat hash, best-child bookkeeping): CPU build only.
#include <cstdio>
#include <cstdlib>
#include <random>
#include "suffix_host.hpp"
int main() {
  std::mt19937 rng(7);
  long checks = 0;
  for (int depth : {3, 8, 64}) {
    for (int vocab : {2, 5, 40}) {
      aic::HostTree t(depth);
      std::vector<std::vector<int>> hist(6);
      for (int step = 0; step < 4000; ++step) {
        const int s = rng() % 6;
        const int n = 1 + rng() % 5;
        for (int i = 0; i < n; ++i) {
          const int tok = rng() % vocab;
          t.append(s, tok);
          hist[s].push_back(tok);
        }
        if (step % 50 == 0) {
          if (t.selfcheck() != 0) { std::printf("selfcheck failed depth %d vocab %d step %d\n", depth, vocab, step); return 1; }
          const auto& h = hist[rng() % 6];
          if (!h.empty()) {
            const int len = 1 + rng() % std::min<size_t>(h.size(), depth + 4);
            std::vector<int32_t> pat(h.end() - len, h.end());
            auto c = t.speculate_tree(pat.data(), len, 1 + rng() % 40, 1.0f + (rng() % 3), float(rng() % 3), 0.05f * (rng() % 5));
            checks += c.token_ids.size() + 1;
          }
        }
      }
      if (t.selfcheck() != 0) { std::printf("final selfcheck failed\n"); return 1; }
    }
  }
  std::printf("ok %ld\n", checks);
  return 0;
}
