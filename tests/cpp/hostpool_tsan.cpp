// HostPool under ThreadSanitizer: regions of 2 and 256 tasks alternate (the hand-over a late worker could corrupt),
// every task must run exactly once and run() must not return before all of them have finished.
#include <atomic>
#include <cstdio>
#include <vector>
#include "sgo_hostpool.h"
int main() {
  sgo::HostPool pool(8);
  long long bad = 0;
  for (int rep = 0; rep < 4000; ++rep) {
    const int n = (rep & 1) ? 256 : 2;
    std::vector<int> hits(n, 0);          // plain ints: a task run twice concurrently is a data race TSAN reports
    std::atomic<int> finished{0};
    pool.run(n, [&](int t) {
      hits[t] += 1;
      finished.fetch_add(1, std::memory_order_relaxed);
    });
    if (finished.load() != n) ++bad;
    for (int t = 0; t < n; ++t) bad += hits[t] != 1;
  }
  std::printf("hostpool: %lld errors\n", bad);
  return bad ? 1 : 0;
}
