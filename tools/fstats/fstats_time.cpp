// Host-side timing of the numEntriesScannedInFilter replay for an AND of two scan leaves over 64 Mi docs: the state machine per chunk on 1..16 threads.
//   g++ -O2 -std=c++17 -pthread -Iinclude -o /tmp/fstats_time tools/fstats/fstats_time.cpp && /tmp/fstats_time
#include "../../pinot_amd/csrc/pg_filter_stats.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
int main() {
  const int32_t n = 64 << 20;
  const size_t words = ((size_t)n + 63) / 64;
  std::mt19937_64 rng(1);
  std::vector<std::vector<uint64_t>> L(2, std::vector<uint64_t>(words));
  for (int l = 0; l < 2; ++l) for (auto& w : L[l]) w = rng() & rng() & (l ? rng() : ~0ull);
  std::vector<const uint64_t*> ptrs{L[0].data(), L[1].data()};
  for (int th : {1, 2, 4, 8, 16}) {
    std::atomic<int> next{0};
    std::vector<int> done(64, 0);
    std::vector<double> busy(64, 0.0); std::atomic<long long> sink{0};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int id) {
      for (int c = next.fetch_add(1); c < 64; c = next.fetch_add(1)) {
        auto a = std::chrono::steady_clock::now();
        for (int s = 0; s < 2; ++s) sink += pg::fstats::simulate_and_chunk(ptrs, c << 20, (c + 1) << 20, s).entries;
        busy[id] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        done[id]++;
      }
    };
    std::vector<std::thread> ts;
    for (int t = 1; t < th; ++t) ts.emplace_back(worker, t);
    worker(0);
    for (auto& t : ts) t.join();
    printf("%2d threads: %.1f ms; chunks per thread:", th, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (int t = 0; t < th; ++t) printf(" %d(%.0fms)", done[t], busy[t]);
    printf("\n");
  }
}
