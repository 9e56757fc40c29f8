// Host-side stress of HostPool (lh_runtime.hpp): the pool that resumes the alignment coroutines of a scheduler group.  Every index of
// every parallel_for must run EXACTLY once, whatever the sizes of consecutive calls and however late a worker leaves the previous one
// (round 3: an index counter that was reset per call let a straggler run an index of the next call twice -- one coroutine resumed by
// two threads).  Compiled as a host program; no device code runs.
#include <atomic>
#include <cstdio>
#include <random>
#include <vector>

#include "../../locus_amd/csrc/lh_runtime.hpp"

int main() {
  HostPool pool(7);
  std::mt19937 rng(12345);
  const int calls = 100000;
  std::vector<std::atomic<int>> hits(64);
  long bad = 0;
  for (int c = 0; c < calls; c++) {
    const int n = 2 + (int)(rng() % 40) + ((c % 97) == 0 ? 20 : 0);   // sizes go up and down between consecutive calls
    for (int i = 0; i < n; i++) hits[i].store(0, std::memory_order_relaxed);
    pool.parallel_for(n, [&](int i) {
      hits[i].fetch_add(1, std::memory_order_relaxed);
      if ((i & 7) == 3) for (volatile int spin = 0; spin < 50; spin = spin + 1) {}   // uneven item lengths
    });
    for (int i = 0; i < n; i++) bad += hits[i].load(std::memory_order_relaxed) != 1;
    for (int i = n; i < 64; i++) bad += hits[i].load(std::memory_order_relaxed) > 1;
  }
  printf("%d parallel_for calls, %ld indices not run exactly once\n", calls, bad);
  printf(bad ? "HOSTPOOL_CHECK_FAILED\n" : "HOSTPOOL_CHECK_OK\n");
  return bad ? 1 : 0;
}
