// ThreadSanitizer stress of csrc/host_pool.h (tests/test_host_pool.py builds and runs it): nested loops from four caller threads at once,
// short items, pauses that let the pool fall asleep.
#include "../../csdotrajectoryplanning_amd/csrc/host_pool.h"
#include <cstdio>
#include <cmath>
using namespace csdo;
int main() {
  // nested loops from several caller threads at once, short items, many rounds
  std::vector<std::thread> callers;
  std::atomic<long> sum{0};
  for (int c = 0; c < 4; ++c)
    callers.emplace_back([&, c]() {
      for (int round = 0; round < 300; ++round) {
        std::vector<int> out(40, 0);
        parallel_for(40, 16, [&](int i) {
          std::vector<int> inner(8, 0);
          parallel_for(8, 4, [&](int j) { inner[j] = i * 8 + j + c; });
          int s = 0;
          for (int v : inner) s += v;
          out[i] = s;
        });
        long s = 0;
        for (int v : out) s += v;
        sum += s;
        if (round % 50 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    });
  for (auto& t : callers) t.join();
  long want = 0;
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 40; ++i) for (int j = 0; j < 8; ++j) want += 300L * (i * 8 + j + c);
  printf("sum %ld want %ld %s\n", sum.load(), want, sum.load() == want ? "ok" : "MISMATCH");
  return sum.load() == want ? 0 : 1;
}
