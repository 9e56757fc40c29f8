// CPU check of locus_amd/host/AdaptiveVoxelization.hpp against a hand-traced run of Locus::ApplyAdaptiveInputVoxelization
// (Locus.cc:780-810): the leaf grows when too many points arrive, holds inside the 0.01 dead band, re-sends every 20th scan,
// and is clamped to [0.01, 5].
#include <cmath>
#include <cstdio>

#include "../../locus_amd/host/AdaptiveVoxelization.hpp"

using locus_hip::AdaptiveVoxelization;

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
  AdaptiveVoxelization a(0.25, 3000);
  double leaf = 0;
  EXPECT(a.Update(6000, &leaf));                      // first call: counter % 20 == 0 forces a send; 0.25 * 2 = 0.5
  EXPECT(std::fabs(leaf - 0.5) < 1e-15 && std::fabs(a.leaf_size() - 0.5) < 1e-15);
  EXPECT(!a.Update(3030, &leaf));                     // 0.5 * 1.01 = 0.505: inside the dead band, value kept
  EXPECT(std::fabs(leaf - 0.505) < 1e-12 && std::fabs(a.leaf_size() - 0.5) < 1e-15);
  EXPECT(a.Update(3300, &leaf));                      // 0.55: moved by 0.05
  EXPECT(std::fabs(a.leaf_size() - 0.55) < 1e-12);
  int sends = 0;
  for (int k = 0; k < 40; k++) sends += a.Update(3000, &leaf) ? 1 : 0;   // steady state: only the every-20th-scan re-send
  EXPECT(sends == 2 && std::fabs(a.leaf_size() - 0.55) < 1e-12);
  EXPECT(a.Update(3000000, &leaf) && leaf == 5.0);    // clamp high
  EXPECT(a.Update(1, &leaf) && leaf == 0.01);         // clamp low
  printf(fails ? "CONTROLLER_CHECK_FAILED\n" : "CONTROLLER_CHECK_OK\n");
  return fails ? 1 : 0;
}
