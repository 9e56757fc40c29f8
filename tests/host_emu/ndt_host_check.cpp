// Host side of the NDT registration (locus_amd/csrc/lh_ndt_host.hpp) against the oracle's restatement of the same pieces of
// pclomp::NormalDistributionsTransform: pose <-> matrix (ndt_omp_impl.hpp:160-170, 906-919), the 6x6 JacobiSVD solve of the
// Newton step (:139-143) and the Gauss constants (:60-70 region).  Links oracle/liblocus_oracle.so -- test code only.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../locus_amd/csrc/lh_ndt_host.hpp"
extern "C" {
#include "../../oracle/locus_oracle.h"
}

int main() {
  std::mt19937 rng(11);
  std::normal_distribution<double> g(0.0, 1.0);
  int bad = 0;
  // pose -> matrix -> pose, bit for bit
  for (int it = 0; it < 2000; it++) {
    double p[6] = {10 * g(rng), 10 * g(rng), 3 * g(rng), 0.5 * g(rng), 0.5 * g(rng), 3.0 * g(rng)};
    if (it % 7 == 0) p[3] = p[4] = 0.0;
    if (it % 11 == 0) std::memset(p, 0, sizeof(p));
    float Ta[16], Tb[16];
    lh::ndt_pose_to_matrix(p, Ta);
    lo_ndt_pose_to_matrix(p, Tb);
    if (std::memcmp(Ta, Tb, sizeof(Ta)) != 0) { bad++; if (bad < 5) std::printf("pose_to_matrix differs at %d\n", it); }
    double qa[6], qb[6];
    lh::ndt_matrix_to_pose(Ta, qa);
    lo_ndt_matrix_to_pose(Tb, qb);
    if (std::memcmp(qa, qb, sizeof(qa)) != 0) { bad++; if (bad < 5) std::printf("matrix_to_pose differs at %d\n", it); }
    for (int k = 0; k < 3; k++)
      if (std::fabs(qa[k] - p[k]) > 1e-5 * (1.0 + std::fabs(p[k]))) { bad++; if (bad < 5) std::printf("translation round trip off at %d\n", it); }
  }
  // Newton step: x = svd(H).solve(-g); well conditioned, badly scaled and rank-deficient Hessians
  for (int it = 0; it < 500; it++) {
    double J[12][6], H[36], b[6], xa[6], xb[6];
    int rows = (it % 5 == 4) ? 4 : 12;  // rows < 6 => rank-deficient
    for (int r = 0; r < 12; r++)
      for (int c = 0; c < 6; c++) J[r][c] = r < rows ? g(rng) * ((it % 3 == 1 && c >= 3) ? 1e4 : 1.0) : 0.0;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double s = 0.0;
        for (int r = 0; r < 12; r++) s += J[r][i] * J[r][j];
        H[i * 6 + j] = s;
      }
    for (int i = 0; i < 6; i++) b[i] = g(rng);
    lh::ndt_svd_solve6(H, b, xa);
    lo_svd_solve6(H, b, xb);
    double num = 0, den = 1e-300;
    for (int i = 0; i < 6; i++) { num = std::fmax(num, std::fabs(xa[i] - xb[i])); den = std::fmax(den, std::fabs(xb[i])); }
    if (!(num <= 1e-12 * den)) { bad++; if (bad < 5) std::printf("svd solve differs at %d: %.3e of %.3e\n", it, num, den); }
    if (rows == 12) {  // full rank: it must actually solve the system
      double res = 0, nb = 0;
      for (int i = 0; i < 6; i++) {
        double s = -b[i];
        for (int j = 0; j < 6; j++) s += H[i * 6 + j] * xa[j];
        res = std::fmax(res, std::fabs(s));
        nb = std::fmax(nb, std::fabs(b[i]));
      }
      if (!(res <= 1e-6 * (1.0 + nb))) { bad++; if (bad < 5) std::printf("svd solve residual %.3e at %d\n", res, it); }
    }
  }
  std::printf(bad ? "NDT_HOST_CHECK_FAILED (%d)\n" : "NDT_HOST_CHECK_OK\n", bad);
  return bad ? 1 : 0;
}
