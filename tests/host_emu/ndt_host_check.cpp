// Host side of the NDT registration (locus_amd/csrc/lh_ndt_host.hpp) against the oracle's restatement of the same pieces of
// pclomp::NormalDistributionsTransform: pose <-> matrix (ndt_omp_impl.hpp:160-170, 906-919), the 6x6 JacobiSVD solve of the
// Newton step (:139-143) and the Gauss constants (:60-70 region).  Links oracle/liblocus_oracle.so -- test code only.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

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
  // the Newton / More-Thuente control flow: the product's ndt_compute_transformation driven by the ORACLE's derivative passes
  // (standing in for the device kernels) must retrace lo_ndt_align step for step -- same transform, iteration and evaluation counts
  for (int scene = 0; scene < 3; scene++) {
    const int n = 6000;
    std::vector<float> tgt(4 * n), src(4 * n);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (int i = 0; i < n; i++) {  // three walls and a floor of a 12 x 8 x 3 m room, 1 cm noise
      float a = u(rng), b = u(rng), x, y, z;
      switch (i % 4) {
        case 0: x = 6.f * a; y = 4.f * b; z = 0.f; break;
        case 1: x = 6.f; y = 4.f * a; z = 1.5f + 1.5f * b; break;
        case 2: x = 6.f * a; y = 4.f; z = 1.5f + 1.5f * b; break;
        default: x = -6.f; y = 4.f * a; z = 1.5f + 1.5f * b; break;
      }
      tgt[4 * i] = x + 0.01f * (float)g(rng); tgt[4 * i + 1] = y + 0.01f * (float)g(rng); tgt[4 * i + 2] = z + 0.01f * (float)g(rng); tgt[4 * i + 3] = 1.f;
    }
    const double truth[6] = {0.15 + 0.1 * scene, -0.1, 0.03, 0.004, -0.006, 0.02 + 0.01 * scene};
    float Tt[16];
    lo_ndt_pose_to_matrix(truth, Tt);
    lo_transform(tgt.data(), nullptr, n, Tt, src.data(), nullptr);  // source = moved target (then NDT must undo the motion)
    lo_ndt_params P;
    lo_ndt_default_params(&P);
    P.resolution = 1.0f;
    P.transformation_epsilon = 1e-3;
    P.max_iterations = 20;
    float G[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (scene == 2) { const double gp[6] = {-0.1, 0.05, 0.0, 0.0, 0.0, -0.01}; lo_ndt_pose_to_matrix(gp, G); }
    lo_ndt_result ro;
    lo_ndt_align(src.data(), n, tgt.data(), n, &P, G, &ro);
    lo_ndt_grid* grid = lo_ndt_grid_build(tgt.data(), n, &P);
    std::vector<float> trans(4 * n);
    lh::NdtEval eval = [&](const double* p, const float* T16, int want_h, int hessian_only, double* score, double* grad, double* hess) {
      lo_transform(src.data(), nullptr, n, T16, trans.data(), nullptr);
      if (hessian_only) lo_ndt_hessian(grid, &P, src.data(), trans.data(), n, p, hess);
      else *score = lo_ndt_derivatives(grid, &P, src.data(), trans.data(), n, p, grad, hess, want_h);
      return true;
    };
    lh::NdtOutcome po;
    bool ident = scene != 2;
    if (!lh::ndt_compute_transformation(eval, G, ident, P.step_size, P.transformation_epsilon, P.max_iterations, &po)) { bad++; std::printf("compute_transformation failed\n"); }
    if (std::memcmp(po.T, ro.T, sizeof(po.T)) != 0 || po.iterations != ro.iterations || po.evaluations != ro.evaluations || po.converged != ro.converged) {
      bad++;
      std::printf("scene %d: control flow differs: iterations %d/%d evaluations %d/%d converged %d/%d\n", scene, po.iterations, ro.iterations,
                  po.evaluations, ro.evaluations, po.converged, ro.converged);
    }
    if (ro.iterations < 1) { bad++; std::printf("scene %d: the oracle did not iterate\n", scene); }
    lo_ndt_grid_free(grid);
  }
  std::printf(bad ? "NDT_HOST_CHECK_FAILED (%d)\n" : "NDT_HOST_CHECK_OK\n", bad);
  return bad ? 1 : 0;
}
