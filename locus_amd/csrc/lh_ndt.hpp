// lh_ndt.hpp -- per-(point, cell) arithmetic of the NDT registration (registration_method: ndt), shared by the
// derivative kernels (lh_kernels.hip) and the host driver (lh_api.hip).
//
// Reference: multithreaded_gicp/include/multithreaded_ndt/ndt_omp_impl.hpp (pclomp::NormalDistributionsTransform, [Magnusson 2009]):
//   :350-476  computeAngleDerivatives   -> NdtFrame (filled on the host once per evaluation, travels in the kernarg)
//   :480-530  computePointDerivatives (float 4x6 / 24x6)      \
//   :576-652  updateDerivatives (float)                         > ndt_term_float
//   :532-574, :713-748 computePointDerivatives (double), updateHessian -> ndt_term_hessian_double
#pragma once
#include <math.h>

#include "lh_device.hpp"

namespace lh {

constexpr int NDT_NSUM = 43;  // score, gradient[6], hessian[36] (the float path's hessian is not exactly symmetric: all 36 kept)

struct NdtFrame {     // everything an evaluation at pose p needs besides the clouds; plain data (kernarg)
  float T[12];        // final_transformation_ rows 0..2 (row-major 3x4): trans = T * x, float (transformPointCloud)
  float j_ang[8][4];  // eq. 6.19 rows a..h as floats (Matrix<float,8,4> j_ang), 4th column 0
  float h_ang[16][4]; // eq. 6.21 rows a2..f3 (row 15 unused)
  double jd[8][3];    // the same in double (j_ang_a_ .. j_ang_h_): computeHessian
  double hd[15][3];   // h_ang_a2_ .. h_ang_f3_
  double d1, d2;      // gauss_d1_, gauss_d2_
  float r2;           // resolution^2: radius of the voxel-centroid search
  int want_h;
};

LH_HD float ndt_dot4(const float* a, const float* b) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

// one (point, cell) term of computeDerivatives.  x = original point, xt = transformed point - cell mean (double), icov = cell's
// inverse covariance (row-major double[9]).  Adds to acc[0] (score), acc[1..6] (gradient), acc[7..42] (hessian, row-major).
template <bool WANT_H>
LH_HD void ndt_term_float(const NdtFrame& f, const float* x3, const double* xt, const double* icov, double* acc) {
  float x4[4] = {x3[0], x3[1], x3[2], 0.0f};
  float pg[4][6];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) pg[a][b] = 0.0f;
  pg[0][0] = pg[1][1] = pg[2][2] = 1.0f;
  float xj[8];
#pragma unroll
  for (int r = 0; r < 8; r++) xj[r] = ndt_dot4(f.j_ang[r], x4);
  pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
  float xt4[4] = {(float)xt[0], (float)xt[1], (float)xt[2], 0.0f};
  float ci[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) ci[a][b] = (a < 3 && b < 3) ? (float)icov[a * 3 + b] : 0.0f;
  const float d2 = (float)f.d2;
  float xci[4];
#pragma unroll
  for (int b = 0; b < 4; b++) xci[b] = ((xt4[0] * ci[0][b] + xt4[1] * ci[1][b]) + xt4[2] * ci[2][b]) + xt4[3] * ci[3][b];
  float e = expf(-d2 * ndt_dot4(xt4, xci) * 0.5f);
  float score_inc = (float)(-f.d1 * (double)e);
  e = d2 * e;
  if (e > 1 || e < 0 || e != e) return;
  e = (float)((double)e * f.d1);
  float cg[4][6];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int j = 0; j < 6; j++) cg[a][j] = ((ci[a][0] * pg[0][j] + ci[a][1] * pg[1][j]) + ci[a][2] * pg[2][j]) + ci[a][3] * pg[3][j];
  float xcg[6];
#pragma unroll
  for (int j = 0; j < 6; j++) xcg[j] = ((xt4[0] * cg[0][j] + xt4[1] * cg[1][j]) + xt4[2] * cg[2][j]) + xt4[3] * cg[3][j];
  acc[0] += (double)score_inc;
#pragma unroll
  for (int j = 0; j < 6; j++) acc[1 + j] += (double)(e * xcg[j]);
  if (!WANT_H) return;
  float xh[15];
#pragma unroll
  for (int r = 0; r < 15; r++) xh[r] = ndt_dot4(f.h_ang[r], x4);
  // point_hessian_ blocks (4 rows each; 4th row 0): block(i, j) for i, j in 3..5 = a b c / b d e / c e f
  const float blk[6][3] = {{0.f, xh[0], xh[1]}, {0.f, xh[2], xh[3]}, {0.f, xh[4], xh[5]}, {xh[6], xh[7], xh[8]}, {xh[9], xh[10], xh[11]}, {xh[12], xh[13], xh[14]}};
  const int which[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  float gcg[6][6];
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) gcg[a][b] = ((pg[0][a] * cg[0][b] + pg[1][a] * cg[1][b]) + pg[2][a] * cg[2][b]) + pg[3][a] * cg[3][b];
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float xch = 0.0f;  // x_trans4_x_c_inv4 * point_hessian_.block<4,6>(i*4, 0), column j: ((0 + 0) + 0) + 0 outside the 3..5 blocks
      if (i >= 3 && j >= 3) {
        const float* v = blk[which[i - 3][j - 3]];
        xch = ((xci[0] * v[0] + xci[1] * v[1]) + xci[2] * v[2]) + xci[3] * 0.0f;
      }
      acc[7 + i * 6 + j] += (double)(e * (-d2 * xcg[i] * xcg[j] + xch + gcg[j][i]));
    }
  }
}

// one (point, cell) term of computeHessian / updateHessian (double path); adds to H[36] (row-major)
LH_HD void ndt_term_hessian_double(const NdtFrame& f, const float* x3, const double* d /*xt - mean*/, const double* ci, double* H) {
  const double x[3] = {(double)x3[0], (double)x3[1], (double)x3[2]};
  double cd[3];
#pragma unroll
  for (int a = 0; a < 3; a++) cd[a] = ci[a * 3] * d[0] + ci[a * 3 + 1] * d[1] + ci[a * 3 + 2] * d[2];
  double e = f.d2 * exp(-f.d2 * (d[0] * cd[0] + d[1] * cd[1] + d[2] * cd[2]) / 2);
  if (e > 1 || e < 0 || e != e) return;
  e *= f.d1;
  double pg[3][6];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) pg[a][b] = 0.0;
  pg[0][0] = pg[1][1] = pg[2][2] = 1.0;
#define LH_DOT3(v) (x[0] * (v)[0] + x[1] * (v)[1] + x[2] * (v)[2])
  pg[1][3] = LH_DOT3(f.jd[0]); pg[2][3] = LH_DOT3(f.jd[1]); pg[0][4] = LH_DOT3(f.jd[2]); pg[1][4] = LH_DOT3(f.jd[3]); pg[2][4] = LH_DOT3(f.jd[4]);
  pg[0][5] = LH_DOT3(f.jd[5]); pg[1][5] = LH_DOT3(f.jd[6]); pg[2][5] = LH_DOT3(f.jd[7]);
  const double blk[6][3] = {{0.0, LH_DOT3(f.hd[0]), LH_DOT3(f.hd[1])}, {0.0, LH_DOT3(f.hd[2]), LH_DOT3(f.hd[3])}, {0.0, LH_DOT3(f.hd[4]), LH_DOT3(f.hd[5])},
                            {LH_DOT3(f.hd[6]), LH_DOT3(f.hd[7]), LH_DOT3(f.hd[8])}, {LH_DOT3(f.hd[9]), LH_DOT3(f.hd[10]), LH_DOT3(f.hd[11])},
                            {LH_DOT3(f.hd[12]), LH_DOT3(f.hd[13]), LH_DOT3(f.hd[14])}};
#undef LH_DOT3
  const int which[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  double cg[3][6];
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int a = 0; a < 3; a++) cg[a][j] = ci[a * 3] * pg[0][j] + ci[a * 3 + 1] * pg[1][j] + ci[a * 3 + 2] * pg[2][j];
  double xd[6];
#pragma unroll
  for (int j = 0; j < 6; j++) xd[j] = d[0] * cg[0][j] + d[1] * cg[1][j] + d[2] * cg[2][j];
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double t2 = 0.0;
      if (i >= 3 && j >= 3) {
        const double* v = blk[which[i - 3][j - 3]];
        double chv[3];
#pragma unroll
        for (int a = 0; a < 3; a++) chv[a] = ci[a * 3] * v[0] + ci[a * 3 + 1] * v[1] + ci[a * 3 + 2] * v[2];
        t2 = d[0] * chv[0] + d[1] * chv[1] + d[2] * chv[2];
      }
      double t3 = pg[0][j] * cg[0][i] + pg[1][j] * cg[1][i] + pg[2][j] * cg[2][i];
      H[i * 6 + j] += e * (-f.d2 * xd[i] * xd[j] + t2 + t3);
    }
  }
}

// ---- target cells: VoxelGridCovariance::applyFilter's per-leaf algebra (voxel_grid_covariance_omp_impl.hpp:215-282), one voxel per
// thread on the device.  3x3 symmetric eigen-decomposition by cyclic Jacobi sweeps, ascending (stands in for SelfAdjointEigenSolver).
LH_HD void ndt_eig_sym3(const double* A, double* ev, double* V /*row-major, eigenvectors in columns*/) {
  double a[9];
  for (int i = 0; i < 9; i++) a[i] = A[i];
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = a[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { double akp = a[k * 3 + p], akq = a[k * 3 + q]; a[k * 3 + p] = c * akp - s * akq; a[k * 3 + q] = s * akp + c * akq; }
        for (int k = 0; k < 3; k++) { double apk = a[p * 3 + k], aqk = a[q * 3 + k]; a[p * 3 + k] = c * apk - s * aqk; a[q * 3 + k] = s * apk + c * aqk; }
        for (int k = 0; k < 3; k++) { double vkp = V[k * 3 + p], vkq = V[k * 3 + q]; V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq; }
      }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0], a[4], a[8]};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (d[order[j]] > d[order[j + 1]]) { int t = order[j]; order[j] = order[j + 1]; order[j + 1] = t; }
  double Vs[9];
  for (int k = 0; k < 3; k++) { ev[k] = d[order[k]]; for (int r = 0; r < 3; r++) Vs[r * 3 + k] = V[r * 3 + order[k]]; }
  for (int i = 0; i < 9; i++) V[i] = Vs[i];
}

// one voxel: raw sums -> (mean, inverse covariance); returns false if the voxel holds too few points (it does not become a cell)
LH_HD bool ndt_finish_cell(const double* sum3, const double* cov6_raw, int np, int min_points, double eig_mult, double* mean3, double* icov9) {
  if (np < min_points) return false;
  double mean[3], cov[9];
  for (int a = 0; a < 3; a++) mean3[a] = mean[a] = sum3[a] / np;
  const double raw[9] = {cov6_raw[0], cov6_raw[1], cov6_raw[2], cov6_raw[1], cov6_raw[3], cov6_raw[4], cov6_raw[2], cov6_raw[4], cov6_raw[5]};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double v = (raw[a * 3 + b] - 2.0 * (sum3[a] * mean[b])) / np + mean[a] * mean[b];  // :236
      cov[a * 3 + b] = v * ((np - 1.0) / np);                                             // :237
    }
  double sym[9], ev[3], V[9];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) sym[a * 3 + b] = cov[(a > b ? a : b) * 3 + (a > b ? b : a)];  // the solver reads the lower triangle
  ndt_eig_sym3(sym, ev, V);
  for (int k = 0; k < 9; k++) icov9[k] = 0.0;
  if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) return true;  // rejected by the eigenvalue check (:250-254): stays searchable, icov = 0
  double minev = eig_mult * ev[2];
  if (ev[0] < minev) {                                      // :258-268
    ev[0] = minev;
    if (ev[1] < minev) ev[1] = minev;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double v = 0;
        for (int k = 0; k < 3; k++) v += V[a * 3 + k] * ev[k] * V[b * 3 + k];
        cov[a * 3 + b] = v;
      }
  }
  double c00 = cov[4] * cov[8] - cov[5] * cov[7], c01 = cov[5] * cov[6] - cov[3] * cov[8], c02 = cov[3] * cov[7] - cov[4] * cov[6];
  double id = 1.0 / (cov[0] * c00 + cov[1] * c01 + cov[2] * c02);
  icov9[0] = c00 * id; icov9[1] = (cov[2] * cov[7] - cov[1] * cov[8]) * id; icov9[2] = (cov[1] * cov[5] - cov[2] * cov[4]) * id;
  icov9[3] = c01 * id; icov9[4] = (cov[0] * cov[8] - cov[2] * cov[6]) * id; icov9[5] = (cov[2] * cov[3] - cov[0] * cov[5]) * id;
  icov9[6] = c02 * id; icov9[7] = (cov[1] * cov[6] - cov[0] * cov[7]) * id; icov9[8] = (cov[0] * cov[4] - cov[1] * cov[3]) * id;
  return true;
}


}  // namespace lh
