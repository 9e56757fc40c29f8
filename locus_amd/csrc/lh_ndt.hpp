// lh_ndt.hpp -- per-(point, cell) arithmetic of the NDT registration (registration_method: ndt), shared by the
// derivative kernels (lh_kernels.hip) and the host driver (lh_api.hip).
//
// Reference: multithreaded_gicp/include/multithreaded_ndt/ndt_omp_impl.hpp (pclomp::NormalDistributionsTransform, [Magnusson 2009]):
//   :350-476  computeAngleDerivatives   -> NdtFrame (filled on the host once per evaluation, travels in the kernarg)
//   :480-530  computePointDerivatives (float 4x6 / 24x6)      \
//   :576-652  updateDerivatives (float)                         > ndt_term_float
//   :532-574, :713-748 computePointDerivatives (double), updateHessian -> ndt_term_hessian_double
#pragma once
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
  if (!f.want_h) return;
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

}  // namespace lh
