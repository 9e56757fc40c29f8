// lh_ndt_host.hpp -- host side of the NDT registration (registration_method: ndt): the Newton / More-Thuente
// control flow of pclomp::NormalDistributionsTransform::computeTransformation (ndt_omp_impl.hpp:101-212, 855-1049), the cell
// finishing of VoxelGridCovariance::applyFilter (voxel_grid_covariance_omp_impl.hpp:215-282) and the small dense algebra
// around them.  Every evaluation of (score, gradient, hessian) is one device pass (k_ndt_derivs); nothing here touches points.
#pragma once
#include <cmath>
#include <cstring>
#include <functional>

#include "lh_ndt.hpp"

namespace lh {

// ---- pose <-> matrix -----------------------------------------------------------------------------------------------------
// (Translation * AngleAxis(x) * AngleAxis(y) * AngleAxis(z)).matrix() in float: Eigen multiplies the AngleAxis factors as
// quaternions (ndt_omp_impl.hpp:160-170, 906-919)
inline void ndt_pose_to_matrix(const double* p6, float* T16 /*col-major*/) {
  struct Q { float w, x, y, z; };
  auto mul = [](Q a, Q b) {
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
  };
  float ax = (float)p6[3], ay = (float)p6[4], az = (float)p6[5];
  Q q = mul(mul(Q{cosf(0.5f * ax), sinf(0.5f * ax), 0.f, 0.f}, Q{cosf(0.5f * ay), 0.f, sinf(0.5f * ay), 0.f}), Q{cosf(0.5f * az), 0.f, 0.f, sinf(0.5f * az)});
  float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
  float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  float R[9] = {1.f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.f - (txx + tyy)};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T16[c * 4 + r] = R[r * 3 + c];
  T16[12] = (float)p6[0]; T16[13] = (float)p6[1]; T16[14] = (float)p6[2];
  T16[3] = T16[7] = T16[11] = 0.f; T16[15] = 1.f;
}
// translation + rotation().eulerAngles(0, 1, 2) of the float matrix (ndt_omp_impl.hpp:127-135; Eigen/src/Geometry/EulerAngles.h)
inline void ndt_matrix_to_pose(const float* T, double* p6) {
  auto M = [&](int r, int c) { return T[c * 4 + r]; };
  float res[3];
  res[0] = atan2f(M(1, 2), M(2, 2));
  float c2 = sqrtf(M(0, 0) * M(0, 0) + M(0, 1) * M(0, 1));
  if (res[0] > 0.f) {
    res[0] -= (float)M_PI;
    res[1] = atan2f(-M(0, 2), -c2);
  } else
    res[1] = atan2f(-M(0, 2), c2);
  float s1 = sinf(res[0]), c1 = cosf(res[0]);
  res[2] = atan2f(s1 * M(2, 0) - c1 * M(1, 0), c1 * M(1, 1) - s1 * M(2, 1));
  p6[0] = T[12]; p6[1] = T[13]; p6[2] = T[14];
  p6[3] = -res[0]; p6[4] = -res[1]; p6[5] = -res[2];
}

// ---- the per-evaluation frame: transform rows + angle derivative tables (computeAngleDerivatives, ndt_omp_impl.hpp:350-476) ----
inline void ndt_gauss(double resolution, double outlier_ratio, double* d1, double* d2, double* d3) {
  double c1 = 10.0 * (1 - outlier_ratio), c2 = outlier_ratio / pow(resolution, 3);  // eq. 6.8 [Magnusson 2009], :101-111
  *d3 = -log(c2);
  *d1 = -log(c1 + c2) - *d3;
  *d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - *d3) / *d1);
}
inline void ndt_fill_frame(NdtFrame& f, const double* p, const float* T16, float resolution, double outlier_ratio, int want_h) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) f.T[r * 4 + c] = T16[c * 4 + r];
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  const double J[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                          {-sy * cz, sy * sz, cy}, {sx * cy * cz, -sx * cy * sz, sx * sy}, {-cx * cy * cz, cx * cy * sz, -cx * sy},
                          {-cy * sz, -cy * cz, 0}, {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0}, {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  const double H[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                           {cx * cy * cz, -cx * cy * sz, cx * sy}, {sx * cy * cz, -sx * cy * sz, sx * sy},
                           {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0}, {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                           {-cy * cz, cy * sz, sy}, {-sx * sy * cz, sx * sy * sz, sx * cy}, {cx * sy * cz, -cx * sy * sz, -cx * cy},
                           {sy * sz, sy * cz, 0}, {-sx * cy * sz, -sx * cy * cz, 0}, {cx * cy * sz, cx * cy * cz, 0},
                           {-cy * cz, cy * sz, 0}, {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  memset(f.j_ang, 0, sizeof(f.j_ang));
  memset(f.h_ang, 0, sizeof(f.h_ang));
  for (int r = 0; r < 8; r++)
    for (int k = 0; k < 3; k++) { f.jd[r][k] = J[r][k]; f.j_ang[r][k] = (float)J[r][k]; }
  for (int r = 0; r < 15; r++)
    for (int k = 0; k < 3; k++) { f.hd[r][k] = H[r][k]; f.h_ang[r][k] = (float)H[r][k]; }
  double d3;
  ndt_gauss(resolution, outlier_ratio, &f.d1, &f.d2, &d3);
  f.r2 = resolution * resolution;
  f.want_h = want_h;
}

// ---- x = pinv(A) b for 6x6 through a one-sided Jacobi SVD: JacobiSVD<Matrix6d>(hessian, FullU|FullV).solve(-gradient) ------
inline void ndt_svd_solve6(const double* A36, const double* b6, double* x6) {
  double U[36], V[36];
  memcpy(U, A36, sizeof(U));
  for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double a = 0, b = 0, g = 0;
        for (int k = 0; k < 6; k++) { a += U[k * 6 + p] * U[k * 6 + p]; b += U[k * 6 + q] * U[k * 6 + q]; g += U[k * 6 + p] * U[k * 6 + q]; }
        if (fabs(g) <= 1e-300 || fabs(g) <= 1e-17 * sqrt(a * b)) continue;
        off += fabs(g);
        double zeta = (b - a) / (2.0 * g);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int k = 0; k < 6; k++) {
          double up = U[k * 6 + p], uq = U[k * 6 + q];
          U[k * 6 + p] = cs * up - sn * uq; U[k * 6 + q] = sn * up + cs * uq;
          double vp = V[k * 6 + p], vq = V[k * 6 + q];
          V[k * 6 + p] = cs * vp - sn * vq; V[k * 6 + q] = sn * vp + cs * vq;
        }
      }
    if (off == 0) break;
  }
  double sig[6], smax = 0;
  for (int j = 0; j < 6; j++) {
    double s = 0;
    for (int k = 0; k < 6; k++) s += U[k * 6 + j] * U[k * 6 + j];
    sig[j] = sqrt(s);
    if (sig[j] > smax) smax = sig[j];
  }
  const double thr = 6.0 * 2.220446049250313e-16 * smax;
  for (int i = 0; i < 6; i++) x6[i] = 0;
  for (int j = 0; j < 6; j++) {
    if (!(sig[j] > thr)) continue;
    double ub = 0;
    for (int k = 0; k < 6; k++) ub += U[k * 6 + j] * b6[k];
    ub /= sig[j] * sig[j];
    for (int i = 0; i < 6; i++) x6[i] += V[i * 6 + j] * ub;
  }
}

// ---- More-Thuente helpers (ndt_omp_impl.hpp:751-853) ---------------------------------------------------------------------------
inline bool ndt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}
inline double ndt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(g_l)) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_n = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? std::fmin(a_t + 0.66 * (a_u - a_t), a_n) : std::fmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = std::sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

// ---- computeTransformation -------------------------------------------------------------------------------------------------------
struct NdtOutcome { float T[16]; int converged, iterations, evaluations; double score; };
// eval(p, T16, want_h, hessian_only, score, grad6, hess36): one device pass at pose p / transform T16
typedef std::function<bool(const double*, const float*, int, int, double*, double*, double*)> NdtEval;

inline bool ndt_compute_transformation(const NdtEval& eval, const float* guess16, bool guess_is_identity, double step_size, double tf_eps,
                                       int max_iterations, NdtOutcome* out) {
  static const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float final_T[16];
  memcpy(final_T, guess_is_identity ? I16 : guess16, sizeof(I16));
  double p[6], dp[6], grad[6], H[36], score = 0;
  ndt_matrix_to_pose(final_T, p);
  int iters = 0, evals = 1;
  bool converged = false;
  if (!eval(p, final_T, 1, 0, &score, grad, H)) return false;   // ndt_omp_impl.hpp:146
  while (!converged) {
    double ng[6];
    for (int k = 0; k < 6; k++) ng[k] = -grad[k];
    ndt_svd_solve6(H, ng, dp);                                   // :153-157
    double norm = 0;
    for (int k = 0; k < 6; k++) norm += dp[k] * dp[k];
    norm = std::sqrt(norm);
    if (norm == 0 || norm != norm) { converged = norm == norm; break; }   // :162-166
    for (int k = 0; k < 6; k++) dp[k] /= norm;
    // computeStepLengthMT(p, dp, norm, step_size, tf_eps / 2, ...)  (:855-1049)
    const double step_max = step_size, step_min = tf_eps / 2;
    double phi_0 = -score, d_phi_0 = 0;
    for (int k = 0; k < 6; k++) d_phi_0 -= grad[k] * dp[k];
    double a_t = 0;
    bool search = true;
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) search = false;
      else { d_phi_0 = -d_phi_0; for (int k = 0; k < 6; k++) dp[k] = -dp[k]; }
    }
    if (search) {
      const double mu = 1.e-4, nu = 0.9;
      int step_iterations = 0;
      double a_l = 0, a_u = 0;
      double f_l = phi_0 - phi_0 - mu * d_phi_0 * a_l, g_l = d_phi_0 - mu * d_phi_0;
      double f_u = phi_0 - phi_0 - mu * d_phi_0 * a_u, g_u = d_phi_0 - mu * d_phi_0;
      bool interval_converged = (step_max - step_min) < 0, open_interval = true;
      a_t = std::fmax(std::fmin(norm, step_max), step_min);
      double x_t[6];
      for (int k = 0; k < 6; k++) x_t[k] = p[k] + dp[k] * a_t;
      ndt_pose_to_matrix(x_t, final_T);
      if (!eval(x_t, final_T, 1, 0, &score, grad, H)) return false;
      evals++;
      double phi_t = -score, d_phi_t = 0;
      for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * dp[k];
      double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
      while (!interval_converged && step_iterations < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        a_t = open_interval ? ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                            : ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = std::fmax(std::fmin(a_t, step_max), step_min);
        for (int k = 0; k < 6; k++) x_t[k] = p[k] + dp[k] * a_t;
        ndt_pose_to_matrix(x_t, final_T);
        if (!eval(x_t, final_T, 0, 0, &score, grad, H)) return false;   // compute_hessian = false: hessian comes back zero
        evals++;
        phi_t = -score;
        d_phi_t = 0;
        for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * dp[k];
        psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
        d_psi_t = d_phi_t - mu * d_phi_0;
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
          open_interval = false;
          f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
          f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
        }
        interval_converged = open_interval ? ndt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                           : ndt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
      }
      if (step_iterations) {   // computeHessian at the accepted step (:1044-1045), double path
        double s_unused, g_unused[6];
        if (!eval(x_t, final_T, 1, 1, &s_unused, g_unused, H)) return false;
      }
    }
    norm = a_t;
    for (int k = 0; k < 6; k++) { dp[k] *= norm; p[k] += dp[k]; }
    if (iters > max_iterations || (iters && std::fabs(norm) < tf_eps)) converged = true;   // :188-193
    iters++;
  }
  memcpy(out->T, final_T, sizeof(final_T));
  out->converged = converged ? 1 : 0;
  out->iterations = iters;
  out->evaluations = evals;
  out->score = score;
  return true;
}

}  // namespace lh
