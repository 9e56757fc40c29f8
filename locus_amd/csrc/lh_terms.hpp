// lh_terms.hpp -- the per-point terms of cost_mode 1 with covariances from normals: Mahalanobis matrix (rank-one form) and residual terms.
// ONE implementation for the fused sweep, k_late and k_walk (lh_kernels.hip); __host__ __device__ so that tests/host_emu/terms_check.cpp can
// hold it against the covariance model of the oracle on the CPU (the product only calls it from kernels).
#pragma once
#include "lh_device.hpp"
#include "lh_bfgs.hpp"

namespace lh {

// ---- cost_mode 1 with covariances from normals: the per-point terms, ONE implementation for the fused sweep, k_late and k_walk ----------
// (the translation unit is compiled without FMA contraction because the float NN distances must round like the oracle's; this
// double-precision evaluation exists only in cost_mode 1, whose parity bar is the reference's own FMA / non-FMA floor.  The fused
// multiply-adds are written out: which kernel evaluates a point depends on the iteration, and every one of them must produce the same bits)
struct PoseD { double T[12]; double G[6]; };   // OuterState::poseT / poseG
LH_HD void pose_from_T(const float* T, PoseD& P) { pose_of_transform(T, P.T, P.G); }
// 1 / d to (nearly) the last bit without the correctly-rounded division's ten instructions: v_rcp_f64 is good to 2^-23, two Newton steps
// square that twice.  d is the determinant of a symmetric positive definite matrix of order one: no scaling, no special cases.
LH_HD double rcp_newton(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
#else
  double r = 1.0 / d;   // (host-side check only: the Newton steps below leave a correctly rounded start unchanged to the last bit or two)
#endif
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
// The Mahalanobis matrix M = (C2 + R C1 R^T)^-1 (gicp.hpp:488-493) for C = I - (1 - eps) n n^T / |n|^2 (cov_from_normal), rank-one form:
//   A = (R R^T + I) - k1 u u^T - k2 v v^T,  u = R n1, v = n2, k_i = (1 - eps) / |n_i|^2 (0 for a zero / non-finite normal: C = I).
// Evaluated as l1 l2 A = l1 l2 G - (kap l2) u u^T - (kap l1) v v^T and M = l1 l2 cof / det: ONE reciprocal instead of three divisions.
// G = R R^T + I comes with the pose (uniform over the wave).  M6: 00 01 02 11 12 22.
LH_HD void maha_rank1(const double* R, int rs, const double* G, double kap, const float4& nn, const float4& tn, double (&M6)[6]) {
  const double n1[3] = {(double)nn.x, (double)nn.y, (double)nn.z}, v[3] = {(double)tn.x, (double)tn.y, (double)tn.z};
  const double l1 = __builtin_fma(n1[2], n1[2], __builtin_fma(n1[1], n1[1], n1[0] * n1[0]));
  const double l2 = __builtin_fma(v[2], v[2], __builtin_fma(v[1], v[1], v[0] * v[0]));
  const bool ok1 = l1 > 0.0 && l1 < 1.0e300, ok2 = l2 > 0.0 && l2 < 1.0e300;
  const double l1s = ok1 ? l1 : 1.0, l2s = ok2 ? l2 : 1.0;
  const double w = l1s * l2s, a1 = ok1 ? kap * l2s : 0.0, a2 = ok2 ? kap * l1s : 0.0;
  double u[3];
#pragma unroll
  for (int r = 0; r < 3; r++) u[r] = __builtin_fma(R[r * rs + 2], n1[2], __builtin_fma(R[r * rs + 1], n1[1], R[r * rs + 0] * n1[0]));
  const double ku[3] = {a1 * u[0], a1 * u[1], a1 * u[2]}, kv[3] = {a2 * v[0], a2 * v[1], a2 * v[2]};
  double A[6];
  {
    int q = 0;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = r; cc < 3; cc++) {
        A[q] = __builtin_fma(-kv[r], v[cc], __builtin_fma(-ku[r], u[cc], w * G[q]));
        q++;
      }
  }
  // symmetric cofactors
  const double c00 = __builtin_fma(A[3], A[5], -(A[4] * A[4])), c01 = __builtin_fma(A[2], A[4], -(A[1] * A[5])), c02 = __builtin_fma(A[1], A[4], -(A[2] * A[3]));
  const double c11 = __builtin_fma(A[0], A[5], -(A[2] * A[2])), c12 = __builtin_fma(A[1], A[2], -(A[0] * A[4])), c22 = __builtin_fma(A[0], A[3], -(A[1] * A[1]));
  const double det = __builtin_fma(A[2], c02, __builtin_fma(A[1], c01, A[0] * c00));
  const double id = w * rcp_newton(det);
  M6[0] = c00 * id; M6[1] = c01 * id; M6[2] = c02 * id; M6[3] = c11 * id; M6[4] = c12 * id; M6[5] = c22 * id;
}
// residual a = T p - t about the sweep's transform (gicp.hpp:382-384 in double), M a and a^T M a
LH_HD void resid_terms(const double* T0, const double (&M6)[6], const double (&pt)[3], const float4& t, double (&Ma)[3], double& aMa) {
  const double a0 = (__builtin_fma(T0[2], pt[2], __builtin_fma(T0[1], pt[1], T0[0] * pt[0])) + T0[3]) - (double)t.x;
  const double a1 = (__builtin_fma(T0[6], pt[2], __builtin_fma(T0[5], pt[1], T0[4] * pt[0])) + T0[7]) - (double)t.y;
  const double a2 = (__builtin_fma(T0[10], pt[2], __builtin_fma(T0[9], pt[1], T0[8] * pt[0])) + T0[11]) - (double)t.z;
  Ma[0] = __builtin_fma(M6[2], a2, __builtin_fma(M6[1], a1, M6[0] * a0));
  Ma[1] = __builtin_fma(M6[4], a2, __builtin_fma(M6[3], a1, M6[1] * a0));
  Ma[2] = __builtin_fma(M6[5], a2, __builtin_fma(M6[4], a1, M6[2] * a0));
  aMa = __builtin_fma(a2, Ma[2], __builtin_fma(a1, Ma[1], a0 * Ma[0]));
}

}  // namespace lh
