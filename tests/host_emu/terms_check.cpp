// Host-side check of locus_amd/csrc/lh_terms.hpp -- the per-point terms of cost_mode 1 (rank-one Mahalanobis matrix with ONE reciprocal,
// residual terms, the pose's doubles) -- against the covariance model of the oracle (oracle/locus_oracle.c lo_cov_from_normals:
// gicp.hpp:81-82's covariances from normals) pushed through the reference's own formula M = (C2 + R C1 R^T)^-1 (gicp.hpp:488-493) in plain
// double arithmetic.  Covers unit, unnormalised, zero and non-finite normals, rotations that are not exactly orthonormal (float T).
// Compiled as a HOST program by hipcc (the header is __host__ __device__); prints TERMS_CHECK_OK.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../locus_amd/csrc/lh_terms.hpp"

extern "C" void lo_cov_from_normals(const float* nrm4, int n, double eps, double* cov9);

using namespace lh;

static void inv3(const double* A, double* Ai) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  Ai[0] = c00 / det; Ai[1] = (A[2] * A[7] - A[1] * A[8]) / det; Ai[2] = (A[1] * A[5] - A[2] * A[4]) / det;
  Ai[3] = c01 / det; Ai[4] = (A[0] * A[8] - A[2] * A[6]) / det; Ai[5] = (A[2] * A[3] - A[0] * A[5]) / det;
  Ai[6] = c02 / det; Ai[7] = (A[1] * A[6] - A[0] * A[7]) / det; Ai[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}

int main() {
  std::mt19937 rng(20260927);
  std::uniform_real_distribution<float> U(-1.0f, 1.0f);
  const double eps = 1e-3;
  double worst_M = 0.0, worst_a = 0.0, worst_G = 0.0;
  int n_special = 0;
  for (int it = 0; it < 20000; it++) {
    // a float transform: small rotation (Euler angles up to 0.3 rad) in float arithmetic -- R R^T != I in its last bits -- and a translation
    float ax = 0.3f * U(rng), ay = 0.3f * U(rng), az = 0.3f * U(rng);
    float cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
    float T[12] = {cy * cz, sx * sy * cz - cx * sz, cx * sy * cz + sx * sz, 3.0f * U(rng),
                   cy * sz, sx * sy * sz + cx * cz, cx * sy * sz - sx * cz, 3.0f * U(rng),
                   -sy,     sx * cy,                cx * cy,                3.0f * U(rng)};
    float4 nn = make_float4(U(rng), U(rng), U(rng), 0.f), tn = make_float4(U(rng), U(rng), U(rng), 0.f);
    const int kind = it % 16;
    if (kind == 0) { float l = std::sqrt(nn.x * nn.x + nn.y * nn.y + nn.z * nn.z); nn.x /= l; nn.y /= l; nn.z /= l; }   // unit normal
    if (kind == 1) nn = make_float4(0.f, 0.f, 0.f, 0.f);                                                                  // no normal: C = I
    if (kind == 2) tn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kind == 3) { nn = make_float4(0.f, 0.f, 0.f, 0.f); tn = nn; }
    if (kind == 4) nn.y = NAN;                                                                                            // non-finite normal: C = I
    if (kind == 5) tn.z = INFINITY;
    if (kind >= 1 && kind <= 5) n_special++;
    float4 p = make_float4(20.f * U(rng), 20.f * U(rng), 3.f * U(rng), 1.f), t = make_float4(p.x + 0.3f * U(rng), p.y + 0.3f * U(rng), p.z + 0.3f * U(rng), 1.f);

    PoseD P;
    pose_from_T(T, P);
    double M6[6], Ma[3], aMa;
    maha_rank1(P.T, 4, P.G, 1.0 - eps, nn, tn, M6);
    const double pt[3] = {(double)p.x, (double)p.y, (double)p.z};
    resid_terms(P.T, M6, pt, t, Ma, aMa);

    // the reference's formula on the oracle's covariances
    float n4[8] = {nn.x, nn.y, nn.z, 0.f, tn.x, tn.y, tn.z, 0.f};
    double C[18];
    lo_cov_from_normals(n4, 2, eps, C);
    double R[9], S[9], Mi[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) R[r * 3 + c] = (double)T[r * 4 + c];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double s = C[9 + r * 3 + c];
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) s += R[r * 3 + a] * C[a * 3 + b] * R[c * 3 + b];
        S[r * 3 + c] = s;
      }
    inv3(S, Mi);
    const double Mref[6] = {Mi[0], Mi[1], Mi[2], Mi[4], Mi[5], Mi[8]};
    double scale = 0.0;
    for (int k = 0; k < 6; k++) scale = std::fmax(scale, std::fabs(Mref[k]));
    for (int k = 0; k < 6; k++) worst_M = std::fmax(worst_M, std::fabs(M6[k] - Mref[k]) / scale);
    // G = R R^T + I
    int q = 0;
    for (int r = 0; r < 3; r++)
      for (int c = r; c < 3; c++) {
        double g = (r == c ? 1.0 : 0.0);
        for (int a = 0; a < 3; a++) g += R[r * 3 + a] * R[c * 3 + a];
        worst_G = std::fmax(worst_G, std::fabs(g - P.G[q++]));
      }
    // residual terms
    double a[3];
    for (int r = 0; r < 3; r++) a[r] = R[r * 3 + 0] * pt[0] + R[r * 3 + 1] * pt[1] + R[r * 3 + 2] * pt[2] + (double)T[r * 4 + 3];
    a[0] -= (double)t.x; a[1] -= (double)t.y; a[2] -= (double)t.z;
    double ma[3] = {Mi[0] * a[0] + Mi[1] * a[1] + Mi[2] * a[2], Mi[3] * a[0] + Mi[4] * a[1] + Mi[5] * a[2], Mi[6] * a[0] + Mi[7] * a[1] + Mi[8] * a[2]};
    double ama = a[0] * ma[0] + a[1] * ma[1] + a[2] * ma[2];
    const double an = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * scale + 1e-300;
    for (int k = 0; k < 3; k++) worst_a = std::fmax(worst_a, std::fabs(Ma[k] - ma[k]) / an);
    worst_a = std::fmax(worst_a, std::fabs(aMa - ama) / (an * an / scale + 1e-300));
  }
  std::printf("cases 20000 (special normals %d)  worst |M - Mref| / max|Mref| = %.3e   worst G = %.3e   worst residual terms = %.3e\n", n_special, worst_M,
              worst_G, worst_a);
  // M is the inverse of a matrix with condition number <= 2 / eps = 2000: 1e-12 leaves three orders over the rounding of either evaluation
  if (!(worst_M < 1e-12) || !(worst_G < 1e-15) || !(worst_a < 1e-11)) {
    std::printf("TERMS_CHECK_FAILED\n");
    return 1;
  }
  std::printf("TERMS_CHECK_OK\n");
  return 0;
}
