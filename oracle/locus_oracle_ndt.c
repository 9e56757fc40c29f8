/* locus_oracle_ndt.c -- CPU restatement of the reference's NDT registration (registration_method: ndt).
 *
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench tools' CPU-baseline legs may use it.
 *
 * Follows multithreaded_gicp/include/multithreaded_ndt/ (pclomp = koide3's ndt_omp, itself PCL's NDT [Magnusson 2009]):
 *   voxel_grid_covariance_omp_impl.hpp:64-97, 166-205, 215-282   target voxel structure (means, covariances, inflation)
 *   voxel_grid_covariance_omp.h:433-466                          radiusSearch over the voxel centroids (KDTREE mode, the default)
 *   ndt_omp_impl.hpp:101-212                                     computeTransformation (Newton + More-Thuente)
 *   ndt_omp_impl.hpp:225-346                                     computeDerivatives (float point derivatives)
 *   ndt_omp_impl.hpp:350-476, 480-530, 576-652                   angle derivatives, point derivatives, updateDerivatives
 *   ndt_omp_impl.hpp:655-748                                     computeHessian / updateHessian (double path)
 *   ndt_omp_impl.hpp:751-853, 855-1049                           updateIntervalMT, trialValueSelectionMT, computeStepLengthMT
 * PARITY UNPINNED: the reference has no NDT test and no stored output; PCL/Eigen internals restated from their published
 * behaviour (Eigen eulerAngles(0,1,2), AngleAxis products through quaternions, JacobiSVD::solve as a pseudo-inverse,
 * SelfAdjointEigenSolver by Jacobi sweeps); float expression order inside Eigen products is not recoverable.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "locus_oracle.h"

struct lo_ndt_grid {
  int n_cells;
  double* mean;    /* [n_cells][3]  leaf.mean_ */
  double* icov;    /* [n_cells][9]  leaf.icov_ (zero for leaves the eigen check rejected: they stay in the kd-tree) */
  float* centroid; /* [n_cells][4]  float centroid, the kd-tree's cloud */
  int* npts;
  lo_tree* tree;
};

void lo_ndt_default_params(lo_ndt_params* p) {
  p->resolution = 1.0f;               /* ndt_omp_impl.hpp:50 */
  p->step_size = 0.1;                 /* :51 */
  p->outlier_ratio = 0.55;            /* :52 */
  p->transformation_epsilon = 0.1;    /* :93 (LOCUS: icp_tf_epsilon) */
  p->max_iterations = 35;             /* :94 (LOCUS: icp_iterations) */
  p->min_points_per_voxel = 6;        /* voxel_grid_covariance_omp.h:186 */
  p->min_covar_eigvalue_mult = 0.01;  /* :187 */
  p->num_threads = 1;
}

typedef struct { int64_t idx; int pt; } vox_ref;
static int vox_ref_cmp(const void* a, const void* b) {
  const vox_ref *x = (const vox_ref*)a, *y = (const vox_ref*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return (x->pt > y->pt) - (x->pt < y->pt);
}

/* VoxelGridCovariance::applyFilter */
lo_ndt_grid* lo_ndt_grid_build(const float* xyz4, int n, const lo_ndt_params* P) {
  float inv = 1.0f / P->resolution; /* inverse_leaf_size_ (float) */
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n; i++) {
    const float* p = xyz4 + 4 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
  }
  int minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; a++) {
    minb[a] = (int)floorf(mn[a] * inv);
    maxb[a] = (int)floorf(mx[a] * inv);
    divb[a] = maxb[a] - minb[a] + 1;
  }
  int64_t mul[3] = {1, divb[0], (int64_t)divb[0] * divb[1]};
  vox_ref* refs = (vox_ref*)malloc(sizeof(vox_ref) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float* p = xyz4 + 4 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int ijk[3];
    for (int a = 0; a < 3; a++) ijk[a] = (int)(floorf(p[a] * inv) - (float)minb[a]);
    refs[m].idx = ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2];
    refs[m].pt = i;
    m++;
  }
  qsort(refs, m, sizeof(vox_ref), vox_ref_cmp); /* std::map order; points of a leaf in input order */
  lo_ndt_grid* g = (lo_ndt_grid*)calloc(1, sizeof(lo_ndt_grid));
  int cap = 0;
  for (int s = 0; s < m;) { int e = s; while (e < m && refs[e].idx == refs[s].idx) e++; if (e - s >= P->min_points_per_voxel) cap++; s = e; }
  g->mean = (double*)malloc(sizeof(double) * 3 * (size_t)(cap + 1));
  g->icov = (double*)calloc((size_t)(cap + 1) * 9, sizeof(double));
  g->centroid = (float*)calloc((size_t)(cap + 1) * 4, sizeof(float));
  g->npts = (int*)malloc(sizeof(int) * (size_t)(cap + 1));
  int c = 0;
  for (int s = 0; s < m;) {
    int e = s;
    while (e < m && refs[e].idx == refs[s].idx) e++;
    int np = e - s;
    if (np >= P->min_points_per_voxel) {
      double sum[3] = {0, 0, 0}, cov[9] = {0};
      float cen[3] = {0, 0, 0};
      for (int k = s; k < e; k++) {
        const float* p = xyz4 + 4 * (size_t)refs[k].pt;
        double d[3] = {p[0], p[1], p[2]};
        for (int a = 0; a < 3; a++) { sum[a] += d[a]; cen[a] += p[a]; }
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) cov[a * 3 + b] += d[a] * d[b];
      }
      double mean[3];
      for (int a = 0; a < 3; a++) { mean[a] = sum[a] / np; g->centroid[4 * c + a] = cen[a] / (float)np; g->mean[3 * c + a] = mean[a]; }
      g->centroid[4 * c + 3] = 1.0f;
      /* cov = (cov_sum - 2 (pt_sum mean^T)) / n + mean mean^T ; cov *= (n - 1) / n */
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double v = (cov[a * 3 + b] - 2.0 * (sum[a] * mean[b])) / np + mean[a] * mean[b];
          cov[a * 3 + b] = v * ((np - 1.0) / np);
        }
      double ev[3], V[9];
      double sym[9];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) sym[a * 3 + b] = cov[(a > b ? a : b) * 3 + (a > b ? b : a)]; /* SelfAdjointEigenSolver reads the lower triangle */
      lo_eig_sym(sym, 3, ev, V); /* ascending, eigenvectors in columns */
      g->npts[c] = np;
      if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) {
        g->npts[c] = -1; /* rejected: icov stays zero, the cell stays searchable (voxel_grid_covariance_omp_impl.hpp:250-254) */
      } else {
        double minev = P->min_covar_eigvalue_mult * ev[2];
        if (ev[0] < minev) {
          ev[0] = minev;
          if (ev[1] < minev) ev[1] = minev;
          /* cov = evecs * diag * evecs^-1 (orthonormal: inverse = transpose) */
          for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
              double v = 0;
              for (int k = 0; k < 3; k++) v += V[a * 3 + k] * ev[k] * V[b * 3 + k];
              cov[a * 3 + b] = v;
            }
        }
        /* icov = cov.inverse() (3x3 cofactors) */
        double* I = g->icov + 9 * (size_t)c;
        double c00 = cov[4] * cov[8] - cov[5] * cov[7], c01 = cov[5] * cov[6] - cov[3] * cov[8], c02 = cov[3] * cov[7] - cov[4] * cov[6];
        double det = cov[0] * c00 + cov[1] * c01 + cov[2] * c02;
        double id = 1.0 / det;
        I[0] = c00 * id; I[1] = (cov[2] * cov[7] - cov[1] * cov[8]) * id; I[2] = (cov[1] * cov[5] - cov[2] * cov[4]) * id;
        I[3] = c01 * id; I[4] = (cov[0] * cov[8] - cov[2] * cov[6]) * id; I[5] = (cov[2] * cov[3] - cov[0] * cov[5]) * id;
        I[6] = c02 * id; I[7] = (cov[1] * cov[6] - cov[0] * cov[7]) * id; I[8] = (cov[0] * cov[4] - cov[1] * cov[3]) * id;
      }
      c++;
    }
    s = e;
  }
  free(refs);
  g->n_cells = c;
  g->tree = c > 0 ? lo_tree_build(g->centroid, c) : NULL;
  return g;
}
void lo_ndt_grid_free(lo_ndt_grid* g) {
  if (!g) return;
  if (g->tree) lo_tree_free(g->tree);
  free(g->mean); free(g->icov); free(g->centroid); free(g->npts);
  free(g);
}
int lo_ndt_grid_cells(const lo_ndt_grid* g, double* mean3, double* icov9, float* centroid4, int cap) {
  int k = g->n_cells < cap ? g->n_cells : cap;
  if (mean3) memcpy(mean3, g->mean, sizeof(double) * 3 * (size_t)k);
  if (icov9) memcpy(icov9, g->icov, sizeof(double) * 9 * (size_t)k);
  if (centroid4) memcpy(centroid4, g->centroid, sizeof(float) * 4 * (size_t)k);
  return g->n_cells;
}

/* ---- transform from the 6-vector: Translation * AngleAxis(x) * AngleAxis(y) * AngleAxis(z), float (ndt_omp_impl.hpp:160-170) ---- */
typedef struct { float w, x, y, z; } qf;
static qf qmul(qf a, qf b) {
  qf r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
void lo_ndt_pose_to_matrix(const double* p6, float* T /*col-major 4x4*/) {
  float ax = (float)p6[3], ay = (float)p6[4], az = (float)p6[5];
  qf qx = {cosf(0.5f * ax), sinf(0.5f * ax), 0.f, 0.f}, qy = {cosf(0.5f * ay), 0.f, sinf(0.5f * ay), 0.f}, qz = {cosf(0.5f * az), 0.f, 0.f, sinf(0.5f * az)};
  qf q = qmul(qmul(qx, qy), qz);
  /* Eigen::QuaternionBase::toRotationMatrix */
  float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
  float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  float R[9] = {1.f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.f - (txx + tyy)};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
  T[12] = (float)p6[0]; T[13] = (float)p6[1]; T[14] = (float)p6[2];
  T[3] = T[7] = T[11] = 0.f; T[15] = 1.f;
}
/* Eigen::MatrixBase::eulerAngles(0, 1, 2) on the float rotation block (ndt_omp_impl.hpp:131-135) */
void lo_ndt_matrix_to_pose(const float* T, double* p6) {
#define M(r, c) T[(c) * 4 + (r)]
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
#undef M
}

/* ---- derivatives ---------------------------------------------------------------------------------------------------------- */
typedef struct {
  double d1, d2, d3;
  float j_ang[8][4];   /* rows a..h of eq. 6.19, as floats (Matrix<float,8,4> j_ang) */
  float h_ang[16][4];  /* rows a2..f3 of eq. 6.21 */
  double jd[8][3], hd[15][3]; /* the double vectors j_ang_a_.. / h_ang_a2_.. used by computeHessian */
} ndt_ctx;

static void gauss_params(const lo_ndt_params* P, ndt_ctx* c) {
  double c1 = 10.0 * (1 - P->outlier_ratio), c2 = P->outlier_ratio / pow(P->resolution, 3);
  c->d3 = -log(c2);
  c->d1 = -log(c1 + c2) - c->d3;
  c->d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - c->d3) / c->d1);
}
static void angle_derivatives(const double* p, ndt_ctx* c) {
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  double J[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
                    {-sy * cz, sy * sz, cy}, {sx * cy * cz, -sx * cy * sz, sx * sy}, {-cx * cy * cz, cx * cy * sz, -cx * sy},
                    {-cy * sz, -cy * cz, 0}, {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0}, {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  double H[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},
                     {cx * cy * cz, -cx * cy * sz, cx * sy}, {sx * cy * cz, -sx * cy * sz, sx * sy},
                     {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0}, {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                     {-cy * cz, cy * sz, sy}, {-sx * sy * cz, sx * sy * sz, sx * cy}, {cx * sy * cz, -cx * sy * sz, -cx * cy},
                     {sy * sz, sy * cz, 0}, {-sx * cy * sz, -sx * cy * cz, 0}, {cx * cy * sz, cx * cy * cz, 0},
                     {-cy * cz, cy * sz, 0}, {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  memset(c->j_ang, 0, sizeof(c->j_ang));
  memset(c->h_ang, 0, sizeof(c->h_ang));
  for (int r = 0; r < 8; r++)
    for (int k = 0; k < 3; k++) { c->jd[r][k] = J[r][k]; c->j_ang[r][k] = (float)J[r][k]; }
  for (int r = 0; r < 15; r++)
    for (int k = 0; k < 3; k++) { c->hd[r][k] = H[r][k]; c->h_ang[r][k] = (float)H[r][k]; }
}
static inline float dot4f(const float* a, const float* b) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

/* one (point, cell) term of computeDerivatives: float point gradient / hessian, updateDerivatives (ndt_omp_impl.hpp:480-530, 576-652) */
static double point_cell_term(const ndt_ctx* c, const double* x /*original point*/, const double* xt /*x_trans - mean*/, const double* icov,
                              double* g6, double* H36, int want_h) {
  float x4[4] = {(float)x[0], (float)x[1], (float)x[2], 0.0f};
  float pg[4][6];
  memset(pg, 0, sizeof(pg));
  pg[0][0] = pg[1][1] = pg[2][2] = 1.0f;
  float xj[8];
  for (int r = 0; r < 8; r++) xj[r] = dot4f(c->j_ang[r], x4);
  pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
  float xt4[4] = {(float)xt[0], (float)xt[1], (float)xt[2], 0.0f};
  float ci[4][4];
  memset(ci, 0, sizeof(ci));
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) ci[a][b] = (float)icov[a * 3 + b];
  float d2 = (float)c->d2;
  float xci[4]; /* x_trans4 * c_inv4 (row vector) */
  for (int b = 0; b < 4; b++) xci[b] = ((xt4[0] * ci[0][b] + xt4[1] * ci[1][b]) + xt4[2] * ci[2][b]) + xt4[3] * ci[3][b];
  float e = expf(-d2 * dot4f(xt4, xci) * 0.5f);
  float score_inc = (float)(-c->d1 * (double)e); /* -gauss_d1_ * e_x_cov_x: double * float, returned through a float */
  e = d2 * e;
  if (e > 1 || e < 0 || e != e) return 0;
  e = (float)((double)e * c->d1); /* e_x_cov_x *= gauss_d1_ */
  float cg[4][6]; /* c_inv4 * point_gradient4 */
  for (int a = 0; a < 4; a++)
    for (int j = 0; j < 6; j++) cg[a][j] = ((ci[a][0] * pg[0][j] + ci[a][1] * pg[1][j]) + ci[a][2] * pg[2][j]) + ci[a][3] * pg[3][j];
  float xcg[6];
  for (int j = 0; j < 6; j++) xcg[j] = ((xt4[0] * cg[0][j] + xt4[1] * cg[1][j]) + xt4[2] * cg[2][j]) + xt4[3] * cg[3][j];
  for (int j = 0; j < 6; j++) g6[j] += (double)(e * xcg[j]);
  if (want_h) {
    float ph[24][6]; /* point_hessian_ (float): 4x1 blocks at rows 12, 16, 20 */
    memset(ph, 0, sizeof(ph));
    float xh[16];
    for (int r = 0; r < 15; r++) xh[r] = dot4f(c->h_ang[r], x4);
    float va[4] = {0, xh[0], xh[1], 0}, vb[4] = {0, xh[2], xh[3], 0}, vc[4] = {0, xh[4], xh[5], 0}, vd[4] = {xh[6], xh[7], xh[8], 0},
          ve[4] = {xh[9], xh[10], xh[11], 0}, vf[4] = {xh[12], xh[13], xh[14], 0};
    for (int k = 0; k < 4; k++) {
      ph[12 + k][3] = va[k]; ph[16 + k][3] = vb[k]; ph[20 + k][3] = vc[k];
      ph[12 + k][4] = vb[k]; ph[16 + k][4] = vd[k]; ph[20 + k][4] = ve[k];
      ph[12 + k][5] = vc[k]; ph[16 + k][5] = ve[k]; ph[20 + k][5] = vf[k];
    }
    float gcg[6][6]; /* point_gradient4^T * (c_inv4 * point_gradient4) */
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++) gcg[a][b] = ((pg[0][a] * cg[0][b] + pg[1][a] * cg[1][b]) + pg[2][a] * cg[2][b]) + pg[3][a] * cg[3][b];
    for (int i = 0; i < 6; i++) {
      float xch[6]; /* x_trans4_x_c_inv4 * point_hessian_.block<4,6>(i*4, 0) */
      for (int j = 0; j < 6; j++) xch[j] = ((xci[0] * ph[i * 4 + 0][j] + xci[1] * ph[i * 4 + 1][j]) + xci[2] * ph[i * 4 + 2][j]) + xci[3] * ph[i * 4 + 3][j];
      for (int j = 0; j < 6; j++) H36[i * 6 + j] += (double)(e * (-d2 * xcg[i] * xcg[j] + xch[j] + gcg[j][i]));
    }
  }
  return (double)score_inc;
}

#define NDT_MAX_NEIGH 256
/* computeDerivatives: returns the score; trans4 = transformed source (what the reference searches with), src4 = original source */
double lo_ndt_derivatives(const lo_ndt_grid* g, const lo_ndt_params* P, const float* src4, const float* trans4, int n, const double* p6,
                          double* grad6, double* hess36, int want_h) {
  ndt_ctx c;
  gauss_params(P, &c);
  angle_derivatives(p6, &c);
  double score = 0;
  for (int k = 0; k < 6; k++) grad6[k] = 0;
  for (int k = 0; k < 36; k++) hess36[k] = 0;
  float r2 = P->resolution * P->resolution;
  for (int i = 0; i < n; i++) {
    const float* xt = trans4 + 4 * (size_t)i;
    const float* xo = src4 + 4 * (size_t)i;
    int32_t nb[NDT_MAX_NEIGH];
    int cnt = g->tree ? lo_radius_search(g->tree, xt, r2, nb, NULL, NDT_MAX_NEIGH) : 0;
    if (cnt > NDT_MAX_NEIGH) cnt = NDT_MAX_NEIGH;
    double sp = 0, gp[6] = {0}, hp[36] = {0};
    double x[3] = {xo[0], xo[1], xo[2]};
    for (int k = 0; k < cnt; k++) {
      const double* mu = g->mean + 3 * (size_t)nb[k];
      double d[3] = {(double)xt[0] - mu[0], (double)xt[1] - mu[1], (double)xt[2] - mu[2]};
      sp += point_cell_term(&c, x, d, g->icov + 9 * (size_t)nb[k], gp, hp, want_h);
    }
    score += sp; /* summed in point order (ndt_omp_impl.hpp:338-343) */
    for (int k = 0; k < 6; k++) grad6[k] += gp[k];
    for (int k = 0; k < 36; k++) hess36[k] += hp[k];
  }
  return score;
}

/* computeHessian / updateHessian: the double path used after the line search moved (ndt_omp_impl.hpp:655-748) */
void lo_ndt_hessian(const lo_ndt_grid* g, const lo_ndt_params* P, const float* src4, const float* trans4, int n, const double* p6_of_last_derivatives,
                    double* hess36) {
  ndt_ctx c;
  gauss_params(P, &c);
  angle_derivatives(p6_of_last_derivatives, &c); /* "unnecessary because only used after regular derivative calculation": same angles */
  for (int k = 0; k < 36; k++) hess36[k] = 0;
  float r2 = P->resolution * P->resolution;
  for (int i = 0; i < n; i++) {
    const float* xt = trans4 + 4 * (size_t)i;
    const float* xo = src4 + 4 * (size_t)i;
    int32_t nb[NDT_MAX_NEIGH];
    int cnt = g->tree ? lo_radius_search(g->tree, xt, r2, nb, NULL, NDT_MAX_NEIGH) : 0;
    if (cnt > NDT_MAX_NEIGH) cnt = NDT_MAX_NEIGH;
    double x[3] = {xo[0], xo[1], xo[2]};
    double pg[3][6];
    memset(pg, 0, sizeof(pg));
    pg[0][0] = pg[1][1] = pg[2][2] = 1.0;
#define DOT3(v) (x[0] * (v)[0] + x[1] * (v)[1] + x[2] * (v)[2])
    pg[1][3] = DOT3(c.jd[0]); pg[2][3] = DOT3(c.jd[1]); pg[0][4] = DOT3(c.jd[2]); pg[1][4] = DOT3(c.jd[3]); pg[2][4] = DOT3(c.jd[4]);
    pg[0][5] = DOT3(c.jd[5]); pg[1][5] = DOT3(c.jd[6]); pg[2][5] = DOT3(c.jd[7]);
    double ph[18][6];
    memset(ph, 0, sizeof(ph));
    double va[3] = {0, DOT3(c.hd[0]), DOT3(c.hd[1])}, vb[3] = {0, DOT3(c.hd[2]), DOT3(c.hd[3])}, vc[3] = {0, DOT3(c.hd[4]), DOT3(c.hd[5])},
           vd[3] = {DOT3(c.hd[6]), DOT3(c.hd[7]), DOT3(c.hd[8])}, ve[3] = {DOT3(c.hd[9]), DOT3(c.hd[10]), DOT3(c.hd[11])},
           vf[3] = {DOT3(c.hd[12]), DOT3(c.hd[13]), DOT3(c.hd[14])};
#undef DOT3
    for (int k = 0; k < 3; k++) {
      ph[9 + k][3] = va[k]; ph[12 + k][3] = vb[k]; ph[15 + k][3] = vc[k];
      ph[9 + k][4] = vb[k]; ph[12 + k][4] = vd[k]; ph[15 + k][4] = ve[k];
      ph[9 + k][5] = vc[k]; ph[12 + k][5] = ve[k]; ph[15 + k][5] = vf[k];
    }
    for (int k = 0; k < cnt; k++) {
      const double* mu = g->mean + 3 * (size_t)nb[k];
      const double* ci = g->icov + 9 * (size_t)nb[k];
      double d[3] = {(double)xt[0] - mu[0], (double)xt[1] - mu[1], (double)xt[2] - mu[2]};
      double cd[3];
      for (int a = 0; a < 3; a++) cd[a] = ci[a * 3] * d[0] + ci[a * 3 + 1] * d[1] + ci[a * 3 + 2] * d[2];
      double e = c.d2 * exp(-c.d2 * (d[0] * cd[0] + d[1] * cd[1] + d[2] * cd[2]) / 2);
      if (e > 1 || e < 0 || e != e) continue;
      e *= c.d1;
      double cg[3][6]; /* c_inv * point_gradient_.col(j) */
      for (int j = 0; j < 6; j++)
        for (int a = 0; a < 3; a++) cg[a][j] = ci[a * 3] * pg[0][j] + ci[a * 3 + 1] * pg[1][j] + ci[a * 3 + 2] * pg[2][j];
      for (int i2 = 0; i2 < 6; i2++) {
        double xd_i = d[0] * cg[0][i2] + d[1] * cg[1][i2] + d[2] * cg[2][i2];
        for (int j = 0; j < 6; j++) {
          double xd_j = d[0] * cg[0][j] + d[1] * cg[1][j] + d[2] * cg[2][j];
          double chv[3];
          for (int a = 0; a < 3; a++) chv[a] = ci[a * 3] * ph[3 * i2 + 0][j] + ci[a * 3 + 1] * ph[3 * i2 + 1][j] + ci[a * 3 + 2] * ph[3 * i2 + 2][j];
          double t2 = d[0] * chv[0] + d[1] * chv[1] + d[2] * chv[2];
          double t3 = pg[0][j] * cg[0][i2] + pg[1][j] * cg[1][i2] + pg[2][j] * cg[2][i2];
          hess36[i2 * 6 + j] += e * (-c.d2 * xd_i * xd_j + t2 + t3);
        }
      }
    }
  }
}

/* ---- 6x6 solve through a one-sided Jacobi SVD (stands in for Eigen::JacobiSVD<Matrix6d>(H, FullU|FullV).solve(b)) -------- */
void lo_svd_solve6(const double* A36, const double* b6, double* x6) {
  double U[36], V[36];
  memcpy(U, A36, sizeof(U)); /* columns of U converge to u_k * sigma_k */
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
  double thr = 6.0 * 2.220446049250313e-16 * smax; /* JacobiSVD default threshold: diagSize * epsilon (relative to the largest) */
  for (int i = 0; i < 6; i++) x6[i] = 0;
  for (int j = 0; j < 6; j++) {
    if (!(sig[j] > thr)) continue;
    double ub = 0; /* (u_j . b) / sigma_j with u_j = U_col / sigma_j */
    for (int k = 0; k < 6; k++) ub += U[k * 6 + j] * b6[k];
    ub /= sig[j] * sig[j];
    for (int i = 0; i < 6; i++) x6[i] += V[i * 6 + j] * ub;
  }
}

/* ---- More-Thuente (ndt_omp_impl.hpp:751-853) ------------------------------------------------------------------------------ */
static int update_interval(double* a_l, double* f_l, double* g_l, double* a_u, double* f_u, double* g_u, double a_t, double f_t, double g_t) {
  if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
  else if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  else if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  return 1;
}
static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    if (a_t > a_l) return fmin(a_t + 0.66 * (a_u - a_t), a_n);
    return fmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

static void transform_cloud(const float* T, const float* in4, int n, float* out4) { lo_transform(in4, NULL, n, T, out4, NULL); }

/* pcl::Registration::align + computeTransformation */
int lo_ndt_align(const float* src4, int n, const float* tgt4, int m, const lo_ndt_params* P, const float* guess16, lo_ndt_result* out) {
  lo_ndt_grid* g = lo_ndt_grid_build(tgt4, m, P);
  float* trans = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float final_T[16];
  memcpy(final_T, I16, sizeof(I16));
  memcpy(trans, src4, sizeof(float) * 4 * (size_t)n); /* align(): output = input */
  int guess_is_identity = 1;
  if (guess16)
    for (int k = 0; k < 16; k++)
      if (guess16[k] != I16[k]) guess_is_identity = 0;
  if (!guess_is_identity) { memcpy(final_T, guess16, sizeof(I16)); transform_cloud(guess16, src4, n, trans); }
  double p[6], dp[6], grad[6], H[36];
  lo_ndt_matrix_to_pose(final_T, p);
  int iters = 0, converged = 0, evals = 1;
  double score = lo_ndt_derivatives(g, P, src4, trans, n, p, grad, H, 1);
  double last_p_for_hessian[6];
  memcpy(last_p_for_hessian, p, sizeof(p));
  while (!converged) {
    double ng[6];
    for (int k = 0; k < 6; k++) ng[k] = -grad[k];
    lo_svd_solve6(H, ng, dp);
    double norm = 0;
    for (int k = 0; k < 6; k++) norm += dp[k] * dp[k];
    norm = sqrt(norm);
    if (norm == 0 || norm != norm) { converged = norm == norm; break; }
    for (int k = 0; k < 6; k++) dp[k] /= norm;
    /* ---- computeStepLengthMT(p, dp, norm, step_size, tf_eps / 2, score, grad, H, trans) ---- */
    double step_max = P->step_size, step_min = P->transformation_epsilon / 2;
    double phi_0 = -score, d_phi_0 = 0;
    for (int k = 0; k < 6; k++) d_phi_0 -= grad[k] * dp[k];
    double a_t = 0;
    int do_search = 1;
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) do_search = 0;
      else { d_phi_0 = -d_phi_0; for (int k = 0; k < 6; k++) dp[k] = -dp[k]; }
    }
    if (do_search) {
      const double mu = 1.e-4, nu = 0.9;
      int step_iterations = 0;
      double a_l = 0, a_u = 0;
      double f_l = phi_0 - phi_0 - mu * d_phi_0 * a_l, g_l = d_phi_0 - mu * d_phi_0;
      double f_u = phi_0 - phi_0 - mu * d_phi_0 * a_u, g_u = d_phi_0 - mu * d_phi_0;
      int interval_converged = (step_max - step_min) < 0, open_interval = 1;
      a_t = norm;
      a_t = fmin(a_t, step_max);
      a_t = fmax(a_t, step_min);
      double x_t[6];
      for (int k = 0; k < 6; k++) x_t[k] = p[k] + dp[k] * a_t;
      lo_ndt_pose_to_matrix(x_t, final_T);
      transform_cloud(final_T, src4, n, trans);
      score = lo_ndt_derivatives(g, P, src4, trans, n, x_t, grad, H, 1);
      memcpy(last_p_for_hessian, x_t, sizeof(x_t));
      evals++;
      double phi_t = -score, d_phi_t = 0;
      for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * dp[k];
      double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
      while (!interval_converged && step_iterations < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        if (open_interval) a_t = trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else a_t = trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = fmin(a_t, step_max);
        a_t = fmax(a_t, step_min);
        for (int k = 0; k < 6; k++) x_t[k] = p[k] + dp[k] * a_t;
        lo_ndt_pose_to_matrix(x_t, final_T);
        transform_cloud(final_T, src4, n, trans);
        double Hdummy[36];
        score = lo_ndt_derivatives(g, P, src4, trans, n, x_t, grad, Hdummy, 0); /* compute_hessian = false: `hessian` is zeroed */
        memset(H, 0, sizeof(H));
        memcpy(last_p_for_hessian, x_t, sizeof(x_t));
        evals++;
        phi_t = -score;
        d_phi_t = 0;
        for (int k = 0; k < 6; k++) d_phi_t -= grad[k] * dp[k];
        psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
        d_psi_t = d_phi_t - mu * d_phi_0;
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
          open_interval = 0;
          f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
          f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
        }
        if (open_interval) interval_converged = update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
        else interval_converged = update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
      }
      if (step_iterations) lo_ndt_hessian(g, P, src4, trans, n, last_p_for_hessian, H);
    }
    /* ---- back in computeTransformation ---- */
    norm = a_t;
    for (int k = 0; k < 6; k++) { dp[k] *= norm; p[k] += dp[k]; }
    if (iters > P->max_iterations || (iters && fabs(norm) < P->transformation_epsilon)) converged = 1;
    iters++;
  }
  memcpy(out->T, final_T, sizeof(final_T));
  out->converged = converged;
  out->iterations = iters;
  out->evaluations = evals;
  out->trans_probability = n > 0 ? score / (double)n : 0.0;
  out->n_cells = g->n_cells;
  free(trans);
  lo_ndt_grid_free(g);
  return 0;
}
