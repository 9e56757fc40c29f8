/*
 * locus_oracle.c -- CPU restatement of the LOCUS GICP registration hot path (see locus_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: checker + timed CPU baseline.  Never linked into the product.
 * Every function cites the reference file:line it restates (paths relative to the LOCUS repo).
 * Third-party arithmetic that is NOT vendored in the reference (PCL 1.10 registration/bfgs.h,
 * FLANN, Eigen, common_nebula_slam) is restated from its published algorithm; where bit-level
 * behaviour cannot be pinned it says "parity unpinned".
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: float distances must be ((dx*dx+dy*dy)+dz*dz) with one rounding per op.
 */
#define _GNU_SOURCE
#include "locus_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void lo_default_params(lo_params* p) {
  /* gicp.h:111-132 class defaults */
  p->max_iterations = 200;
  p->max_inner_iterations = 20;
  p->corr_dist = 5.0;
  p->transformation_epsilon = 5e-4;
  p->rotation_epsilon = 2e-3;
  p->gicp_epsilon = 1e-3;
  p->k_correspondences = 20;
  p->recompute_source_cov = 0;
  p->recompute_target_cov = 0;
  p->num_threads = 1;
  p->parallel_cost = 0;
}

/* ------------------------------------------------------------------------------------------
 * Exact nearest-neighbour index.  Stands in for pcl::search::KdTree -> FLANN KDTreeSingleIndex
 * (gicp.h:385-391, gicp.hpp:108-109).  FLANN's L2_Simple accumulates diff*diff in float over
 * x,y,z in order; ties are implementation-defined in FLANN ("parity unpinned") -- here: lowest
 * original index.  Nodes carry tight float boxes; the box distance uses the same float
 * operation order as the point distance, so box_d2 <= point_d2 holds exactly (monotone rounding)
 * and pruning on box_d2 > best is exact.
 * ------------------------------------------------------------------------------------------ */
#define LO_LEAF 15

typedef struct {
  float lo[3], hi[3];
  int left, right; /* children; -1 for leaf */
  int begin, end;  /* point range in the reordered arrays */
} lo_node;

struct lo_tree {
  int n;
  float* pts; /* reordered, 3 floats per point */
  int* idx;   /* original index of reordered point */
  lo_node* nodes;
  int nnodes, cap;
};

static inline float d2f(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

static inline float boxd2(const float* q, const float* lo, const float* hi) {
  float dx = fmaxf(fmaxf(lo[0] - q[0], q[0] - hi[0]), 0.0f);
  float dy = fmaxf(fmaxf(lo[1] - q[1], q[1] - hi[1]), 0.0f);
  float dz = fmaxf(fmaxf(lo[2] - q[2], q[2] - hi[2]), 0.0f);
  return (dx * dx + dy * dy) + dz * dz;
}

static int tree_new_node(lo_tree* t) {
  if (t->nnodes == t->cap) {
    t->cap = t->cap ? t->cap * 2 : 1024;
    t->nodes = (lo_node*)realloc(t->nodes, sizeof(lo_node) * (size_t)t->cap);
  }
  return t->nnodes++;
}

/* quickselect on (pts, idx) pairs along dim so that element k is in sorted position */
static void select_k(float* pts, int* idx, int lo, int hi, int k, int dim) {
  while (hi - lo > 1) {
    int mid = lo + (hi - lo) / 2;
    float a = pts[3 * lo + dim], b = pts[3 * mid + dim], c = pts[3 * (hi - 1) + dim];
    float pivot = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
    int i = lo, j = hi - 1;
    while (i <= j) {
      while (pts[3 * i + dim] < pivot) i++;
      while (pts[3 * j + dim] > pivot) j--;
      if (i <= j) {
        float tmp[3];
        memcpy(tmp, pts + 3 * i, 12);
        memcpy(pts + 3 * i, pts + 3 * j, 12);
        memcpy(pts + 3 * j, tmp, 12);
        int ti = idx[i];
        idx[i] = idx[j];
        idx[j] = ti;
        i++;
        j--;
      }
    }
    if (k <= j)
      hi = j + 1;
    else if (k >= i)
      lo = i;
    else
      return;
  }
}

static int tree_build_rec(lo_tree* t, int begin, int end) {
  int id = tree_new_node(t);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = begin; i < end; i++)
    for (int d = 0; d < 3; d++) {
      float v = t->pts[3 * i + d];
      if (v < lo[d]) lo[d] = v;
      if (v > hi[d]) hi[d] = v;
    }
  memcpy(t->nodes[id].lo, lo, 12);
  memcpy(t->nodes[id].hi, hi, 12);
  t->nodes[id].begin = begin;
  t->nodes[id].end = end;
  t->nodes[id].left = t->nodes[id].right = -1;
  if (end - begin > LO_LEAF) {
    int dim = 0;
    float ext = hi[0] - lo[0];
    if (hi[1] - lo[1] > ext) { ext = hi[1] - lo[1]; dim = 1; }
    if (hi[2] - lo[2] > ext) { ext = hi[2] - lo[2]; dim = 2; }
    if (ext > 0.0f) {
      int mid = begin + (end - begin) / 2;
      select_k(t->pts, t->idx, begin, end, mid, dim);
      int l = tree_build_rec(t, begin, mid);
      int r = tree_build_rec(t, mid, end);
      t->nodes[id].left = l;
      t->nodes[id].right = r;
    }
    /* ext == 0: all points identical -> keep as one (big) leaf */
  }
  return id;
}

lo_tree* lo_tree_build(const float* xyz4, int n) {
  lo_tree* t = (lo_tree*)calloc(1, sizeof(lo_tree));
  t->n = n;
  t->pts = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  t->idx = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    t->pts[3 * i + 0] = xyz4[4 * i + 0];
    t->pts[3 * i + 1] = xyz4[4 * i + 1];
    t->pts[3 * i + 2] = xyz4[4 * i + 2];
    t->idx[i] = i;
  }
  if (n > 0) tree_build_rec(t, 0, n);
  return t;
}

void lo_tree_free(lo_tree* t) {
  if (!t) return;
  free(t->pts);
  free(t->idx);
  free(t->nodes);
  free(t);
}

/* lexicographic (d2, idx) "better than" */
static inline int better(float d, int i, float bd, int bi) { return d < bd || (d == bd && i < bi); }

static void nn1_rec(const lo_tree* t, int node, const float* q, float* bd, int* bi) {
  const lo_node* nd = &t->nodes[node];
  if (nd->left < 0) {
    for (int i = nd->begin; i < nd->end; i++) {
      float d = d2f(q, t->pts + 3 * i);
      if (better(d, t->idx[i], *bd, *bi)) { *bd = d; *bi = t->idx[i]; }
    }
    return;
  }
  float dl = boxd2(q, t->nodes[nd->left].lo, t->nodes[nd->left].hi);
  float dr = boxd2(q, t->nodes[nd->right].lo, t->nodes[nd->right].hi);
  int first = nd->left, second = nd->right;
  float df = dl, ds = dr;
  if (dr < dl) { first = nd->right; second = nd->left; df = dr; ds = dl; }
  if (df <= *bd) nn1_rec(t, first, q, bd, bi);
  if (ds <= *bd) nn1_rec(t, second, q, bd, bi);
}

static void nn1_one(const lo_tree* t, const float* q, int32_t* idx, float* d2) {
  float bd = INFINITY;
  int bi = 0x7fffffff;
  if (t->n > 0) nn1_rec(t, 0, q, &bd, &bi);
  /* no neighbour for a query that is not finite (NaN: nothing compares closer; Inf: every distance is +inf) -- the query
     pcl::KdTreeFLANN::nearestKSearch's point_representation_->isValid() rejects */
  *idx = (t->n > 0 && bi != 0x7fffffff && bd < INFINITY) ? bi : -1;
  *d2 = bd;
}

void lo_nn1(const lo_tree* t, const float* q4, int nq, int32_t* idx, float* d2, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) if (threads > 1)
  for (int i = 0; i < nq; i++) nn1_one(t, q4 + 4 * (size_t)i, &idx[i], &d2[i]);
}

void lo_nn1_brute(const float* xyz4, int n, const float* q4, int nq, int32_t* idx, float* d2) {
  for (int i = 0; i < nq; i++) {
    float bd = INFINITY;
    int bi = -1;
    for (int j = 0; j < n; j++) {
      float d = d2f(q4 + 4 * (size_t)i, xyz4 + 4 * (size_t)j);
      if (d < bd) { bd = d; bi = j; } /* ascending j => ties keep the lowest index */
    }
    idx[i] = bi;
    d2[i] = bd;
  }
}

/* k-NN: sorted insertion list of the k best (d2, idx), ascending */
typedef struct { float* d; int* i; int k, cnt; } knn_list;

static inline void knn_insert(knn_list* L, float d, int id) {
  if (L->cnt == L->k && !better(d, id, L->d[L->k - 1], L->i[L->k - 1])) return;
  int p = (L->cnt < L->k) ? L->cnt++ : L->k - 1;
  while (p > 0 && better(d, id, L->d[p - 1], L->i[p - 1])) {
    L->d[p] = L->d[p - 1];
    L->i[p] = L->i[p - 1];
    p--;
  }
  L->d[p] = d;
  L->i[p] = id;
}

static void knn_rec(const lo_tree* t, int node, const float* q, knn_list* L) {
  const lo_node* nd = &t->nodes[node];
  if (nd->left < 0) {
    for (int i = nd->begin; i < nd->end; i++) knn_insert(L, d2f(q, t->pts + 3 * i), t->idx[i]);
    return;
  }
  float dl = boxd2(q, t->nodes[nd->left].lo, t->nodes[nd->left].hi);
  float dr = boxd2(q, t->nodes[nd->right].lo, t->nodes[nd->right].hi);
  int first = nd->left, second = nd->right;
  float df = dl, ds = dr;
  if (dr < dl) { first = nd->right; second = nd->left; df = dr; ds = dl; }
  if (L->cnt < L->k || df <= L->d[L->k - 1]) knn_rec(t, first, q, L);
  if (L->cnt < L->k || ds <= L->d[L->k - 1]) knn_rec(t, second, q, L);
}

void lo_knn(const lo_tree* t, const float* q4, int nq, int k, int32_t* idx, float* d2, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) if (threads > 1)
  for (int i = 0; i < nq; i++) {
    knn_list L = {d2 + (size_t)i * k, idx + (size_t)i * k, k, 0};
    if (t->n > 0) knn_rec(t, 0, q4 + 4 * (size_t)i, &L);
    for (int j = L.cnt; j < k; j++) { L.d[j] = INFINITY; L.i[j] = -1; }
  }
}

void lo_knn_brute(const float* xyz4, int n, const float* q4, int nq, int k, int32_t* idx, float* d2) {
  for (int i = 0; i < nq; i++) {
    knn_list L = {d2 + (size_t)i * k, idx + (size_t)i * k, k, 0};
    for (int j = 0; j < n; j++) knn_insert(&L, d2f(q4 + 4 * (size_t)i, xyz4 + 4 * (size_t)j), j);
    for (int j = L.cnt; j < k; j++) { L.d[j] = INFINITY; L.i[j] = -1; }
  }
}

/* ------------------------------------------------------------------------------------------
 * K6 rigid transform of a cloud (pcl::transformPointCloud at gicp.hpp:440,586;
 * transformPointCloudWithNormals at PointCloudLocalization.cc:197,218,325).  Float arithmetic:
 * y_i = ((m_i0*x + m_i1*y) + m_i2*z) + m_i3 ; n'_i = (m_i0*nx + m_i1*ny) + m_i2*nz.
 * PCL's SSE evaluation order is not pinned in-repo ("parity unpinned"); this order is the one the
 * HIP path reproduces bit-exactly.
 * ------------------------------------------------------------------------------------------ */
static inline void xform_pt(const float* T, const float* p, float* o) {
  /* T column-major: T[c*4 + r] */
  for (int r = 0; r < 3; r++) o[r] = ((T[0 + r] * p[0] + T[4 + r] * p[1]) + T[8 + r] * p[2]) + T[12 + r];
}
static inline void xform_nrm(const float* T, const float* p, float* o) {
  for (int r = 0; r < 3; r++) o[r] = (T[0 + r] * p[0] + T[4 + r] * p[1]) + T[8 + r] * p[2];
}

void lo_transform(const float* xyz4, const float* nrm4, int n, const float* T16, float* out_xyz4, float* out_nrm4) {
  for (int i = 0; i < n; i++) {
    float o[3];
    xform_pt(T16, xyz4 + 4 * (size_t)i, o);
    out_xyz4[4 * (size_t)i + 0] = o[0];
    out_xyz4[4 * (size_t)i + 1] = o[1];
    out_xyz4[4 * (size_t)i + 2] = o[2];
    out_xyz4[4 * (size_t)i + 3] = 1.0f;
    if (nrm4 && out_nrm4) {
      xform_nrm(T16, nrm4 + 4 * (size_t)i, o);
      out_nrm4[4 * (size_t)i + 0] = o[0];
      out_nrm4[4 * (size_t)i + 1] = o[1];
      out_nrm4[4 * (size_t)i + 2] = o[2];
      out_nrm4[4 * (size_t)i + 3] = nrm4[4 * (size_t)i + 3];
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Symmetric eigen-decomposition (cyclic Jacobi, double).  Stands in for Eigen::JacobiSVD<Matrix3d>
 * (gicp.hpp:140) and Eigen::SelfAdjointEigenSolver (utils.cc:130-154).  Ascending eigenvalues,
 * eigenvectors in columns; V row-major.
 * ------------------------------------------------------------------------------------------ */
void lo_eig_sym(const double* Ain, int n, double* evals, double* V) {
  double A[36];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) A[i * n + j] = (i >= j) ? Ain[i * n + j] : Ain[j * n + i]; /* lower triangle, like Eigen */
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) evals[i] = A[i * n + i];
  for (int i = 0; i < n; i++) { /* selection sort ascending */
    int m = i;
    for (int j = i + 1; j < n; j++)
      if (evals[j] < evals[m]) m = j;
    if (m != i) {
      double t = evals[i];
      evals[i] = evals[m];
      evals[m] = t;
      for (int k = 0; k < n; k++) {
        double tv = V[k * n + i];
        V[k * n + i] = V[k * n + m];
        V[k * n + m] = tv;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * K3' CalculateCovarianceFromNormals (call site gicp.hpp:81-82; definition lives in the un-vendored
 * common_nebula_slam/frontend_utils -> "parity unpinned").  Restated, basis-independent form:
 * C = I - (1-eps) * n n^T for the unit normal n (= rotate diag(eps,1,1) so the eps axis is n).
 * Zero / non-finite normal => C = I (assumption, flagged in DESIGN.md).
 * ------------------------------------------------------------------------------------------ */
static void cov_from_normal(const float* nf, double eps, double* C) {
  double n[3] = {nf[0], nf[1], nf[2]};
  double l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  for (int i = 0; i < 9; i++) C[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (!(l2 > 0.0) || !isfinite(l2)) return;
  double inv = 1.0 / sqrt(l2);
  n[0] *= inv; n[1] *= inv; n[2] *= inv;
  double s = 1.0 - eps;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] -= s * n[i] * n[j];
}

void lo_cov_from_normals(const float* nrm4, int n, double eps, double* cov9) {
  for (int i = 0; i < n; i++) cov_from_normal(nrm4 + 4 * (size_t)i, eps, cov9 + 9 * (size_t)i);
}

/* K3 computeCovariances, k-NN branch (gicp.hpp:85-154): double mean/cov over the k neighbours in the
 * order the search returns them (ascending distance), JacobiSVD, C = U diag(1,1,eps) U^T. Because
 * the two unit singular values are equal, C = I - (1-eps) u3 u3^T with u3 the smallest singular vector. */
int lo_cov_knn(const float* xyz4, int n, const lo_tree* t, int k, double eps, double* cov9, int threads) {
  if (k > n) return LO_EINVAL; /* gicp.hpp:72-79 */
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (threads > 1)
  for (int i = 0; i < n; i++) {
    int idx[64];
    float dd[64];
    knn_list L = {dd, idx, k, 0};
    knn_rec(t, 0, xyz4 + 4 * (size_t)i, &L);
    double mean[3] = {0, 0, 0}, cov[9] = {0};
    for (int j = 0; j < k; j++) {
      const float* p = xyz4 + 4 * (size_t)idx[j];
      double x = p[0], y = p[1], z = p[2];
      mean[0] += x; mean[1] += y; mean[2] += z;
      cov[0] += x * x;
      cov[3] += y * x; cov[4] += y * y;
      cov[6] += z * x; cov[7] += z * y; cov[8] += z * z;
    }
    for (int a = 0; a < 3; a++) mean[a] /= (double)k;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b <= a; b++) {
        cov[a * 3 + b] /= (double)k;
        cov[a * 3 + b] -= mean[a] * mean[b];
        cov[b * 3 + a] = cov[a * 3 + b];
      }
    double ev[3], V[9];
    lo_eig_sym(cov, 3, ev, V);
    /* smallest singular value = smallest |eigenvalue| */
    int s = 0;
    for (int a = 1; a < 3; a++)
      if (fabs(ev[a]) < fabs(ev[s])) s = a;
    double u[3] = {V[0 * 3 + s], V[1 * 3 + s], V[2 * 3 + s]};
    double* C = cov9 + 9 * (size_t)i;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) C[a * 3 + b] = ((a == b) ? 1.0 : 0.0) - (1.0 - eps) * u[a] * u[b];
  }
  return LO_OK;
}

/* ------------------------------------------------------------------------------------------
 * K4: NN + Mahalanobis sweep (gicp.hpp:464-498).  M = (R C1 R^T + C2)^-1, Eigen 3x3 inverse =
 * cofactor formula (Eigen/src/LU/InverseImpl.h compute_inverse_size3_helper).
 * ------------------------------------------------------------------------------------------ */
static void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
static void mat3_mul_bt(const double* A, const double* B, double* C) { /* C = A * B^T */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3 + 0] * B[j * 3 + 0] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
static inline double cof3(const double* m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void mat3_inv(const double* m, double* inv) {
  double c00 = cof3(m, 0, 0), c10 = cof3(m, 1, 0), c20 = cof3(m, 2, 0);
  double det = c00 * m[0] + c10 * m[3] + c20 * m[6];
  double invdet = 1.0 / det;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) inv[i * 3 + j] = cof3(m, j, i) * invdet;
}

int lo_nn_mahalanobis(const float* out_xyz4, int n, const lo_tree* tgt_tree, const double* cov_src9,
                      const double* cov_tgt9, const float* T16, const double* R9, double corr_dist,
                      int32_t* tgt_idx, double* maha9, int threads) {
  const double dist_threshold = corr_dist * corr_dist; /* gicp.hpp:438 */
  if (threads < 1) threads = 1;
  int failure = 0; /* gicp.hpp:461: a plain int written from the OMP loop (a benign race: only ever set to 1) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (threads > 1)
  for (int i = 0; i < n; i++) {
    float q[4];
    xform_pt(T16, out_xyz4 + 4 * (size_t)i, q); /* gicp.hpp:469 */
    int32_t j;
    float d;
    nn1_one(tgt_tree, q, &j, &d);
    tgt_idx[i] = -1;
    if (j < 0) { /* searchForNeighbors returned false (gicp.hpp:471-478): no neighbour -- here: a non-finite query */
#pragma omp atomic write
      failure = 1;
      continue;
    }
    if ((double)d < dist_threshold) { /* gicp.hpp:483 */
      double M[9], tmp[9];
      mat3_mul(R9, cov_src9 + 9 * (size_t)i, M);      /* M = R*C1            gicp.hpp:488 */
      mat3_mul_bt(M, R9, tmp);                        /* temp = M*R^T        gicp.hpp:490 */
      const double* C2 = cov_tgt9 + 9 * (size_t)j;
      for (int a = 0; a < 9; a++) tmp[a] += C2[a];    /* temp += C2          gicp.hpp:491 */
      mat3_inv(tmp, maha9 + 9 * (size_t)i);           /* M = temp.inverse()  gicp.hpp:493 */
      tgt_idx[i] = j;
    }
  }
  return failure;
}

/* ------------------------------------------------------------------------------------------
 * applyState (gicp.hpp:619-634): R = AngleAxisf(x5,Z)*AngleAxisf(x4,Y)*AngleAxisf(x3,X) in FLOAT.
 * Eigen evaluates AngleAxis*AngleAxis through float quaternions (Quaternion(aa): w=cos(a/2),
 * v=sin(a/2)*axis; product; toRotationMatrix) -- restated here; t = I so R*I = R and col(3) = x0..2.
 * ------------------------------------------------------------------------------------------ */
typedef struct { float w, x, y, z; } quatf;
static quatf quat_mul(quatf a, quatf b) {
  quatf r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
void lo_apply_state(const double* x, float* T) {
  float hz = 0.5f * (float)x[5], hy = 0.5f * (float)x[4], hx = 0.5f * (float)x[3];
  quatf qz = {cosf(hz), 0.0f, 0.0f, sinf(hz)};
  quatf qy = {cosf(hy), 0.0f, sinf(hy), 0.0f};
  quatf qx = {cosf(hx), sinf(hx), 0.0f, 0.0f};
  quatf q = quat_mul(quat_mul(qz, qy), qx);
  float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  float R[9];
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) T[c * 4 + r] = R[r * 3 + c];
  T[3] = T[7] = T[11] = 0.0f;
  T[12] = (float)x[0]; T[13] = (float)x[1]; T[14] = (float)x[2]; T[15] = 1.0f;
}

/* computeRDerivative (gicp.hpp:160-214) */
static void compute_r_derivative(const double* x, const double* R /*row-major*/, double* g) {
  double phi = x[3], theta = x[4], psi = x[5];
  double cphi = cos(phi), sphi = sin(phi), ctheta = cos(theta), stheta = sin(theta), cpsi = cos(psi), spsi = sin(psi);
  double dPhi[9], dTheta[9], dPsi[9];
#define E(M, r, c) M[(r) * 3 + (c)]
  E(dPhi, 0, 0) = 0; E(dPhi, 1, 0) = 0; E(dPhi, 2, 0) = 0;
  E(dPhi, 0, 1) = sphi * spsi + cphi * cpsi * stheta;
  E(dPhi, 1, 1) = -cpsi * sphi + cphi * spsi * stheta;
  E(dPhi, 2, 1) = cphi * ctheta;
  E(dPhi, 0, 2) = cphi * spsi - cpsi * sphi * stheta;
  E(dPhi, 1, 2) = -cphi * cpsi - sphi * spsi * stheta;
  E(dPhi, 2, 2) = -ctheta * sphi;
  E(dTheta, 0, 0) = -cpsi * stheta; E(dTheta, 1, 0) = -spsi * stheta; E(dTheta, 2, 0) = -ctheta;
  E(dTheta, 0, 1) = cpsi * ctheta * sphi; E(dTheta, 1, 1) = ctheta * sphi * spsi; E(dTheta, 2, 1) = -sphi * stheta;
  E(dTheta, 0, 2) = cphi * cpsi * ctheta; E(dTheta, 1, 2) = cphi * ctheta * spsi; E(dTheta, 2, 2) = -cphi * stheta;
  E(dPsi, 0, 0) = -ctheta * spsi; E(dPsi, 1, 0) = cpsi * ctheta; E(dPsi, 2, 0) = 0;
  E(dPsi, 0, 1) = -cphi * cpsi - sphi * spsi * stheta; E(dPsi, 1, 1) = -cphi * spsi + cpsi * sphi * stheta; E(dPsi, 2, 1) = 0;
  E(dPsi, 0, 2) = cpsi * sphi - cphi * spsi * stheta; E(dPsi, 1, 2) = sphi * spsi + cphi * cpsi * stheta; E(dPsi, 2, 2) = 0;
#undef E
  /* matricesInnerProd (gicp.h:361-370): r += mat1(j,i)*mat2(i,j), i outer, j inner */
  double r3 = 0, r4 = 0, r5 = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      r3 += dPhi[j * 3 + i] * R[i * 3 + j];
      r4 += dTheta[j * 3 + i] * R[i * 3 + j];
      r5 += dPsi[j * 3 + i] * R[i * 3 + j];
    }
  g[3] = r3; g[4] = r4; g[5] = r5;
}

/* ------------------------------------------------------------------------------------------
 * K5: OptimizationFunctorWithIndices (gicp.hpp:291-402).  The reference has three entry points
 * (operator(), df, fdf) that run the same per-correspondence arithmetic; they are restated as one
 * routine with flags so the pass count can still be reported per entry point.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* src;       /* "output" cloud (guess-transformed source) xyz4 */
  const float* tgt;       /* target xyz4 */
  const int32_t* src_idx; /* compacted, gicp.hpp:509-514 */
  const int32_t* tgt_idx;
  int m;
  const double* maha9;    /* mahalanobis_[src index] */
  int parallel;           /* fully-parallel CPU variant */
  int threads;
  long passes;
} cost_ctx;

/* Sensitivity probe (analysis only, see DESIGN.md "float noise floor"): variant 1 evaluates the functor's float
 * T*p with fused multiply-adds, as a -march=native (FMA) build of the reference would; variant 0 (default) = no FMA. */
static int g_cost_variant = 0;
void lo_set_cost_variant(int v) { g_cost_variant = v; }
/* pcl::BFGS deviation switch (no copy of pcl/registration/bfgs.h exists here; tools/golden.py carries the same switch): 1 = the quadratic
   interpolation of Fletcher's line search accepts its stationary point if `c > a` (the reported reading of PCL's port) instead of GSL's `c > 0` */
static int g_bfgs_quad_curv_gt_a = 0;
void lo_set_bfgs_variant(int quad_curv_gt_a) { g_bfgs_quad_curv_gt_a = quad_curv_gt_a; }
static inline void xform_pt_cost(const float* T, const float* p, float* o) {
  if (g_cost_variant == 1) {
    for (int r = 0; r < 3; r++) o[r] = fmaf(T[8 + r], p[2], fmaf(T[4 + r], p[1], T[0 + r] * p[0])) + T[12 + r];
  } else {
    for (int r = 0; r < 3; r++) o[r] = ((T[0 + r] * p[0] + T[4 + r] * p[1]) + T[8 + r] * p[2]) + T[12 + r];
  }
}

static void cost_sums(const cost_ctx* c, const double* x, double* S /*13*/) {
  float T[16];
  lo_apply_state(x, T); /* base_transformation_ = I  (gicp.hpp:435, 367-368) */
  double f = 0, g0 = 0, g1 = 0, g2 = 0, R[9] = {0};
  if (!c->parallel) {
    for (int i = 0; i < c->m; i++) {
      const float* ps = c->src + 4 * (size_t)c->src_idx[i];
      const float* pt = c->tgt + 4 * (size_t)c->tgt_idx[i];
      float pp[3];
      xform_pt_cost(T, ps, pp);                                       /* gicp.hpp:382 */
      double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])}; /* :384 float subtract */
      const double* M = c->maha9 + 9 * (size_t)c->src_idx[i];
      double t0 = M[0] * res[0] + M[1] * res[1] + M[2] * res[2];
      double t1 = M[3] * res[0] + M[4] * res[1] + M[5] * res[2];
      double t2 = M[6] * res[0] + M[7] * res[1] + M[8] * res[2];
      f += res[0] * t0 + res[1] * t1 + res[2] * t2;                   /* :388 */
      g0 += t0; g1 += t1; g2 += t2;                                   /* :392 */
      double p0 = ps[0], p1 = ps[1], p2 = ps[2];                      /* :393-394 base = I */
      R[0] += p0 * t0; R[1] += p0 * t1; R[2] += p0 * t2;              /* :396 */
      R[3] += p1 * t0; R[4] += p1 * t1; R[5] += p1 * t2;
      R[6] += p2 * t0; R[7] += p2 * t1; R[8] += p2 * t2;
    }
  } else {
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, r8 = 0;
#pragma omp parallel for schedule(static) num_threads(c->threads) reduction(+ : f, g0, g1, g2, r0, r1, r2, r3, r4, r5, r6, r7, r8)
    for (int i = 0; i < c->m; i++) {
      const float* ps = c->src + 4 * (size_t)c->src_idx[i];
      const float* pt = c->tgt + 4 * (size_t)c->tgt_idx[i];
      float pp[3];
      xform_pt(T, ps, pp);
      double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
      const double* M = c->maha9 + 9 * (size_t)c->src_idx[i];
      double t0 = M[0] * res[0] + M[1] * res[1] + M[2] * res[2];
      double t1 = M[3] * res[0] + M[4] * res[1] + M[5] * res[2];
      double t2 = M[6] * res[0] + M[7] * res[1] + M[8] * res[2];
      f += res[0] * t0 + res[1] * t1 + res[2] * t2;
      g0 += t0; g1 += t1; g2 += t2;
      double p0 = ps[0], p1 = ps[1], p2 = ps[2];
      r0 += p0 * t0; r1 += p0 * t1; r2 += p0 * t2;
      r3 += p1 * t0; r4 += p1 * t1; r5 += p1 * t2;
      r6 += p2 * t0; r7 += p2 * t1; r8 += p2 * t2;
    }
    R[0] = r0; R[1] = r1; R[2] = r2; R[3] = r3; R[4] = r4; R[5] = r5; R[6] = r6; R[7] = r7; R[8] = r8;
  }
  S[0] = f; S[1] = g0; S[2] = g1; S[3] = g2;
  for (int i = 0; i < 9; i++) S[4 + i] = R[i];
}

static void cost_finish(const double* S, int m, const double* x, double* f, double* g) {
  *f = S[0] / (double)m;                 /* gicp.hpp:398 */
  double s = 2.0 / (double)m;
  g[0] = S[1] * s; g[1] = S[2] * s; g[2] = S[3] * s; /* :399 */
  double R[9];
  for (int i = 0; i < 9; i++) R[i] = S[4 + i] * s;   /* :400 */
  compute_r_derivative(x, R, g);                     /* :401 */
}

void lo_cost_fdf(const float* out_xyz4, const float* tgt_xyz4, const int32_t* src_idx, const int32_t* tgt_idx,
                 int m, const double* maha9, const double* x6, double* f, double* g6, double* sums13) {
  cost_ctx c = {out_xyz4, tgt_xyz4, src_idx, tgt_idx, m, maha9, 0, 1, 0};
  double S[13];
  cost_sums(&c, x6, S);
  cost_finish(S, m, x6, f, g6);
  if (sums13) memcpy(sums13, S, sizeof(S));
}

/* ------------------------------------------------------------------------------------------
 * pcl::BFGS (pcl/registration/bfgs.h, PCL 1.10 -- NOT vendored; an Eigen port of GSL
 * multimin vector_bfgs2 + Fletcher's line search, linear_minimize.c).  Restated from the published
 * GSL algorithm with PCL's deviations as recalled: quadratic roots by the plain formula
 * (PolynomialSolver<Scalar,2> specialisation), "c > a" curvature test quirk NOT reproduced (we use
 * c > 0 as GSL does -- the difference only matters when 0 < c <= a).  "BFGS step-level parity
 * unpinned" (SURVEY.md 8c).  Used by estimateRigidTransformationBFGS (gicp.hpp:218-287).
 * ------------------------------------------------------------------------------------------ */
enum { BFGS_RUNNING = -1, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 1 };

typedef struct {
  cost_ctx* fn;
  /* parameters (gicp.hpp:253-258) */
  double rho, sigma, tau1, tau2, tau3, step_size;
  int order, bracket_iters, section_iters;
  /* state */
  int iter;
  double f, delta_f, fp0, pnorm, g0norm;
  double x0[6], g0[6], dx0[6], dg0[6], p[6], gradient[6], dx[6];
  /* wrapper cache (GSL wrap_* / PCL applyF, applyDF, applyFDF) */
  double x_alpha[6], g_alpha[6], f_alpha, df_alpha, f_cache_key, df_cache_key, x_cache_key, g_cache_key;
} bfgs_t;

static double dot6(const double* a, const double* b) {
  double s = 0;
  for (int i = 0; i < 6; i++) s += a[i] * b[i];
  return s;
}
static double norm6(const double* a) { return sqrt(dot6(a, a)); }

static double fn_f(cost_ctx* c, const double* x) { /* operator() gicp.hpp:291-317 */
  double S[13], f, g[6];
  cost_sums(c, x, S);
  c->passes++;
  cost_finish(S, c->m, x, &f, g);
  return f;
}
static void fn_df(cost_ctx* c, const double* x, double* g) { /* df gicp.hpp:321-358 */
  double S[13], f;
  cost_sums(c, x, S);
  c->passes++;
  cost_finish(S, c->m, x, &f, g);
}
static void fn_fdf(cost_ctx* c, const double* x, double* f, double* g) { /* fdf gicp.hpp:362-402 */
  double S[13];
  cost_sums(c, x, S);
  c->passes++;
  cost_finish(S, c->m, x, f, g);
}

static void bfgs_moveto(bfgs_t* b, double alpha) {
  if (alpha == b->x_cache_key) return;
  for (int i = 0; i < 6; i++) b->x_alpha[i] = b->x0[i] + alpha * b->p[i];
  b->x_cache_key = alpha;
}
static double bfgs_slope(bfgs_t* b) { return dot6(b->g_alpha, b->p); }
static double bfgs_apply_f(bfgs_t* b, double alpha) {
  if (alpha == b->f_cache_key) return b->f_alpha;
  bfgs_moveto(b, alpha);
  b->f_alpha = fn_f(b->fn, b->x_alpha);
  b->f_cache_key = alpha;
  return b->f_alpha;
}
static double bfgs_apply_df(bfgs_t* b, double alpha) {
  if (alpha == b->df_cache_key) return b->df_alpha;
  bfgs_moveto(b, alpha);
  if (alpha != b->g_cache_key) {
    fn_df(b->fn, b->x_alpha, b->g_alpha);
    b->g_cache_key = alpha;
  }
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = alpha;
  return b->df_alpha;
}
static void bfgs_apply_fdf(bfgs_t* b, double alpha, double* f, double* df) {
  if (alpha == b->f_cache_key && alpha == b->df_cache_key) {
    *f = b->f_alpha;
    *df = b->df_alpha;
    return;
  }
  if (alpha == b->f_cache_key || alpha == b->df_cache_key) {
    *f = bfgs_apply_f(b, alpha);
    *df = bfgs_apply_df(b, alpha);
    return;
  }
  bfgs_moveto(b, alpha);
  fn_fdf(b->fn, b->x_alpha, &b->f_alpha, b->g_alpha);
  b->f_cache_key = alpha;
  b->g_cache_key = alpha;
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = alpha;
  *f = b->f_alpha;
  *df = b->df_alpha;
}
static void bfgs_update_position(bfgs_t* b, double alpha, double* x, double* f, double* g) {
  double fa, dfa;
  bfgs_apply_fdf(b, alpha, &fa, &dfa);
  *f = b->f_alpha;
  memcpy(x, b->x_alpha, sizeof(double) * 6);
  memcpy(g, b->g_alpha, sizeof(double) * 6);
}
static void bfgs_change_direction(bfgs_t* b) {
  memcpy(b->x_alpha, b->x0, sizeof(double) * 6);
  b->x_cache_key = 0.0;
  b->f_cache_key = 0.0;
  memcpy(b->g_alpha, b->g0, sizeof(double) * 6);
  b->g_cache_key = 0.0;
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = 0.0;
}

static double cubic_eval(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }

static double bfgs_interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin,
                               double xmax, int order) {
  /* Map [a,b] to [0,1] (GSL linear_minimize.c interpolate) */
  double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a), fmin;
  if (ymin > ymax) { double t = ymin; ymin = ymax; ymax = t; }
  if (order > 2 && !(fpb != fpb) && fpb != INFINITY) {
    fpa = fpa * (b - a);
    fpb = fpb * (b - a);
    double eta = 3 * (fb - fa) - 2 * fpa - fpb;
    double xi = fpa + fpb - 2 * (fb - fa);
    double c0 = fa, c1 = fpa, c2 = eta, c3 = xi;
    y = ymin;
    fmin = cubic_eval(c0, c1, c2, c3, ymin);
    { double v = cubic_eval(c0, c1, c2, c3, ymax); if (v < fmin) { y = ymax; fmin = v; } }
    /* roots of the derivative c1 + 2 c2 z + 3 c3 z^2 (PCL: plain quadratic formula) */
    double qa = 3 * c3, qb = 2 * c2, qc = c1;
    double disc = qb * qb - 4 * qc * qa;
    if (disc > 0) {
      double sd = sqrt(disc);
      double y0 = (-qb - sd) / (2 * qa), y1 = (-qb + sd) / (2 * qa);
      if (y0 > y1) { double t = y0; y0 = y1; y1 = t; }
      if (y0 > ymin && y0 < ymax) { double v = cubic_eval(c0, c1, c2, c3, y0); if (v < fmin) { y = y0; fmin = v; } }
      if (y1 > ymin && y1 < ymax) { double v = cubic_eval(c0, c1, c2, c3, y1); if (v < fmin) { y = y1; fmin = v; } }
    } else if (disc == 0) {
      double y0 = -qb / (2 * qa);
      if (y0 > ymin && y0 < ymax) { double v = cubic_eval(c0, c1, c2, c3, y0); if (v < fmin) { y = y0; fmin = v; } }
    }
  } else {
    fpa = fpa * (b - a);
    double fl = fa + ymin * (fpa + ymin * (fb - fa - fpa));
    double fh = fa + ymax * (fpa + ymax * (fb - fa - fpa));
    double c = 2 * (fb - fa - fpa); /* curvature */
    y = ymin;
    fmin = fl;
    if (fh < fmin) { y = ymax; fmin = fh; }
    if (c > (g_bfgs_quad_curv_gt_a ? a : 0.0)) {
      double z = -fpa / c;
      if (z > ymin && z < ymax) {
        double f = fa + z * (fpa + z * (fb - fa - fpa));
        if (f < fmin) { y = z; fmin = f; }
      }
    }
  }
  return a + y * (b - a);
}

static int bfgs_line_search(bfgs_t* B, double alpha1, double* alpha_new) {
  double rho = B->rho, sigma = B->sigma, tau1 = B->tau1, tau2 = B->tau2, tau3 = B->tau3;
  double f0, fp0, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
  double alpha = alpha1, alpha_prev = 0.0;
  double a, b, fa, fb, fpa, fpb;
  int i = 0;
  bfgs_apply_fdf(B, 0.0, &f0, &fp0);
  falpha_prev = f0;
  fpalpha_prev = fp0;
  a = 0.0; b = alpha; fa = f0; fb = 0.0; fpa = fp0; fpb = 0.0;
  /* bracketing */
  while (i++ < B->bracket_iters) {
    falpha = bfgs_apply_f(B, alpha);
    if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
      a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
      b = alpha; fb = falpha; fpb = NAN;
      break;
    }
    fpalpha = bfgs_apply_df(B, alpha);
    if (fabs(fpalpha) <= -sigma * fp0) { /* Fletcher's sigma test */
      *alpha_new = alpha;
      return BFGS_SUCCESS;
    }
    if (fpalpha >= 0) {
      a = alpha; fa = falpha; fpa = fpalpha;
      b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
      break;
    }
    delta = alpha - alpha_prev;
    {
      double lower = alpha + delta, upper = alpha + tau1 * delta;
      alpha_next = bfgs_interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, lower, upper, B->order);
    }
    alpha_prev = alpha;
    falpha_prev = falpha;
    fpalpha_prev = fpalpha;
    alpha = alpha_next;
  }
  /* sectioning of bracket [a,b] */
  while (i++ < B->section_iters) {
    delta = b - a;
    {
      double lower = a + tau2 * delta, upper = b - tau3 * delta;
      alpha = bfgs_interpolate(a, fa, fpa, b, fb, fpb, lower, upper, B->order);
    }
    falpha = bfgs_apply_f(B, alpha);
    if ((a - alpha) * fpa <= DBL_EPSILON) return BFGS_NOPROGRESS; /* roundoff prevents progress */
    if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
      b = alpha; fb = falpha; fpb = NAN;
    } else {
      fpalpha = bfgs_apply_df(B, alpha);
      if (fabs(fpalpha) <= -sigma * fp0) {
        *alpha_new = alpha;
        return BFGS_SUCCESS;
      }
      if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
        b = a; fb = fa; fpb = fpa;
        a = alpha; fa = falpha; fpa = fpalpha;
      } else {
        a = alpha; fa = falpha; fpa = fpalpha;
      }
    }
  }
  return BFGS_SUCCESS;
}

static void bfgs_init(bfgs_t* b, cost_ctx* fn, const double* x) { /* BFGS::minimizeInit */
  b->fn = fn;
  b->iter = 0;
  b->delta_f = 0;
  memset(b->dx, 0, sizeof(b->dx));
  fn_fdf(fn, x, &b->f, b->gradient);
  memcpy(b->x0, x, sizeof(double) * 6);
  memcpy(b->g0, b->gradient, sizeof(double) * 6);
  b->g0norm = norm6(b->g0);
  for (int i = 0; i < 6; i++) b->p[i] = b->gradient[i] * (-1.0 / b->g0norm);
  b->pnorm = norm6(b->p);
  b->fp0 = -b->g0norm;
  memcpy(b->x_alpha, b->x0, sizeof(double) * 6);
  b->x_cache_key = 0;
  b->f_alpha = b->f;
  b->f_cache_key = 0;
  memcpy(b->g_alpha, b->g0, sizeof(double) * 6);
  b->g_cache_key = 0;
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = 0;
}

static int bfgs_one_step(bfgs_t* b, double* x) { /* BFGS::minimizeOneStep */
  double alpha = 0.0, alpha1, f0 = b->f;
  if (b->pnorm == 0.0 || b->g0norm == 0.0 || b->fp0 == 0) {
    memset(b->dx, 0, sizeof(b->dx));
    return BFGS_NOPROGRESS;
  }
  if (b->delta_f < 0) {
    double del = fmax(-b->delta_f, 10 * DBL_EPSILON * fabs(f0));
    alpha1 = fmin(1.0, 2.0 * del / (-b->fp0));
  } else
    alpha1 = fabs(b->step_size);
  int status = bfgs_line_search(b, alpha1, &alpha);
  if (status != BFGS_SUCCESS) return status;
  bfgs_update_position(b, alpha, x, &b->f, b->gradient);
  b->delta_f = b->f - f0;
  {
    double dxg, dgg, dxdg, dgnorm, A, Bc;
    for (int i = 0; i < 6; i++) { b->dx0[i] = x[i] - b->x0[i]; b->dx[i] = b->dx0[i]; b->dg0[i] = b->gradient[i] - b->g0[i]; }
    dxg = dot6(b->dx0, b->gradient);
    dgg = dot6(b->dg0, b->gradient);
    dxdg = dot6(b->dx0, b->dg0);
    dgnorm = norm6(b->dg0);
    if (dxdg != 0) {
      Bc = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * Bc + dgg / dxdg;
    } else {
      Bc = 0;
      A = 0;
    }
    for (int i = 0; i < 6; i++) b->p[i] = -A * b->dx0[i];
    for (int i = 0; i < 6; i++) b->p[i] += b->gradient[i];
    for (int i = 0; i < 6; i++) b->p[i] += -Bc * b->dg0[i];
  }
  memcpy(b->g0, b->gradient, sizeof(double) * 6);
  memcpy(b->x0, x, sizeof(double) * 6);
  b->g0norm = norm6(b->g0);
  b->pnorm = norm6(b->p);
  double dir = (dot6(b->p, b->gradient) > 0) ? -1.0 : 1.0;
  for (int i = 0; i < 6; i++) b->p[i] *= dir / b->pnorm;
  b->pnorm = norm6(b->p);
  b->fp0 = dot6(b->p, b->g0);
  bfgs_change_direction(b);
  return BFGS_SUCCESS;
}

/* estimateRigidTransformationBFGS (gicp.hpp:218-287). T16 in/out column-major float. */
static int estimate_rigid_bfgs(cost_ctx* c, int max_inner, float* T16, int* n_inner, double* f_end) {
  if (c->m < 4) return LO_ETOO_FEW; /* gicp.hpp:225 */
#define TM(r, col) ((double)T16[(col) * 4 + (r)])
  double x[6];
  x[0] = TM(0, 3); x[1] = TM(1, 3); x[2] = TM(2, 3);
  x[3] = atan2(TM(2, 1), TM(2, 2));
  x[4] = asin(-TM(2, 0));
  x[5] = atan2(TM(1, 0), TM(0, 0));
#undef TM
  const double gradient_tol = 1e-2;
  bfgs_t b;
  memset(&b, 0, sizeof(b));
  b.sigma = 0.01; b.rho = 0.01; b.tau1 = 9; b.tau2 = 0.05; b.tau3 = 0.5; b.order = 3; /* gicp.hpp:253-258 */
  b.step_size = 1; b.bracket_iters = 100; b.section_iters = 100;                      /* pcl::BFGS::Parameters defaults */
  int inner = 0, result;
  bfgs_init(&b, c, x);
  do {
    inner++;
    result = bfgs_one_step(&b, x);
    if (result) break;
    result = (norm6(b.gradient) < gradient_tol) ? BFGS_SUCCESS : BFGS_RUNNING; /* testGradient */
  } while (result == BFGS_RUNNING && inner < max_inner);
  *n_inner = inner;
  *f_end = b.f;
  if (result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || inner == max_inner) {
    lo_apply_state(x, T16); /* setIdentity + applyState gicp.hpp:277-278 */
    return LO_OK;
  }
  return LO_ESOLVER;
}

int lo_estimate_rigid_bfgs(const float* out_xyz4, const float* tgt_xyz4, const int32_t* src_idx, const int32_t* tgt_idx, int m,
                           const double* maha9, int max_inner, float* T16, int* n_inner, double* f_end, int* passes) {
  cost_ctx c = {out_xyz4, tgt_xyz4, src_idx, tgt_idx, m, maha9, 0, 1, 0};
  int st = estimate_rigid_bfgs(&c, max_inner, T16, n_inner, f_end);
  if (passes) *passes = (int)c.passes;
  return st;
}

/* The same solve with its inner steps written down (tests/test_second_restatement.py: held against tools/golden.py's independently written
 * vector_bfgs2 + Fletcher line search): after minimizeInit (row 0) and after every minimizeOneStep that returned success, the state x, the
 * cost f, the gradient norm and the number of functor evaluations so far.  Returns the number of rows; *result = the last step's return
 * value (BFGS_SUCCESS = gradient below 1e-2, BFGS_NOPROGRESS, BFGS_RUNNING = max_inner reached). */
int lo_estimate_rigid_bfgs_trace(const float* out_xyz4, const float* tgt_xyz4, const int32_t* src_idx, const int32_t* tgt_idx, int m,
                                 const double* maha9, int max_inner, const double* x_start6, double* xs, double* fs, double* gnorms, int* evals,
                                 int cap, int* result) {
  cost_ctx c = {out_xyz4, tgt_xyz4, src_idx, tgt_idx, m, maha9, 0, 1, 0};
  double x[6];
  memcpy(x, x_start6, sizeof x);
  bfgs_t b;
  memset(&b, 0, sizeof(b));
  b.sigma = 0.01; b.rho = 0.01; b.tau1 = 9; b.tau2 = 0.05; b.tau3 = 0.5; b.order = 3;
  b.step_size = 1; b.bracket_iters = 100; b.section_iters = 100;
  int rows = 0, inner = 0, res;
  bfgs_init(&b, &c, x);
  if (rows < cap) { memcpy(xs, x, sizeof x); fs[0] = b.f; gnorms[0] = norm6(b.gradient); evals[0] = (int)c.passes; rows = 1; }
  do {
    inner++;
    res = bfgs_one_step(&b, x);
    if (res) break;
    if (rows < cap) { memcpy(xs + 6 * rows, x, sizeof x); fs[rows] = b.f; gnorms[rows] = norm6(b.gradient); evals[rows] = (int)c.passes; rows++; }
    res = (norm6(b.gradient) < 1e-2) ? BFGS_SUCCESS : BFGS_RUNNING;
  } while (res == BFGS_RUNNING && inner < max_inner);
  if (result) *result = res;
  return rows;
}

/* ------------------------------------------------------------------------------------------
 * a3 + a6: pcl::Registration::align -> computeTransformation (gicp.hpp:406-617)
 * ------------------------------------------------------------------------------------------ */
static void mat4f_mul(const float* A, const float* B, float* C) { /* column-major C = A*B, float */
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
      C[c * 4 + r] = s;
    }
}

int lo_gicp_align(const float* src_xyz4, const float* src_nrm4, int n, const float* tgt_xyz4,
                  const float* tgt_nrm4, int m, const lo_params* P, const float* guess16, lo_result* res,
                  lo_trace* trace, float* aligned_xyz4) {
  double t_start = now_s();
  memset(res, 0, sizeof(*res));
  static const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(res->T, I16, sizeof(I16));
  if (n <= 0 || m <= 0 || !src_xyz4 || !tgt_xyz4) { res->status = LO_EINVAL; return LO_EINVAL; }
  if ((!P->recompute_source_cov && !src_nrm4) || (!P->recompute_target_cov && !tgt_nrm4)) { res->status = LO_EINVAL; return LO_EINVAL; }
  if (P->k_correspondences > 64) { res->status = LO_EINVAL; return LO_EINVAL; }
  const float* guess = guess16 ? guess16 : I16;
  int threads = P->num_threads < 1 ? 1 : P->num_threads;
  if (trace) trace->n_iters = 0;

  /* initCompute: target kd-tree (a3) */
  double t0 = now_s();
  lo_tree* tree = lo_tree_build(tgt_xyz4, m);
  lo_tree* tree_src = NULL;
  res->t_index = now_s() - t0;

  /* covariances (gicp.hpp:422-432) */
  t0 = now_s();
  double* cov_tgt = (double*)malloc(sizeof(double) * 9 * (size_t)m);
  double* cov_src = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  double* maha = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  int32_t* tgt_idx_full = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* src_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* tgt_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  float* output = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  int rc = LO_OK;
  if (P->recompute_target_cov)
    rc = lo_cov_knn(tgt_xyz4, m, tree, P->k_correspondences, P->gicp_epsilon, cov_tgt, threads);
  else
    lo_cov_from_normals(tgt_nrm4, m, P->gicp_epsilon, cov_tgt);
  if (rc == LO_OK) {
    if (P->recompute_source_cov) {
      tree_src = lo_tree_build(src_xyz4, n); /* initComputeReciprocal gicp.hpp:412 */
      rc = lo_cov_knn(src_xyz4, n, tree_src, P->k_correspondences, P->gicp_epsilon, cov_src, threads);
    } else
      lo_cov_from_normals(src_nrm4, n, P->gicp_epsilon, cov_src);
  }
  res->t_cov = now_s() - t0;
  if (rc != LO_OK) goto done;

  {
    float transformation[16], previous[16];
    memcpy(transformation, I16, sizeof(I16)); /* align(): transformation_ = I */
    memcpy(previous, I16, sizeof(I16));
    int nr_iterations = 0, converged = 0;
    double delta = 0.0;
    /* mahalanobis_.resize(N, I) gicp.hpp:418 */
    for (int i = 0; i < n; i++)
      for (int a = 0; a < 9; a++) maha[9 * (size_t)i + a] = (a % 4 == 0) ? 1.0 : 0.0;
    lo_transform(src_xyz4, NULL, n, guess, output, NULL); /* gicp.hpp:440 */

    while (!converged) {
      /* transform_R = double(transformation_) * double(guess)   gicp.hpp:450-458 */
      double TR[16];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          double s = 0.0;
          for (int k = 0; k < 4; k++) s += (double)transformation[k * 4 + i] * (double)guess[j * 4 + k];
          TR[i * 4 + j] = s;
        }
      double R9[9] = {TR[0], TR[1], TR[2], TR[4], TR[5], TR[6], TR[8], TR[9], TR[10]};
      t0 = now_s();
      int failure = lo_nn_mahalanobis(output, n, tree, cov_src, cov_tgt, transformation, R9, P->corr_dist, tgt_idx_full, maha, threads);
      res->t_nn += now_s() - t0;
      if (failure) { /* gicp.hpp:504-506: return; final_transformation_ keeps what pcl::Registration::align reset it to (identity),
                        converged_ stays false, nr_iterations_ is what it was */
        res->status = LO_ENO_NN;
        res->converged = 0;
        res->iterations = nr_iterations;
        goto done;
      }
      int cnt = 0; /* compaction gicp.hpp:509-514 */
      for (int i = 0; i < n; i++)
        if (tgt_idx_full[i] >= 0) { src_idx[cnt] = i; tgt_idx[cnt] = tgt_idx_full[i]; cnt++; }
      memcpy(previous, transformation, sizeof(previous)); /* gicp.hpp:518 */
      t0 = now_s();
      cost_ctx c = {output, tgt_xyz4, src_idx, tgt_idx, cnt, maha, P->parallel_cost, threads, 0};
      int n_inner = 0;
      double f_end = 0;
      int st = estimate_rigid_bfgs(&c, P->max_inner_iterations, transformation, &n_inner, &f_end);
      res->t_opt += now_s() - t0;
      res->total_passes += c.passes;
      res->n_corr_last = cnt;
      if (st != LO_OK) { /* exception -> caught -> break  gicp.hpp:542-547 */
        res->status = st;
        break;
      }
      delta = 0.0; /* gicp.hpp:526-541 */
      for (int k = 0; k < 4; k++)
        for (int l = 0; l < 4; l++) {
          double ratio = (k < 3 && l < 3) ? 1.0 / P->rotation_epsilon : 1.0 / P->transformation_epsilon;
          /* fabs(previous_transformation_(k, l) - transformation_(k, l)): two Matrix4f entries, a FLOAT subtraction (gicp.hpp:535-536) */
          double c_delta = ratio * (double)fabsf(previous[l * 4 + k] - transformation[l * 4 + k]);
          if (c_delta > delta) delta = c_delta;
        }
      if (trace && nr_iterations < LO_MAX_TRACE) {
        int it = nr_iterations;
        memcpy(trace->T[it], transformation, sizeof(transformation));
        trace->n_corr[it] = cnt;
        trace->n_passes[it] = (int)c.passes;
        trace->n_inner[it] = n_inner;
        trace->f_end[it] = f_end;
        trace->delta[it] = delta;
        trace->n_iters = it + 1;
      }
      nr_iterations++;
      if (nr_iterations >= P->max_iterations || delta < 1) { /* gicp.hpp:566 */
        converged = 1;
        memcpy(previous, transformation, sizeof(previous));
      }
    }
    mat4f_mul(previous, guess, res->T); /* final_transformation_ = previous_transformation_ * guess  gicp.hpp:583 */
    res->converged = converged;
    res->iterations = nr_iterations;
    if (aligned_xyz4) lo_transform(src_xyz4, NULL, n, res->T, aligned_xyz4, NULL); /* gicp.hpp:586 */
  }
done:
  if (rc != LO_OK) res->status = rc;
  free(cov_tgt); free(cov_src); free(maha); free(tgt_idx_full); free(src_idx); free(tgt_idx); free(output);
  lo_tree_free(tree);
  lo_tree_free(tree_src);
  res->t_total = now_s() - t_start;
  return res->status;
}

/* K7 pcl::Registration::getFitnessScore(max_range = DBL_MAX): mean of float NN d2 summed in double */
double lo_fitness(const float* src_xyz4, int n, const float* T16, const lo_tree* tgt_tree, int threads) {
  float* d2 = (float*)malloc(sizeof(float) * (size_t)n);
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  float* q = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  lo_transform(src_xyz4, NULL, n, T16, q, NULL);
  lo_nn1(tgt_tree, q, n, idx, d2, threads);
  double s = 0.0;
  int nr = 0;
  for (int i = 0; i < n; i++)
    if (idx[i] >= 0) { s += (double)d2[i]; nr++; }
  free(d2); free(idx); free(q);
  return nr > 0 ? s / nr : DBL_MAX;
}

/* ------------------------------------------------------------------------------------------
 * K8: normalizePCloud (utils.cc:106-128) + ComputeAp_ForPoint2PlaneICP (PointCloudLocalization.cc:723-750)
 * ------------------------------------------------------------------------------------------ */
void lo_normalize_cloud(const float* xyz4, int n, float* out) {
  /* pcl::compute3DCentroid: float accumulation of finite points, /count */
  float cx = 0, cy = 0, cz = 0;
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    const float* p = xyz4 + 4 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    cx += p[0]; cy += p[1]; cz += p[2];
    cnt++;
  }
  if (cnt > 0) { cx /= (float)cnt; cy /= (float)cnt; cz /= (float)cnt; }
  float dist = 0;
  for (int i = 0; i < n; i++) {
    const float* p = xyz4 + 4 * (size_t)i;
    float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
    dist = dist + sqrtf((dx * dx + dy * dy) + dz * dz); /* utils.cc:118 */
  }
  float factor = (float)n / dist; /* utils.cc:120 */
  float T[16] = {factor, 0, 0, 0, 0, factor, 0, 0, 0, 0, factor, 0, -factor * cx, -factor * cy, -factor * cz, 1.0f};
  /* utils.cc:124 also writes -factor into T(3,3); pcl::transformPointCloud only uses the top 3 rows */
  lo_transform(xyz4, NULL, n, T, out, NULL);
}

void lo_p2plane_Ap(const float* q, int n, const float* ref_nrm4, const int64_t* corr, double* Ap) {
  for (int i = 0; i < 36; i++) Ap[i] = 0.0;
  for (int i = 0; i < n; i++) {
    double a[3] = {q[4 * (size_t)i], q[4 * (size_t)i + 1], q[4 * (size_t)i + 2]};
    const float* nf = ref_nrm4 + 4 * (size_t)corr[i];
    double nn[3] = {nf[0], nf[1], nf[2]};
    if (isnan(a[0]) || isnan(a[1]) || isnan(a[2]) || isnan(nn[0]) || isnan(nn[1]) || isnan(nn[2])) continue; /* :742 */
    double H[6] = {a[1] * nn[2] - a[2] * nn[1], a[2] * nn[0] - a[0] * nn[2], a[0] * nn[1] - a[1] * nn[0], nn[0], nn[1], nn[2]};
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) Ap[r * 6 + c] += H[r] * H[c];
  }
}

/* H2: 0.05^2 * Ap^-1, Eigen LDLT (pivoted, lower, unblocked) D-clamp, NaN guard, condition number
 * (PointCloudLocalization.cc:487-538).  Eigen's general 6x6 inverse is PartialPivLU; restated as
 * Gauss-Jordan with partial pivoting (same result to rounding). */
static int inv_n(const double* A, int n, double* inv) {
  double a[36 * 2];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) { a[i * 2 * n + j] = A[i * n + j]; a[i * 2 * n + n + j] = (i == j); }
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (fabs(a[r * 2 * n + c]) > fabs(a[piv * 2 * n + c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 2 * n; j++) { double t = a[c * 2 * n + j]; a[c * 2 * n + j] = a[piv * 2 * n + j]; a[piv * 2 * n + j] = t; }
    double d = a[c * 2 * n + c];
    for (int j = 0; j < 2 * n; j++) a[c * 2 * n + j] /= d; /* d == 0 -> inf/nan like Eigen */
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      double f = a[r * 2 * n + c];
      if (f != 0.0 || isnan(f))
        for (int j = 0; j < 2 * n; j++) a[r * 2 * n + j] -= f * a[c * 2 * n + j];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) inv[i * n + j] = a[i * 2 * n + n + j];
  return 0;
}

int lo_icp_covariance(const double* Ap, double upper_bound, double* cov, double* cond) {
  const int n = 6;
  double inv[36];
  inv_n(Ap, n, inv);
  for (int i = 0; i < 36; i++) cov[i] = 0.05 * 0.05 * inv[i]; /* :487 */
  /* Eigen::LDLT<Lower>::compute, unblocked with diagonal pivoting; L = matrixL(), D = vectorD().
     The reference recomposes WITHOUT the permutation (:518) -- quirk kept. */
  double Mx[36];
  memcpy(Mx, cov, sizeof(Mx));
  for (int k = 0; k < n; k++) {
    int big = k;
    double bv = fabs(Mx[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (fabs(Mx[i * n + i]) > bv) { bv = fabs(Mx[i * n + i]); big = i; }
    if (big != k) {
      int s = n - big - 1;
      for (int j = 0; j < k; j++) { double t = Mx[k * n + j]; Mx[k * n + j] = Mx[big * n + j]; Mx[big * n + j] = t; }
      for (int i = 0; i < s; i++) { double t = Mx[(big + 1 + i) * n + k]; Mx[(big + 1 + i) * n + k] = Mx[(big + 1 + i) * n + big]; Mx[(big + 1 + i) * n + big] = t; }
      { double t = Mx[k * n + k]; Mx[k * n + k] = Mx[big * n + big]; Mx[big * n + big] = t; }
      for (int i = k + 1; i < big; i++) { double t = Mx[i * n + k]; Mx[i * n + k] = Mx[big * n + i]; Mx[big * n + i] = t; }
    }
    int rs = n - k - 1;
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; j++) temp[j] = Mx[j * n + j] * Mx[k * n + j];
      double s = 0;
      for (int j = 0; j < k; j++) s += Mx[k * n + j] * temp[j];
      Mx[k * n + k] -= s;
      for (int i = 0; i < rs; i++) {
        double t = 0;
        for (int j = 0; j < k; j++) t += Mx[(k + 1 + i) * n + j] * temp[j];
        Mx[(k + 1 + i) * n + k] -= t;
      }
    }
    double akk = Mx[k * n + k];
    int valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) break;
    if (rs > 0 && valid)
      for (int i = 0; i < rs; i++) Mx[(k + 1 + i) * n + k] /= akk;
  }
  double L[36], D[6];
  for (int i = 0; i < n; i++) {
    D[i] = Mx[i * n + i];
    for (int j = 0; j < n; j++) L[i * n + j] = (i == j) ? 1.0 : (i > j ? Mx[i * n + j] : 0.0);
  }
  const double lower_bound = 1e-12;
  for (int i = 0; i < n; i++)
    if (isnan(D[i])) { /* :499-503 */
      for (int a = 0; a < 36; a++) cov[a] = (a % 7 == 0) ? upper_bound : 0.0;
      return 0;
    }
  int recompute = 0;
  for (int i = 0; i < n; i++) {
    if (D[i] <= 0) { D[i] = lower_bound; recompute = 1; }
    if (D[i] > upper_bound) { D[i] = upper_bound; recompute = 1; }
  }
  if (recompute)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += L[i * n + k] * D[k] * L[j * n + k];
        cov[i * n + j] = s;
      }
  int has_nan = 0;
  for (int a = 0; a < 36; a++)
    if (isnan(cov[a])) has_nan = 1;
  if (has_nan)
    for (int a = 0; a < 36; a++) cov[a] = (a % 7 == 0) ? upper_bound : 0.0;
  /* condition number: singular values of a symmetric matrix = |eigenvalues| */
  double sym[36], ev[6], V[36];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) sym[i * n + j] = 0.5 * (cov[i * n + j] + cov[j * n + i]);
  lo_eig_sym(sym, n, ev, V);
  double smax = 0, smin = DBL_MAX;
  for (int i = 0; i < n; i++) {
    double s = fabs(ev[i]);
    if (s > smax) smax = s;
    if (s < smin) smin = s;
  }
  if (cond) *cond = smax / smin;
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * K1: pcl::VoxelGrid (custom_voxel_grid.cc:76-87 -> VoxelGrid<PCLPointCloud2>::applyFilter, PCL 1.10,
 * not vendored).  Restated: bbox over finite points passing the field limits; int index with
 * floor(p*inv_leaf) - min_b; sort by index; all-field float centroid; output in ascending voxel index.
 * std::sort is unstable in PCL so the intra-voxel summation order is "parity unpinned"; here: input order.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int idx; int pt; } vox_pair;
static int vox_cmp(const void* a, const void* b) {
  const vox_pair* x = (const vox_pair*)a;
  const vox_pair* y = (const vox_pair*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt);
}

/* nrm4 / out_nrm4 non-NULL: pcl::VoxelGrid<pcl::PointXYZINormal> (PointCloudFilter.cc:119-124).  PCL >= 1.8 fills a
 * pcl::CentroidPoint<PointT> per voxel (filters/impl/voxel_grid.hpp); its accumulators (common/impl/accumulators.hpp, not
 * vendored) are float sums in leaf order: AccumulatorXYZ, AccumulatorIntensity and AccumulatorCurvature return sum / n,
 * AccumulatorNormal returns the NORMALISED sum of the normal 4-vectors (normal_x, normal_y, normal_z, 0); a zero sum stays
 * zero (Eigen >= 3.3 normalized()). */
static int voxel_grid_impl(const float* xyzi, const float* nrm4, int n, float leaf, int limit_axis, double lo, double hi, float* out,
                           float* out_nrm4, int out_cap) {
  float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int any = 0;
  for (int i = 0; i < n; i++) {
    const float* p = xyzi + 4 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    if (limit_axis >= 0 && (p[limit_axis] > (float)hi || p[limit_axis] < (float)lo)) continue;
    for (int d = 0; d < 3; d++) {
      if (p[d] < mn[d]) mn[d] = p[d];
      if (p[d] > mx[d]) mx[d] = p[d];
    }
    any = 1;
  }
  if (!any) return 0;
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
  int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
  int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) return -1;
  int minb[3], maxb[3], divb[3], mul[3];
  for (int d = 0; d < 3; d++) {
    minb[d] = (int)floorf(mn[d] * inv);
    maxb[d] = (int)floorf(mx[d] * inv);
    divb[d] = maxb[d] - minb[d] + 1;
  }
  mul[0] = 1; mul[1] = divb[0]; mul[2] = divb[0] * divb[1];
  vox_pair* pairs = (vox_pair*)malloc(sizeof(vox_pair) * (size_t)n);
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    const float* p = xyzi + 4 * (size_t)i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    if (limit_axis >= 0 && (p[limit_axis] > (float)hi || p[limit_axis] < (float)lo)) continue;
    int ijk0 = (int)(floorf(p[0] * inv) - (float)minb[0]);
    int ijk1 = (int)(floorf(p[1] * inv) - (float)minb[1]);
    int ijk2 = (int)(floorf(p[2] * inv) - (float)minb[2]);
    pairs[cnt].idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
    pairs[cnt].pt = i;
    cnt++;
  }
  qsort(pairs, (size_t)cnt, sizeof(vox_pair), vox_cmp);
  int nout = 0;
  for (int s = 0; s < cnt;) {
    int e = s;
    float acc[4] = {0, 0, 0, 0}, na[4] = {0, 0, 0, 0};
    while (e < cnt && pairs[e].idx == pairs[s].idx) {
      const float* p = xyzi + 4 * (size_t)pairs[e].pt;
      acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; acc[3] += p[3];
      if (nrm4) {
        const float* q = nrm4 + 4 * (size_t)pairs[e].pt;
        na[0] += q[0]; na[1] += q[1]; na[2] += q[2]; na[3] += q[3];   /* [3] = curvature (its own accumulator) */
      }
      e++;
    }
    float c = (float)(e - s);
    if (nout < out_cap) {
      out[4 * (size_t)nout + 0] = acc[0] / c;
      out[4 * (size_t)nout + 1] = acc[1] / c;
      out[4 * (size_t)nout + 2] = acc[2] / c;
      out[4 * (size_t)nout + 3] = acc[3] / c;
      if (nrm4 && out_nrm4) {
        float z = (na[0] * na[0] + na[1] * na[1]) + na[2] * na[2];
        if (z > 0.0f) { float l = sqrtf(z); na[0] = na[0] / l; na[1] = na[1] / l; na[2] = na[2] / l; }
        out_nrm4[4 * (size_t)nout + 0] = na[0];
        out_nrm4[4 * (size_t)nout + 1] = na[1];
        out_nrm4[4 * (size_t)nout + 2] = na[2];
        out_nrm4[4 * (size_t)nout + 3] = na[3] / c;
      }
    }
    nout++;
    s = e;
  }
  free(pairs);
  return nout;
}

int lo_voxel_grid(const float* xyzi, int n, float leaf, int limit_axis, double lo, double hi, float* out, int out_cap) {
  return voxel_grid_impl(xyzi, NULL, n, leaf, limit_axis, lo, hi, out, NULL, out_cap);
}
int lo_voxel_grid_pointf(const float* xyzi, const float* nrm4, int n, float leaf, float* out, float* out_nrm4, int out_cap) {
  return voxel_grid_impl(xyzi, nrm4, n, leaf, -1, 0.0, 0.0, out, out_nrm4, out_cap);
}

/* ------------------------------------------------------------------------------------------
 * K3 (filter flavour): pcl::NormalEstimationOMP k-NN (normal_computation.cc:26-59; PCL 1.10
 * features/normal_3d.h + common/centroid.hpp + common/eigen.hpp, not vendored).  Restated:
 * float 9-accumulator mean/covariance over the k neighbours (ascending distance), pcl::eigen33
 * closed-form smallest eigenpair in float, curvature = |lambda0 / trace|, flip towards viewpoint (0,0,0).
 * ------------------------------------------------------------------------------------------ */
static void roots2f(float b, float c, float* r) {
  r[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  float sd = sqrtf(d);
  r[2] = 0.5f * (b + sd);
  r[1] = 0.5f * (b - sd);
}
static void roots3f(const float* m /*row-major sym*/, float* r) {
  float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
  float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  float c2 = m00 + m11 + m22;
  if (fabsf(c0) < FLT_EPSILON) {
    roots2f(c2, c1, r);
    return;
  }
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  float rho = sqrtf(-a_over_3);
  float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
  float ct = cosf(theta), st = sinf(theta);
  r[0] = c2_over_3 + 2.0f * rho * ct;
  r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
  r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
  float t;
  if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  if (r[1] >= r[2]) {
    t = r[1]; r[1] = r[2]; r[2] = t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
  }
  if (r[0] <= 0.0f) roots2f(c2, c1, r);
}
static void cross3f(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void eigen33_smallest(const float* mat, float* eval, float* evec) {
  float scale = 0.0f;
  for (int i = 0; i < 9; i++)
    if (fabsf(mat[i]) > scale) scale = fabsf(mat[i]);
  if (scale <= FLT_MIN) scale = 1.0f;
  float sm[9], r[3];
  for (int i = 0; i < 9; i++) sm[i] = mat[i] / scale;
  roots3f(sm, r);
  *eval = r[0] * scale;
  sm[0] -= r[0]; sm[4] -= r[0]; sm[8] -= r[0];
  float v1[3], v2[3], v3[3];
  cross3f(sm + 0, sm + 3, v1);
  cross3f(sm + 0, sm + 6, v2);
  cross3f(sm + 3, sm + 6, v3);
  float l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2];
  float l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2];
  float l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
  const float* v;
  float l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  float s = sqrtf(l);
  evec[0] = v[0] / s; evec[1] = v[1] / s; evec[2] = v[2] / s;
}

/* NormalEstimation::computeFeature body for one point, given its neighbour list in search order
 * (pcl/features/normal_3d.h computePointNormal -> computeMeanAndCovarianceMatrix (float accumulators) ->
 * solvePlaneParameters (eigen33) -> flipNormalTowardsViewpoint, vp = 0) */
static void normal_from_neighbours(const float* xyz4, const int* idx, int cnt, const float* q, float* o) {
  if (cnt < 3) { o[0] = o[1] = o[2] = o[3] = NAN; return; }
  float acc[9] = {0};
  for (int j = 0; j < cnt; j++) {
    const float* p = xyz4 + 4 * (size_t)idx[j];
    acc[0] += p[0] * p[0]; acc[1] += p[0] * p[1]; acc[2] += p[0] * p[2];
    acc[3] += p[1] * p[1]; acc[4] += p[1] * p[2]; acc[5] += p[2] * p[2];
    acc[6] += p[0]; acc[7] += p[1]; acc[8] += p[2];
  }
  float c = (float)cnt;
  for (int a = 0; a < 9; a++) acc[a] /= c;
  float cov[9];
  cov[0] = acc[0] - acc[6] * acc[6];
  cov[1] = acc[1] - acc[6] * acc[7];
  cov[2] = acc[2] - acc[6] * acc[8];
  cov[4] = acc[3] - acc[7] * acc[7];
  cov[5] = acc[4] - acc[7] * acc[8];
  cov[8] = acc[5] - acc[8] * acc[8];
  cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
  float ev, nv[3];
  eigen33_smallest(cov, &ev, nv);
  float eig_sum = cov[0] + cov[4] + cov[8];
  float curv = (eig_sum != 0.0f) ? fabsf(ev / eig_sum) : 0.0f;
  /* flipNormalTowardsViewpoint, vp = (0,0,0) */
  float vx = 0.0f - q[0], vy = 0.0f - q[1], vz = 0.0f - q[2];
  float cos_theta = (vx * nv[0] + vy * nv[1]) + vz * nv[2];
  if (cos_theta < 0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
  o[0] = nv[0]; o[1] = nv[1]; o[2] = nv[2]; o[3] = curv;
}

void lo_normals_knn(const float* xyz4, int n, const lo_tree* t, int k, float* out, int threads) {
  if (threads < 1) threads = 1;
  if (k > 64) k = 64;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) if (threads > 1)
  for (int i = 0; i < n; i++) {
    int idx[64];
    float dd[64];
    knn_list L = {dd, idx, k, 0};
    const float* q = xyz4 + 4 * (size_t)i;
    float* o = out + 4 * (size_t)i;
    if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) { o[0] = o[1] = o[2] = o[3] = NAN; continue; }
    knn_rec(t, 0, q, &L);
    normal_from_neighbours(xyz4, idx, L.cnt, q, o);
  }
}

/* radius search (normal_computation.cc:71-74 -> setRadiusSearch; FLANN RadiusResultSet keeps dist < r^2, PCL returns the
 * neighbours sorted by distance); ties ordered by index here ("parity unpinned", see the NN index note) */
typedef struct { float d; int i; } rad_hit;
typedef struct { rad_hit* h; int cnt, cap; } rad_list;
static void radius_rec(const lo_tree* t, int node, const float* q, float r2, rad_list* L) {
  const lo_node* nd = &t->nodes[node];
  if (boxd2(q, nd->lo, nd->hi) >= r2) return;
  if (nd->left < 0) {
    for (int i = nd->begin; i < nd->end; i++) {
      float d = d2f(q, t->pts + 3 * i);
      if (d < r2) {
        if (L->cnt == L->cap) { L->cap = L->cap ? 2 * L->cap : 64; L->h = (rad_hit*)realloc(L->h, sizeof(rad_hit) * L->cap); }
        L->h[L->cnt].d = d; L->h[L->cnt].i = t->idx[i]; L->cnt++;
      }
    }
    return;
  }
  radius_rec(t, nd->left, q, r2, L);
  radius_rec(t, nd->right, q, r2, L);
}
static int rad_cmp(const void* a, const void* b) {
  const rad_hit *x = (const rad_hit*)a, *y = (const rad_hit*)b;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  return (x->i > y->i) - (x->i < y->i);
}
/* all points with d2 < r2 around q (3 floats), ascending (d2, index); returns the count (at most cap are written) */
int lo_radius_search(const lo_tree* t, const float* q, float r2, int32_t* idx, float* d2, int cap) {
  rad_list L = {NULL, 0, 0};
  if (t->n > 0) radius_rec(t, 0, q, r2, &L);
  qsort(L.h, L.cnt, sizeof(rad_hit), rad_cmp);
  int k = L.cnt < cap ? L.cnt : cap;
  for (int j = 0; j < k; j++) { idx[j] = L.h[j].i; if (d2) d2[j] = L.h[j].d; }
  int cnt = L.cnt;
  free(L.h);
  return cnt;
}
void lo_normals_radius(const float* xyz4, int n, const lo_tree* t, float radius, float* out, int threads) {
  if (threads < 1) threads = 1;
  float r2 = radius * radius;
#pragma omp parallel num_threads(threads) if (threads > 1)
  {
    rad_list L = {NULL, 0, 0};
    int* idx = NULL; int idx_cap = 0;
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
      const float* q = xyz4 + 4 * (size_t)i;
      float* o = out + 4 * (size_t)i;
      if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) { o[0] = o[1] = o[2] = o[3] = NAN; continue; }
      L.cnt = 0;
      if (t->n > 0) radius_rec(t, 0, q, r2, &L);
      qsort(L.h, L.cnt, sizeof(rad_hit), rad_cmp);
      if (L.cnt > idx_cap) { idx_cap = 2 * L.cnt; idx = (int*)realloc(idx, sizeof(int) * idx_cap); }
      for (int j = 0; j < L.cnt; j++) idx[j] = L.h[j].i;
      normal_from_neighbours(xyz4, idx, L.cnt, q, o);
    }
    free(L.h); free(idx);
  }
}

/* PCD v0.7 reader for the reference fixtures (FIELDS x y z intensity, DATA binary|ascii) */
int lo_read_pcd_xyzi(const char* path, float* out, int cap) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  char line[512];
  int points = -1, binary = -1, nfields = 0;
  while (fgets(line, sizeof(line), f)) {
    if (!strncmp(line, "FIELDS", 6)) {
      nfields = 0;
      for (char* p = line + 6; *p; p++)
        if (*p != ' ' && *p != '\n' && (p[-1] == ' ')) nfields++;
    }
    if (!strncmp(line, "POINTS", 6)) points = atoi(line + 7);
    if (!strncmp(line, "DATA", 4)) {
      binary = strstr(line, "binary") != NULL;
      break;
    }
  }
  if (points < 0 || binary < 0 || nfields != 4 || points > cap) { fclose(f); return -2; }
  if (binary) {
    if (fread(out, sizeof(float) * 4, (size_t)points, f) != (size_t)points) { fclose(f); return -3; }
  } else {
    for (int i = 0; i < points; i++)
      if (fscanf(f, "%f %f %f %f", out + 4 * i, out + 4 * i + 1, out + 4 * i + 2, out + 4 * i + 3) != 4) { fclose(f); return -3; }
  }
  fclose(f);
  return points;
}
