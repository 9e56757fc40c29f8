/*
 * locus_oracle.h -- CPU restatement ("oracle") of the LOCUS GICP registration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline.  The product path (locus_amd/csrc)
 * never links, imports or calls it.
 *
 * Parity status: the reference (PCL + FLANN + Eigen + common_nebula_slam) cannot be built
 * in this image, so this restatement is pinned only by the known-answer tests the
 * reference's own test-suite holds: hollow-cube translation KAT, plane Ap
 * KAT (56.7753 / 100), clamped-covariance KATs, 3x3 eigen KAT and the two garage PCD
 * fixtures (inputs only; thread-count invariance).  BFGS step-level parity with pcl::BFGS,
 * FLANN tie order, pcl::VoxelGrid intra-voxel summation order and
 * CalculateCovarianceFromNormals (un-vendored) are "parity unpinned".
 *
 * Layout conventions: clouds are float32 [n][4] (x, y, z, 1) plus optional float32 [n][4]
 * normals (nx, ny, nz, curvature).  4x4 transforms are column-major float (Eigen::Matrix4f
 * memory order).  Covariances / Mahalanobis matrices are row-major double [n][9].
 */
#ifndef LOCUS_ORACLE_H_
#define LOCUS_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lo_tree lo_tree;

typedef struct {
  int max_iterations;            /* gicp.h:129  (20 in LOCUS yaml) */
  int max_inner_iterations;      /* gicp.h:121  (20 odom / 50 loc) */
  double corr_dist;              /* corr_dist_threshold_ */
  double transformation_epsilon; /* gicp.h:130 */
  double rotation_epsilon;       /* gicp.h:119 */
  double gicp_epsilon;           /* gicp.h:118 */
  int k_correspondences;         /* gicp.h:112 */
  int recompute_source_cov;      /* gicp.h:115 */
  int recompute_target_cov;      /* gicp.h:116 */
  int num_threads;               /* gicp.h:134-141; OMP on the two loops the reference parallelises */
  int parallel_cost;             /* 0 = serial cost functor like the reference (gicp.hpp:291-402);
                                    1 = "fully parallel CPU" variant (OMP reduction), reported separately */
} lo_params;

#define LO_MAX_TRACE 256
typedef struct {
  int n_iters;
  float T[LO_MAX_TRACE][16];     /* transformation_ after each outer iteration (column-major) */
  int n_corr[LO_MAX_TRACE];
  int n_passes[LO_MAX_TRACE];    /* cost-functor passes over the correspondences in that iteration */
  int n_inner[LO_MAX_TRACE];
  double f_end[LO_MAX_TRACE];    /* cost at the end of the BFGS solve */
  double delta[LO_MAX_TRACE];
} lo_trace;

typedef struct {
  float T[16];                   /* final_transformation_ (gicp.hpp:583), column-major */
  int converged;
  int iterations;
  int n_corr_last;
  int status;                    /* 0 ok; <0 see LO_E* */
  long total_passes;             /* total cost-functor passes */
  double t_index, t_cov, t_nn, t_opt, t_total; /* seconds */
} lo_result;

enum { LO_OK = 0, LO_EINVAL = -1, LO_ENOMEM = -2, LO_ETOO_FEW = -4, LO_ESOLVER = -5, LO_ENO_NN = -6 };

void lo_default_params(lo_params* p);
void lo_set_cost_variant(int v); /* analysis only: 1 = FMA-contracted float T*p in the cost functor */
void lo_set_bfgs_variant(int quad_curv_gt_a); /* 1 = pcl::BFGS's reported `c > a` curvature test of the quadratic interpolation (GSL: `c > 0`) */

/* exact nearest-neighbour index (stands in for pcl::search::KdTree -> FLANN KDTreeSingleIndex);
   distances are float ((dx*dx+dy*dy)+dz*dz), ties -> lowest original index. */
lo_tree* lo_tree_build(const float* xyz4, int n);
void lo_tree_free(lo_tree* t);
void lo_nn1(const lo_tree* t, const float* q4, int nq, int32_t* idx, float* d2, int threads);
void lo_nn1_brute(const float* xyz4, int n, const float* q4, int nq, int32_t* idx, float* d2);
void lo_knn(const lo_tree* t, const float* q4, int nq, int k, int32_t* idx, float* d2, int threads);
void lo_knn_brute(const float* xyz4, int n, const float* q4, int nq, int k, int32_t* idx, float* d2);

/* K6: y = T*x float (pcl::transformPointCloud); with_normals also rotates normals */
void lo_transform(const float* xyz4, const float* nrm4, int n, const float* T16, float* out_xyz4, float* out_nrm4);

/* K3': C = I - (1-eps) n n^T (CalculateCovarianceFromNormals restated, gicp.hpp:81-82) */
void lo_cov_from_normals(const float* nrm4, int n, double eps, double* cov9);
/* K3: k-NN covariance + SVD regularisation (gicp.hpp:85-154) */
int lo_cov_knn(const float* xyz4, int n, const lo_tree* t, int k, double eps, double* cov9, int threads);

/* K4: one NN + Mahalanobis sweep (gicp.hpp:464-498). T16 = transformation_ (float, col-major),
   R9 = row-major double 3x3 of transformation_*guess.  out: tgt_idx[n] (-1 = unmatched), maha9[n][9]. */
int lo_nn_mahalanobis(const float* out_xyz4, int n, const lo_tree* tgt_tree, const double* cov_src9,
                       const double* cov_tgt9, const float* T16, const double* R9, double corr_dist,
                       int32_t* tgt_idx, double* maha9, int threads);

/* K5: the cost functor fdf (gicp.hpp:362-402) on compacted correspondences.
   x = (tx,ty,tz, roll,pitch,yaw); out f, g[6]; also raw sums[13] (f, g_t[3], R[9]) before /m. */
void lo_cost_fdf(const float* out_xyz4, const float* tgt_xyz4, const int32_t* src_idx, const int32_t* tgt_idx,
                 int m, const double* maha9, const double* x6, double* f, double* g6, double* sums13);
/* estimateRigidTransformationBFGS (gicp.hpp:218-287) on given correspondences; T16 in/out */
int lo_estimate_rigid_bfgs(const float* out_xyz4, const float* tgt_xyz4, const int32_t* src_idx, const int32_t* tgt_idx, int m,
                           const double* maha9, int max_inner, float* T16, int* n_inner, double* f_end, int* passes);
/* ... and the same solve from state x_start6 with its inner steps recorded (row 0 = after minimizeInit, then one row per successful
 * minimizeOneStep): xs [cap][6], fs, gnorms, evals (functor evaluations so far).  Returns the rows written; *result = the last step's status
 * (0 success: gradient below 1e-2, 1 no progress, -1 still running when max_inner was reached). */
int lo_estimate_rigid_bfgs_trace(const float* out_xyz4, const float* tgt_xyz4, const int32_t* src_idx, const int32_t* tgt_idx, int m,
                                 const double* maha9, int max_inner, const double* x_start6, double* xs, double* fs, double* gnorms, int* evals,
                                 int cap, int* result);
/* applyState (gicp.hpp:619-634): T = [Rz(x5)Ry(x4)Rx(x3) | x0..2] in float, column-major out */
void lo_apply_state(const double* x6, float* T16);

/* a3+a6: pcl::Registration::align + computeTransformation (gicp.hpp:406-617) */
int lo_gicp_align(const float* src_xyz4, const float* src_nrm4, int n, const float* tgt_xyz4,
                  const float* tgt_nrm4, int m, const lo_params* P, const float* guess16, lo_result* res,
                  lo_trace* trace, float* aligned_xyz4);

/* K7: pcl::Registration::getFitnessScore (max_range = DBL_MAX) */
double lo_fitness(const float* src_xyz4, int n, const float* T16, const lo_tree* tgt_tree, int threads);

/* K8: normalizePCloud (utils.cc:106-128) + ComputeAp_ForPoint2PlaneICP (PointCloudLocalization.cc:723-750).
   corr[i] indexes ref_nrm4; Ap row-major 6x6. */
void lo_normalize_cloud(const float* xyz4, int n, float* out_xyz4);
void lo_p2plane_Ap(const float* qnorm_xyz4, int n, const float* ref_nrm4, const int64_t* corr, double* Ap36);
/* H2: ComputePoint2PlaneICPCovariance conditioning (PointCloudLocalization.cc:487-538).
   returns 1 ok / 0 failure (NaN in D); cov row-major 6x6, cond = condition number. */
int lo_icp_covariance(const double* Ap36, double upper_bound, double* cov36, double* cond);
/* doEigenDecomp3x3 / 6x6 (utils.cc:130-154): ascending eigenvalues, eigenvectors in columns (row-major out) */
void lo_eig_sym(const double* A, int n, double* evals, double* evecs);

/* K1: pcl::VoxelGrid semantics (custom_voxel_grid.cc:76-87). in/out float [n][4] = x,y,z,intensity.
   limit_axis: -1 none, 0/1/2 = x/y/z pass-through [lo,hi].  returns out count, or -1 on index overflow
   (PCL then copies input to output). */
int lo_voxel_grid(const float* xyzi, int n, float leaf, int limit_axis, double lo, double hi, float* out_xyzi,
                  int out_cap);
/* the PointXYZINormal flavour, pcl::VoxelGrid<PointF> (PointCloudFilter.cc:119-124): nrm4 = nx,ny,nz,curvature per point;
   same voxels and order, every field averaged (pcl::CentroidPoint accumulators), normals re-normalised */
int lo_voxel_grid_pointf(const float* xyzi, const float* nrm4, int n, float leaf, float* out_xyzi, float* out_nrm4, int out_cap);
/* K3 (filter flavour): pcl::NormalEstimationOMP k-NN (normal_computation.cc:26-59), viewpoint (0,0,0).
   out_nrm4 = nx,ny,nz,curvature. */
void lo_normals_knn(const float* xyz4, int n, const lo_tree* t, int k, float* out_nrm4, int threads);
/* radius flavour (normal_computation.cc:71-74): all neighbours with d2 < radius^2, sorted by distance; < 3 neighbours -> NaN
   (the nodelet then drops those points, normal_computation.cc:52-56) */
void lo_normals_radius(const float* xyz4, int n, const lo_tree* t, float radius, float* out_nrm4, int threads);

/* ---- NDT (registration_method: ndt), restated in locus_oracle_ndt.c -- "parity unpinned" (no reference test) ---- */
typedef struct {
  float resolution;              /* voxel edge, ndt_omp_impl.hpp:50 (1.0) */
  double step_size;              /* More-Thuente maximum step, :51 (0.1) */
  double outlier_ratio;          /* :52 (0.55) */
  double transformation_epsilon; /* :93 (0.1); LOCUS passes icp_tf_epsilon */
  int max_iterations;            /* :94 (35); LOCUS passes icp_iterations */
  int min_points_per_voxel;      /* voxel_grid_covariance_omp.h:186 (6) */
  double min_covar_eigvalue_mult;/* :187 (0.01) */
  int num_threads;
} lo_ndt_params;
typedef struct {
  float T[16];                   /* final_transformation_, column-major */
  int converged, iterations, evaluations, n_cells;
  double trans_probability;      /* score / n (ndt_omp_impl.hpp:211) */
} lo_ndt_result;
typedef struct lo_ndt_grid lo_ndt_grid;
void lo_ndt_default_params(lo_ndt_params* p);
lo_ndt_grid* lo_ndt_grid_build(const float* tgt_xyz4, int m, const lo_ndt_params* P);
void lo_ndt_grid_free(lo_ndt_grid* g);
/* copies up to cap cells (ascending voxel index = kd-tree order); returns the cell count */
int lo_ndt_grid_cells(const lo_ndt_grid* g, double* mean3, double* icov9, float* centroid4, int cap);
void lo_ndt_pose_to_matrix(const double* p6, float* T16);
void lo_ndt_matrix_to_pose(const float* T16, double* p6);
/* computeDerivatives (float point derivatives): returns the score; hess36 row-major, zero when want_h == 0 */
double lo_ndt_derivatives(const lo_ndt_grid* g, const lo_ndt_params* P, const float* src_xyz4, const float* trans_xyz4, int n, const double* p6,
                          double* grad6, double* hess36, int want_h);
/* computeHessian (double path) */
void lo_ndt_hessian(const lo_ndt_grid* g, const lo_ndt_params* P, const float* src_xyz4, const float* trans_xyz4, int n, const double* p6,
                    double* hess36);
void lo_svd_solve6(const double* A36, const double* b6, double* x6);
int lo_ndt_align(const float* src_xyz4, int n, const float* tgt_xyz4, int m, const lo_ndt_params* P, const float* guess16, lo_ndt_result* out);

/* radius search on the exact index: points with d2 < r2, ascending (d2, index); returns the count (<= cap written) */
int lo_radius_search(const lo_tree* t, const float* q3, float r2, int32_t* idx, float* d2, int cap);

/* PCD v0.7 binary/ascii reader for the reference's own fixtures (x y z intensity). returns n or <0 */
int lo_read_pcd_xyzi(const char* path, float* out_xyzi, int cap);

#ifdef __cplusplus
}
#endif
#endif
