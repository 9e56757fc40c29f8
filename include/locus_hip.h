/*
 * locus_hip.h -- C ABI of the MI355X-native GICP registration hot path for LOCUS.
 *
 * This is the drop-in boundary: thin C++ adapters that keep the reference's class surfaces
 * (pcl::Registration<PointF,PointF> selected by `registration_method`, the pcl_ros::Filter nodelets,
 * PointCloudOdometry / PointCloudLocalization) call these entry points; see INTEGRATION.md for the
 * adapter a LOCUS maintainer would add.  Plain pointers and sizes only; no C++/torch types.
 * All `file:line` citations are relative to the LOCUS repository.
 *
 * Conventions
 *   - 4x4 transforms are 16 floats, COLUMN-major (= Eigen::Matrix4f memory order).
 *   - host clouds are described by lh_cloud_view (base, count, stride, field offsets) so both
 *     pcl::PointXYZI (32 B) and pcl::PointXYZINormal (48 B = PointF) arrays can be passed as is.
 *   - error convention: 0 = OK, < 0 = LH_E*; no C++ exception crosses the boundary.
 *   - a handle is NOT re-entrant (one lidar callback at a time, Locus.cc:64-68); distinct handles may be
 *     used from distinct threads.
 *   - every entry point requires a HIP device; there is no CPU fallback (LH_EDEVICE when none).
 */
#ifndef LOCUS_HIP_H_
#define LOCUS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_ABI_VERSION 1

typedef int lh_status;
enum {
  LH_OK = 0,
  LH_EINVAL = -1,       /* bad argument / empty cloud / k_correspondences > cloud size (gicp.hpp:72-79) */
  LH_ENOMEM = -2,
  LH_EDEVICE = -3,      /* HIP error or no device */
  LH_ETOO_FEW_CORR = -4,/* < 4 correspondences: NotEnoughPointsException (gicp.hpp:225-232) */
  LH_ESOLVER = -5,      /* SolverDidntConvergeException (gicp.hpp:280-285) */
  LH_ENO_NN = -6        /* no nearest neighbour found (gicp.hpp:471-478) */
};

typedef struct lh_ctx lh_ctx;     /* one per GPU / process rank */
typedef struct lh_cloud lh_cloud; /* device-resident cloud (+ lazily built NN index) */
typedef struct lh_gicp lh_gicp;   /* one registration object = one `icp_` member (PointCloudOdometry.h:154) */

/* caller-owned, read-only host memory describing a pcl::PointCloud<PointT>::points array */
typedef struct {
  const void* base;
  uint32_t count;
  uint32_t stride;        /* bytes per point: 32 (PointXYZI) or 48 (PointXYZINormal) or 16 (xyz1); the library reads (count - 1) * stride +
                             the end of the last used field bytes from base, in ONE copy: a stride far larger than the fields costs PCIe time */
  uint32_t off_xyz;       /* byte offset of float x,y,z */
  uint32_t off_normal;    /* byte offset of float normal_x,y,z ; UINT32_MAX if the type has none */
  uint32_t off_intensity; /* byte offset of float intensity   ; UINT32_MAX if none */
  uint32_t off_curvature; /* byte offset of float curvature   ; UINT32_MAX if none */
} lh_cloud_view;

/* knobs of MultithreadedGeneralizedIterativeClosestPoint (gicp.h:111-132 defaults in comments) */
typedef struct {
  int max_iterations;            /* 200  ; LOCUS yaml 20 (point_cloud_odometry/config/parameters.yaml:12) */
  int max_inner_iterations;      /* 20   ; localization 50 (PointCloudLocalization.cc:238) */
  double corr_dist;              /* 5.0  ; 1.0 odom / 0.2 loc */
  double transformation_epsilon; /* 5e-4 ; 1e-3 / 1e-5 */
  double rotation_epsilon;       /* 2e-3 */
  double gicp_epsilon;           /* 1e-3 */
  int k_correspondences;         /* 20 */
  int recompute_source_cov;      /* 0 = covariances from stored normals (production), 1 = k-NN + SVD */
  int recompute_target_cov;
  int num_threads;               /* accepted for surface compatibility (setNumThreads, gicp.h:134-141); ignored */
  int enable_timing;             /* enableTimingOutput (gicp.h:143): collect per-kernel HIP-event times */
  int cost_mode;                 /* how OptimizationFunctorWithIndices (gicp.hpp:291-402) is evaluated on the device:
                                    0 = one pass over the correspondences per evaluation, reference arithmetic (float T*p);
                                    1 = (default) second-order moments: the sums are exactly quadratic in the 12 entries of T,
                                        so ONE 74-moment reduction per outer iteration serves every BFGS evaluation (double
                                        T*p instead of float: differs from mode 0 by the reference's own float rounding noise).
                                    Tolerances against the reference-algorithm CPU restatement, AS TESTED (tests/test_gpu_align.py, every bench.py
                                    run; distributions over 64 full-size pairs in profiles/r04_parity_distributions.json; in brackets the
                                    distance between the restatement's own two legal builds -- float T*p with / without FMA contraction):
                                      20 forced iterations on 100k-point scans
                                        mode 0: |dt| median 0 (bit-identical trace on small clouds), p90 <= 1e-4 m, every pair <= 2.5e-4 m
                                                (only the order in which block sums of f are added differs; it can flip one line-search
                                                comparison in ~575 evaluations); |dR| <= 1e-4; fitness rel median 0, every pair <= 1e-3
                                        mode 1: |dt| median <= 1e-4 m [7.4e-5], p90 <= 2.5e-4 m [2.4e-4], a pair beyond that only where the
                                                reference's two builds part by more on that pair, never beyond 5e-3 [max 3.3e-3];
                                                |dR| <= 1.3e-4; fitness rel median <= 1e-4 [8.2e-5], p90 <= 5e-4, every pair <= 2e-3;
                                                largest per-iteration |dT| of a pair: median <= 1.5e-3, p90 <= 1e-2 [4e-4 / 7e-3]
                                      production stopping rule (tf_eps 1e-3, rotation_epsilon 2e-3: the result is defined to that scale)
                                        mode 0: |dt| median 0, p90 <= 3e-4 m, every pair <= 1e-3 m; iteration count equal on every pair
                                        mode 1: |dt| median <= 5e-4 m [2.1e-4], p90 <= 3e-3 m [3.2e-3], every pair <= 2e-2 m [2.1e-2];
                                                |dR| <= 5e-4; fitness rel median <= 5e-4, p90 <= 3e-3; iteration count within 3 */
  int solver;                    /* where the loop between two sweeps runs in cost_mode 1 (the BFGS solve on the 74 moments, the convergence
                                    test): 2 = on the device (k_solve: the host only enqueues iterations and looks at the pairs' states
                                    every few rounds); 1 = on the host, one sync per outer iteration (the path the source-sharded pair
                                    takes anyway, lh_set_allreduce); 0 = (default) device for batches of >= 8 pairs in flight, host for
                                    fewer (lower latency for one pair at a time).  Same code, same arithmetic (lh_math.hpp): the
                                    results are bit-identical either way. */
  int bfgs_quad_curv;            /* pcl::BFGS is un-vendored in the reference tree (gicp.hpp:249-271 calls it); its line search is restated from the GSL
                                    algorithm it was ported from.  One reported deviation of the port moves results measurably (1.7-2.5e-4 m on the
                                    reference's own fixtures, tests/test_second_restatement.py): the quadratic interpolation accepts its stationary
                                    point if `c > a` instead of GSL's `c > 0`.  0 = (default) GSL's test; 1 = the reported PCL reading -- for a
                                    maintainer who can compare with a real PCL build.  The CPU oracle has the same switch (lo_set_bfgs_variant). */
} lh_gicp_params;

typedef struct {
  float T[16];                   /* final_transformation_ (gicp.hpp:583) */
  int converged;                 /* hasConverged() */
  int iterations;                /* nr_iterations_ */
  int n_correspondences_last;
  int status;                    /* LH_OK, or the exception the reference would have caught (gicp.hpp:542-547) */
  double fitness;                /* getFitnessScore(); NaN unless requested via lh_gicp_fitness */
  int cost_passes;               /* device passes over the correspondences (fused f+df evaluations) */
  int reserved;
} lh_gicp_result;

/* per-outer-iteration trace (parity tests compare it with the oracle's) */
#define LH_MAX_TRACE 256
typedef struct {
  int n_iters;
  float T[LH_MAX_TRACE][16];
  int n_corr[LH_MAX_TRACE];
  int n_passes[LH_MAX_TRACE];
  int n_inner[LH_MAX_TRACE];
  double f_end[LH_MAX_TRACE];
  double delta[LH_MAX_TRACE];
} lh_gicp_trace;

/* ---- context -------------------------------------------------------------------------------- */
int lh_abi_version(void);
const char* lh_status_string(lh_status s);
/* One context per GPU per process is the supported configuration (the host classes share it, locus_amd/host): calls on a
 * context are issued in order on its stream, and temporary device buffers are recycled in that order.  Two contexts on
 * the SAME device must not have asynchronous work in flight at the same time (call lh_synchronize between them).
 * HIP runtime setting: the batch entry points keep up to sixteen groups of pairs in flight on as many HIP streams and count on their
 * kernels overlapping; the runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable
 * at the process's FIRST HIP call (9.7 k -> 12.2 k scan-pairs/s on the 512-pair bench queue with 24).  The library does NOT touch the
 * environment on its own (no equivalent of the reference's omp_set_num_threads side effect, gicp.h:138): see lh_runtime_init /
 * lh_runtime_info below and INTEGRATION.md section 5. */
/* Optional, explicit: GPU_MAX_HW_QUEUES = max_hw_queues (0: the library's recommendation, 24) unless the deployment already set the variable.
 * Only effective BEFORE the process's first HIP call; call it from the main thread before other threads exist (it is setenv).  Returns
 * LH_EINVAL for a value outside 0..128. */
lh_status lh_runtime_init(int max_hw_queues);
typedef struct {
  int hw_queues_env;          /* GPU_MAX_HW_QUEUES as this process sees it now (-1: unset) */
  int streams_probed;         /* 16 */
  double stream_concurrency;  /* MEASURED: how many of sixteen one-wave kernels on the context's sixteen streams were resident at the same instant (16 = all; ~4 = the runtime's default) */
  int adequate;               /* stream_concurrency >= 12: batches of >= 64 pairs get the overlap the scheduler counts on */
  int reserved;
} lh_runtime_info_t;
/* What the HIP runtime actually gives this process (an overlap probe on the context's own sixteen scheduler streams -- created if the scheduler has
 * not needed them yet -- ~2 ms, measured once per device and cached).  The scheduler runs
 * the same probe the first time a batch spreads over more than four streams and reports a shortfall once on stderr. */
lh_status lh_runtime_info(lh_ctx* ctx, lh_runtime_info_t* out);
lh_status lh_create(lh_ctx** out, int device_id);
void lh_destroy(lh_ctx* ctx);
lh_status lh_synchronize(lh_ctx* ctx);
void lh_default_gicp_params(lh_gicp_params* p);          /* gicp.h:111-132 */

/* ---- device-resident clouds ------------------------------------------------------------------ */
/* copies xyz (+normals, intensity when present) to HBM.  Replaces holding a PointCloudF::Ptr
   (PointCloudOdometry.h:125-131). */
lh_status lh_cloud_create(lh_ctx* ctx, const lh_cloud_view* view, lh_cloud** out);
void lh_cloud_destroy(lh_cloud* c);
uint32_t lh_cloud_size(const lh_cloud* c);
/* build / drop the NN index = tree_->setInputCloud() inside pcl::Registration::initCompute (K2) */
lh_status lh_cloud_build_index(lh_cloud* c);
lh_status lh_cloud_drop_index(lh_cloud* c);
/* download into a caller array of `stride`-byte points (writes xyz, normals, intensity where offsets given) */
lh_status lh_cloud_download(const lh_cloud* c, void* out_base, uint32_t stride, uint32_t off_xyz,
                            uint32_t off_normal, uint32_t off_intensity, uint32_t off_curvature);
/* y = T*x (pcl::transformPointCloud, gicp.hpp:440,586; PointCloudOdometry.cc:255,260) ;
   with_normals != 0 also rotates normals (transformPointCloudWithNormals, PointCloudLocalization.cc:197,218,325).
   out may alias in. */
lh_status lh_cloud_transform(const lh_cloud* in, const float T[16], int with_normals, lh_cloud** out);
/* points [first, first+count) as a new cloud, device to device: a rank's source shard (lh_set_allreduce below) */
lh_status lh_cloud_slice(const lh_cloud* in, uint32_t first, uint32_t count, lh_cloud** out);
/* multi-lidar merge (PointCloudMerger.cc:158-159, `*merged = *a + *b`): the inputs' points in order, device to device;
   normals / intensity are kept only if every input has them */
lh_status lh_cloud_concat(lh_cloud* const* parts, int n_parts, lh_cloud** out);

/* ---- registration object (MultithreadedGeneralizedIterativeClosestPoint) ---------------------- */
lh_status lh_gicp_create(lh_ctx* ctx, const lh_gicp_params* p, lh_gicp** out);
void lh_gicp_destroy(lh_gicp* g);
lh_status lh_gicp_set_params(lh_gicp* g, const lh_gicp_params* p);
/* setInputSource (gicp.h:162-179): copies to device, invalidates source covariances */
lh_status lh_gicp_set_source(lh_gicp* g, const lh_cloud_view* v);
/* setInputTarget (gicp.h:196-200): copies to device, invalidates target covariances + index */
lh_status lh_gicp_set_target(lh_gicp* g, const lh_cloud_view* v);
/* device-resident variants (the handle borrows the cloud; caller keeps ownership) */
lh_status lh_gicp_set_source_cloud(lh_gicp* g, lh_cloud* c);
lh_status lh_gicp_set_target_cloud(lh_gicp* g, lh_cloud* c);
/* odometry fast path for copyPointCloud(*query_, *reference_) (PointCloudOdometry.cc:243-244): the
   current source becomes the target without leaving HBM; results identical to set_target of the same data */
lh_status lh_gicp_promote_source_to_target(lh_gicp* g);
/* pcl::Registration::align + computeTransformation (gicp.hpp:406-617).  guess NULL = identity.
   aligned_out (nullable): count*stride bytes, receives final_T * input xyz at off_xyz (gicp.hpp:586). */
lh_status lh_gicp_align(lh_gicp* g, const float guess[16], lh_gicp_result* out, lh_gicp_trace* trace,
                        void* aligned_out, uint32_t stride, uint32_t off_xyz);
/* getFitnessScore(max_range = DBL_MAX) of the last alignment (K7) */
lh_status lh_gicp_fitness(lh_gicp* g, double* fitness);

/* One huge pair sharded by SOURCE points over several GPUs: every rank holds the whole target
   (+ index) and a disjoint slice of the source (with its normals / covariances), sets this hook and makes the same
   lh_gicp_align / lh_gicp_fitness calls.  The hook sums `n` doubles in place over the ranks (RCCL/gloo all-reduce in the
   caller's runtime); it is called once per outer iteration with the 74 moment sums (cost_mode 1) or once per cost
   evaluation with the 13 sums + count of the functor (cost_mode 0, gicp.hpp:291-402), and every rank then runs the same
   BFGS on the same numbers, so all ranks return the same transform.  NULL removes the hook.  Non-zero return -> LH_EDEVICE.
   Not for lh_gicp_align_batch (ranks hold different pairs there: no exchange step at all). */
typedef int (*lh_allreduce_fn)(double* sums, int n, void* user);
lh_status lh_set_allreduce(lh_ctx* ctx, lh_allreduce_fn fn, void* user);
/* The same exchange WITHOUT leaving the device (cost_mode 1): with this hook the sharded pair runs the device-driven loop (k_solve) and the
   library calls fn between the moment reduction and the solve of every outer iteration, with the pair's 8 x 76 chunk sums where they lie in
   HBM and the HIP stream the iteration is queued on.  fn ENQUEUES an in-place SUM over the ranks on that stream -- for RCCL:
   ncclAllReduce(dev_sums, dev_sums, n, ncclDouble, ncclSum, comm, (hipStream_t)stream) -- and returns without synchronising: no host copy,
   no host synchronisation per iteration.  Every rank ends with the same bits, so every rank's k_solve takes the same decisions.  The host
   hook above still serves what is summed on the host (cost_mode 0, lh_gicp_fitness); install both (lh_rccl_install_sum_hook does).
   NULL removes the hook. */
typedef int (*lh_device_allreduce_fn)(double* dev_sums, int n, void* stream, void* user);
lh_status lh_set_device_allreduce(lh_ctx* ctx, lh_device_allreduce_fn fn, void* user);
/* getSearchMethodTarget()->nearestKSearch(pt, 1, ...) for every point of q (PointCloudLocalization.cc:327-336) */
lh_status lh_nn1(lh_gicp* g, const lh_cloud_view* q, int32_t* idx, float* d2);
lh_status lh_nn1_cloud(lh_cloud* target, const lh_cloud* q, int32_t* idx, float* d2);
/* k-NN on a cloud's own index (pcl::search::KdTree::nearestKSearch with k > 1; gicp.hpp:108-109) */
lh_status lh_knn_cloud(lh_cloud* target, const lh_cloud* q, int k, int32_t* idx, float* d2);

/* batched alignment of independent scan pairs on this context's GPU (BASELINE configs 4/5): pair p aligns
   src[p] -> tgt[p].  The NN index of every target is (re)built inside the call, like align() does.
   max_in_flight: pairs that share the GPU at any time (one HIP stream per group of 21-32 pairs; about 100 bytes of device
   workspace per source point and slot).  Throughput keeps growing with it until the streams cover each other's latencies: on
   100 k-point pairs 64 in flight ran 8 540 pairs/s, 512 in flight 10 200 (DESIGN.md sections 5, 7).  Every pair's own outcome
   (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN) is in out[i].status; the return value reports usage / device errors. */
lh_status lh_gicp_align_batch(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src,
                              lh_cloud* const* tgt, const float* guesses /* n_pairs*16 or NULL */,
                              lh_gicp_result* out, int max_in_flight /* 0 = default (64) */);
/* the same, plus align()'s output cloud of every pair (gicp.hpp:586, pcl::transformPointCloud(*input_, output,
   final_transformation_)): aligned[i] = final T * src[i], xyz transformed and every other field copied, written on the device
   as the pair retires.  aligned[i] == NULL on entry: a new device cloud is created (the caller destroys it); otherwise an
   existing cloud of src[i]'s size on this context, overwritten.  aligned == NULL: lh_gicp_align_batch. */
lh_status lh_gicp_align_batch_out(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src, lh_cloud* const* tgt,
                                  const float* guesses, lh_gicp_result* out, lh_cloud** aligned /* n_pairs, nullable */, int max_in_flight);
/* The odometry stream (PointCloudOdometry.cc:237-322 over a queue of n_scans device clouds): pair i aligns scans[i + 1] (the query) to
   scans[i] (the reference: `copyPointCloud(*query_, *reference_)` of the previous update), out[0 .. n_scans - 2].  A scan's index is built
   ONCE -- by this call, or before it by lh_normals_knn_batch -- and kept with the cloud, where lh_gicp_align_batch rebuilds every target
   (initCompute).  The index is a function of the cloud alone: results identical to lh_gicp_align_batch on the same pairs.
   guesses: (n_scans - 1) x 16 or NULL. */
lh_status lh_gicp_align_stream(lh_ctx* ctx, const lh_gicp_params* p, int n_scans, lh_cloud* const* scans, const float* guesses,
                               lh_gicp_result* out, int max_in_flight);

/* ---- several GPUs from one process (BASELINE configs 4/5) ---------------------------------------------------
   Independent scan pairs shard over GPUs with NO exchange step, so a single C++ process (the LOCUS node is one,
   locus/src/Locus.cc:47-71) needs no collective: create one context per visible device (lh_device_count, lh_create) and hand
   the whole batch over.
     lh_gicp_align_batch_multi        device-resident clouds: pair i runs on the context that owns src[i] and tgt[i] (both on the
                                      same one, which must be in ctxs); one host thread per device; results in pair order
     lh_gicp_align_batch_multi_views  host-resident clouds (PCL point arrays): contiguous blocks of pairs per context (pair i ->
                                      ctxs[i * n_ctx / n_pairs]), uploaded there, aligned, freed; when tgt[i] describes the same
                                      buffer as src[i-1] (an odometry stream) and both fall on one device it is uploaded once
   Contexts that share a device are served one after the other (see lh_create).  Ranks of a multi-process job (one process per
   GPU, torch.distributed / MPI) use the single-context calls and gather lh_gicp_result with their runtime's all-gather;
   liblocus_hip_rccl.so (include/locus_hip_rccl.h) does it with RCCL. */
int lh_device_count(void);
lh_status lh_gicp_align_batch_multi(int n_ctx, lh_ctx* const* ctxs, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src,
                                    lh_cloud* const* tgt, const float* guesses, lh_gicp_result* out, lh_cloud** aligned /* nullable */,
                                    int max_in_flight);
lh_status lh_gicp_align_batch_multi_views(int n_ctx, lh_ctx* const* ctxs, const lh_gicp_params* p, int n_pairs, const lh_cloud_view* src,
                                          const lh_cloud_view* tgt, const float* guesses, lh_gicp_result* out, int max_in_flight);

/* ---- building blocks exported for parity tests and for the localization wrapper ---------------- */
/* K3: computeCovariances k-NN branch (gicp.hpp:85-154) -> row-major 3x3 doubles [n][9] on the host */
lh_status lh_cov_knn(lh_cloud* c, int k, double gicp_epsilon, double* cov9_out);
/* K4: one NN + Mahalanobis sweep (gicp.hpp:464-498) with explicit transformation_ (col-major float) and guess.
   tgt_idx[n] (-1 unmatched), maha9 [n][9] row-major (entries of unmatched points are unspecified).
   Consecutive calls without changing the clouds are warm (previous neighbours + certificates), like align()'s sweeps. */
lh_status lh_gicp_debug_sweep(lh_gicp* g, const float T[16], const float guess[16], int32_t* tgt_idx, double* maha9);
/* one sweep of the cost_mode 1 kernels (the fused sweep for the first sweeps of a pair, k_late + k_walk afterwards: sweep_index
   selects exactly as align() would) with explicit transformation_; guess = identity, covariances from the clouds' normals.
   tgt_idx[n] = neighbour of every source point (-1 none) -- the state the certificates and second-chance tests maintain, so a
   test can hold it against a brute-force search after every sweep; *walks = queries that ran the tree traversal; sums74 = the 74
   moment sums of the sweep (nullable).  sweep_index 0 starts cold (seed pass); later calls are warm like align()'s sweeps. */
lh_status lh_gicp_debug_sweep_fused(lh_gicp* g, const float T[16], int sweep_index, int32_t* tgt_idx, uint64_t* walks, double* sums74);
/* instrumentation: out[0] = source points whose sweep ran the tree traversal, out[1] = source points swept (cumulative) */
lh_status lh_gicp_debug_stats(lh_gicp* g, uint64_t out[2], int reset);
/* instrumentation of the tree traversal (1-NN of T*q; cand = optional warm-start candidate per query, leaf_prescan = look at
   the candidate's leaf first): out = {sum node visits, sum leaf visits, sum over waves of the per-wave max visits, number of
   waves, max visits of any query} */
/* debug: the index as it lies in HBM -- sorted4: (n + 8) x 4 floats (x, y, z, original index; 8 pads), nodes: min(nodes_capacity, n) 64-byte 4-ary
   nodes (only those reachable from the root are meaningful), header64: the 64-byte tree header.  lh_debug_small_index(enable): clouds of at
   most 4 096 points are indexed by ONE launch of one workgroup (LOCUS's ~3 000-point operating point) -- 0 sends every cloud through the general
   build, 1 restores the default, a negative value only queries; returns the previous setting.  The two builds give identical trees. */
lh_status lh_debug_index_dump(lh_cloud* c, float* sorted4, void* nodes, uint32_t nodes_capacity, void* header64);
int lh_debug_small_index(int enable);
lh_status lh_debug_traversal_stats(lh_cloud* target, const lh_cloud* q, const float T[16], const int32_t* cand, int leaf_prescan,
                                   uint64_t out[5]);
/* K5: one cost-functor pass fdf(x) (gicp.hpp:362-402) on the correspondences of the last sweep */
lh_status lh_gicp_debug_cost(lh_gicp* g, const double x[6], double* f, double g6[6], double sums13[13], int* m);

/* K8: normalizePCloud + ComputeAp_ForPoint2PlaneICP (utils.cc:106-128, PointCloudLocalization.cc:723-750);
   corr[i] indexes `reference` normals; Ap row-major 6x6 */
lh_status lh_p2plane_information(lh_ctx* ctx, const lh_cloud* query, const lh_cloud* reference, const int64_t* corr,
                                 double Ap[36]);
/* H2: ComputePoint2PlaneICPCovariance conditioning (PointCloudLocalization.cc:487-538); host-side, 6x6 */
lh_status lh_icp_covariance(const double Ap[36], double icp_max_covariance, double cov[36], double* condition_number);

/* PointCloudLocalization::MeasurementUpdate's device work in ONE call on the handle's source (the query) and target (the reference),
   both device-resident since setInputSource / setInputTarget (PointCloudLocalization.cc:305-336, 398-421, 469-486, 694-750):
     icp_->align                                        -> result (lh_gicp_align's statuses; on LH_ETOO_FEW_CORR / LH_ESOLVER / LH_ENO_NN the
                                                           call goes on with the transform align left, like the reference does)
     transformPointCloudWithNormals(*query, aligned, T) -> aligned_out (nullable): xyz at off_xyz, normals at off_normal
                                                           (0xffffffff: no normals) of count * stride bytes
     nearestKSearch(aligned point, 1) for every point   -> corr (nullable, n int32; -1: a non-finite point has no neighbour)
     normalizePCloud(query) + ComputeAp_ForPoint2PlaneICP(query_normalized, reference, corr) -> Ap (row-major 6x6), want_information != 0
     0.05^2 Ap^-1, LDLT clamp, condition number         -> covariance, condition_number (lh_icp_covariance; covariance_ok 0 = its
                                                           "failed to find eigen values" return)
   Nothing is uploaded, and the host waits once: for the 21 sums of Ap, the correspondences and the aligned cloud it asked for.
   The pieces are the parity-tested ones (lh_gicp_align, lh_cloud_transform, lh_nn1, lh_p2plane_information, lh_icp_covariance)
   and the call returns their bits.  Returns lh_gicp_align's status. */
typedef struct lh_measurement {
  lh_gicp_result result;
  double Ap[36];
  double covariance[36];
  double condition_number;
  int32_t have_information;
  int32_t covariance_ok;
} lh_measurement;
lh_status lh_gicp_measurement_update(lh_gicp* g, const float guess[16], int want_information, double icp_max_covariance, lh_measurement* out,
                                     int32_t* corr, void* aligned_out, uint32_t stride, uint32_t off_xyz, uint32_t off_normal);
/* the same with the aligned query left in HBM (*aligned: a new cloud the caller destroys; nullable) -- the device-resident LOCUS flow
   (Locus.cc:474-489: scan -> fixed frame -> map neighbours -> sensor frame -> MeasurementUpdate -> map insert) then moves one scan up
   and two 6x6 matrices down per update */
lh_status lh_gicp_measurement_update_cloud(lh_gicp* g, const float guess[16], int want_information, double icp_max_covariance, lh_measurement* out,
                                           int32_t* corr, lh_cloud** aligned);

/* K1: CustomVoxelGrid::filter (custom_voxel_grid.cc:76-87): voxel centroid of x,y,z,intensity, pass-through
   limits on one axis (limit_axis -1 none / 0,1,2), output in ascending voxel index.  out = xyzi float[cap][4].
   *out_count is the number of voxels (may exceed cap: then only cap are written).  LH_EINVAL on int32 index overflow. */
lh_status lh_voxel_grid(lh_ctx* ctx, const lh_cloud_view* in, float leaf, int limit_axis, double lo, double hi,
                        float* out_xyzi, uint32_t out_capacity, uint32_t* out_count);
/* device-resident K1: the voxelised cloud (x, y, z, intensity centroids) is created on the GPU; with lh_normals_knn_cloud and
   lh_gicp_set_*_cloud a raw scan goes voxel grid -> normals -> GICP without crossing PCIe */
lh_status lh_cloud_voxel_grid(const lh_cloud* in, float leaf, int limit_axis, double lo, double hi, lh_cloud** out);
/* the second voxel-grid site of the path: pcl::VoxelGrid<PointF> in PointCloudFilter::Filter (PointCloudFilter.cc:119-124,
   PointF = pcl::PointXYZINormal, no limit field).  Same voxels in the same order as K1, but EVERY field is averaged the way
   pcl::CentroidPoint's accumulators do it: x, y, z, intensity and curvature are float sums / n, the normal is the float sum
   of the voxel's normals, normalised (a zero sum stays zero).  `in` must carry normals (LH_EINVAL otherwise). */
lh_status lh_cloud_voxel_grid_pointf(const lh_cloud* in, float leaf, lh_cloud** out);
/* K3 (filter flavour): NormalComputation::filter (normal_computation.cc:26-59), k-NN, viewpoint (0,0,0).
   out = float[count][4] (nx, ny, nz, curvature) */
lh_status lh_normals_knn(lh_ctx* ctx, const lh_cloud_view* in, int k, float* out_normals4);
lh_status lh_normals_knn_cloud(lh_cloud* c, int k); /* in place: fills the cloud's normals on the device */
/* the same for a queue of scans (every LOCUS scan passes the normal filter before UpdateEstimate): ONE batched index build for the
   clouds that have no index yet and ONE k-NN launch for all of them; the index stays with each cloud for the alignment that follows.
   Results are identical to n_clouds calls of lh_normals_knn_cloud. */
lh_status lh_normals_knn_batch(lh_cloud* const* clouds, int n_clouds, int k);
/* computeCovariances' k-NN branch (gicp.hpp:85-154) for a queue of clouds: what lh_gicp_* computes on demand when
   recompute_source_cov / recompute_target_cov is set, in one launch; the covariances stay with the clouds (valid for this k, epsilon) */
lh_status lh_cov_knn_batch(lh_cloud* const* clouds, int n_clouds, int k, double gicp_epsilon);
/* radius mode of the same nodelet (normal_search_method = radius, normal_computation.cc:71-74; default radius 0.3): every
   neighbour with d2 < radius^2 enters the covariance; fewer than 3 neighbours -> NaN normal and curvature.  The nodelet then
   drops those points (pcl::removeNaNNormalsFromPointCloud, normal_computation.cc:52-56): lh_cloud_remove_nan_normals is
   that order-preserving compaction on the device (new cloud; LH_EINVAL if nothing survives). */
lh_status lh_normals_radius(lh_ctx* ctx, const lh_cloud_view* in, float radius, float* out_normals4);
lh_status lh_normals_radius_cloud(lh_cloud* c, float radius);
lh_status lh_cloud_remove_nan_normals(const lh_cloud* in, lh_cloud** out);

/* ---- NDT: registration_method "ndt" = pclomp::NormalDistributionsTransform<PointF, PointF>
   (multithreaded_gicp/include/multithreaded_ndt/ndt_omp.h, ndt_omp_impl.hpp; SetupICP NDT branch PointCloudOdometry.cc:182-196).
   Neighbour search = KDTREE (the class default, never changed by LOCUS): the voxels whose CENTROID lies within `resolution` of
   the transformed point.  "Parity unpinned": the reference holds no NDT test or stored output. ---- */
typedef struct {
  float resolution;               /* setResolution, voxel edge (1.0) */
  int max_iterations;             /* setMaximumIterations (35; LOCUS: icp_iterations) */
  double step_size;               /* setStepSize: More-Thuente maximum step (0.1) */
  double outlier_ratio;           /* setOulierRatio (0.55) */
  double transformation_epsilon;  /* setTransformationEpsilon (0.1; LOCUS: icp_tf_epsilon) */
  double min_covar_eigvalue_mult; /* VoxelGridCovariance::setCovEigValueInflationRatio (0.01) */
  int min_points_per_voxel;       /* VoxelGridCovariance::setMinPointPerVoxel (6) */
  int reserved0;
} lh_ndt_params;
typedef struct lh_ndt lh_ndt;
void lh_default_ndt_params(lh_ndt_params* p);
lh_status lh_ndt_create(lh_ctx* ctx, const lh_ndt_params* p, lh_ndt** out);
void lh_ndt_destroy(lh_ndt* g);
lh_status lh_ndt_set_params(lh_ndt* g, const lh_ndt_params* p);          /* a new resolution re-initialises the voxel structure */
lh_status lh_ndt_set_source(lh_ndt* g, const lh_cloud_view* v);
lh_status lh_ndt_set_target(lh_ndt* g, const lh_cloud_view* v);          /* setInputTarget -> init(): voxel statistics of the target */
lh_status lh_ndt_set_source_cloud(lh_ndt* g, lh_cloud* c);               /* borrowed device clouds */
lh_status lh_ndt_set_target_cloud(lh_ndt* g, lh_cloud* c);
/* align + computeTransformation (ndt_omp_impl.hpp:101-212).  Result fields: T = final_transformation_, converged, iterations,
   fitness = getTransformationProbability() (score / n), cost_passes = device evaluations, n_correspondences_last = target cells */
lh_status lh_ndt_align(lh_ndt* g, const float guess[16], lh_gicp_result* out, void* aligned_out, uint32_t stride, uint32_t off_xyz);
/* test hooks: the target's cells (mean, inverse covariance, float centroid; ascending voxel index) and one evaluation of
   computeDerivatives (hessian_only = 0) / computeHessian (hessian_only = 1) at the pose p6 = (x, y, z, roll, pitch, yaw) */
lh_status lh_ndt_debug_cells(lh_ndt* g, int* n_cells, double* mean3, double* icov9, float* centroid4, int cap);
lh_status lh_ndt_debug_derivatives(lh_ndt* g, const double p6[6], int want_h, int hessian_only, double* score, double grad6[6], double hess36[36]);

/* SURVEY 8f-2 (next row): BodyFilter (body_filter.cc:27-52) = pcl::CropBox with min/max corners, a box yaw (setRotation(0, 0,
   rotation)) and setNegative(true): the points INSIDE the robot's body box are removed.  negative = 0 keeps the inside
   instead.  Order preserved, non-finite points dropped, normals / intensity travel along; LH_EINVAL if nothing survives.
   (The point is rotated into the box frame with cosf/sinf(yaw) in float; PCL inverts an Affine3f -- "parity unpinned" for
   points within rounding of a box face.) */
lh_status lh_cloud_crop_box(const lh_cloud* in, const float min_pt[3], const float max_pt[3], float yaw, int negative, lh_cloud** out);

/* SURVEY 8f-1 (next row): IPointCloudMapper::ApproxNearestNeighbors (Locus.cc:479-483) -- for every point of `query` (already in
   the map frame) its nearest map point, copied with normal and intensity into a new cloud of query-size (exact search) */
lh_status lh_cloud_nearest_neighbors(lh_cloud* map, const lh_cloud* query, lh_cloud** out);

/* The local map itself, device resident: IPointCloudMapper::InsertPoints / Refresh (Locus.cc:464-465, 531-538).
   point_cloud_mapper is un-vendored ("parity unpinned"); restated from its BLAM lineage: a point enters the map iff the
   octree voxel of edge `octree_resolution` it falls into is still empty, so the map keeps the FIRST point offered per voxel,
   in input order.  Voxel = floor(double(p) / resolution).  Non-finite points are never inserted.
     lh_map_insert   appends the accepted points of `points` (already in the fixed frame; normals / intensity travel along,
                     zeros if a later cloud lacks them); *n_inserted = how many (the reference's incremental_points size)
     lh_map_refresh  keeps the points with |p - center|_inf <= half_extent (mapper_->Refresh with box_filter_size,
                     lo_settings.yaml:58), order preserved
     lh_map_cloud    the map as a cloud handle (borrowed; NULL while empty): target of lh_gicp_set_target_cloud, first
                     argument of lh_cloud_nearest_neighbors; its NN index is rebuilt lazily after the map changed */
typedef struct lh_map lh_map;
lh_status lh_map_create(lh_ctx* ctx, double octree_resolution, lh_map** out);
void lh_map_destroy(lh_map* m);
lh_status lh_map_insert(lh_map* m, const lh_cloud* points, uint32_t* n_inserted);
lh_status lh_map_refresh(lh_map* m, const float center[3], float half_extent);
lh_cloud* lh_map_cloud(lh_map* m);
uint32_t lh_map_size(const lh_map* m);

/* ---- instrumentation (enableTimingOutput analogue) ------------------------- */
typedef struct {
  char name[32];
  uint64_t launches;
  double total_ms;  /* HIP-event time on the context's stream */
  double bytes;     /* algorithmic bytes accumulated with SURVEY.md 8d's model */
} lh_kernel_stat;
lh_status lh_profile_enable(lh_ctx* ctx, int on);
lh_status lh_profile_reset(lh_ctx* ctx);
int lh_profile_get(lh_ctx* ctx, lh_kernel_stat* out, int cap); /* returns number of entries */

#ifdef __cplusplus
}
#endif
#endif /* LOCUS_HIP_H_ */
