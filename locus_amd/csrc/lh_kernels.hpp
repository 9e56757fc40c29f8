// lh_kernels.hpp -- launch interface of the HIP kernels (implemented in lh_kernels.hip / lh_sort.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/locus_hip.h"
#include "lh_bfgs.hpp"
#include "lh_device.hpp"

namespace lh {

constexpr int MAX_JOBS = 32;       // jobs (scan pairs) per batched launch; kernarg stays < 4 KB (32 x 56 B)
constexpr int COST_CHUNK = 512;    // source points per cost-kernel workgroup (fixed => deterministic sums)
constexpr int COST_NSUM = 14;      // f, g_t[3], R[9], count
constexpr int MOM_CHUNK = 1024;    // source points per moment-kernel workgroup
constexpr int MOM_NSUM = 74;       // c0, B[3][4], H[6][10], count
constexpr int MOM_ROW = 76;        // a partial row on the device: the 74 sums, the number of tree walks, one pad

// static description of one scan pair's device buffers (lives in device memory, indexed by slot)
struct PairDesc {
  const float4* src;        // "output" cloud = guess * input, xyz1                      [n]
  const float4* src_nrm;    // source normals (cov-from-normals mode) or null
  const double* src_cov6;   // source covariances (k-NN mode), 6 planes of n_pad doubles, or null
  const float4* tgt_xyz;    // target xyz1 in ORIGINAL order                             [m]
  const float4* tgt_nrm;    // target normals or null
  const double* tgt_cov6;   // target covariances planes (stride m_pad) or null
  const float4* tgt_sorted; // Morton-sorted target (x,y,z,id)
  const NodeX* tgt_nodes;
  const TreeHeader* tgt_hdr;
  int32_t* prev_nn;         // warm-start NN index per source point                      [n]
  float4* cert;             // (query x,y,z at the last full search, lower bound on the other points' d2; < 0: that search found no neighbour) [n]
  float* rec;               // per source point the neighbour prev_nn points at, gathered: positions as packed triples [3 n_pad], then the
                            // normals [3 n_pad] -- written whenever prev_nn changes, so a sweep whose certificates hold reads ONE round of loads
  unsigned long long* stats; // [0] += queries that ran the tree traversal, [1] += queries (instrumentation)
  float4* corr;             // per source point: (tgt x, y, z, bitcast tgt idx | -1)     [n]
  double* maha6;            // 6 planes of n_pad doubles: M00 M01 M02 M11 M12 M22
  int n, n_pad, m, m_pad;
  int guess_identity;  // guess3 = I (the usual case): R = double(transformation_) without the 3x3 product
  int src_cov_pad;  // plane stride of src_cov6
  double corr_dist2;
  double gicp_eps;
  double guess3[9];  // top-left 3x3 of `guess` (row-major, as double): R = double(transformation_) * double(guess), gicp.hpp:450-460
  // device-driven loop (k_solve): the loop's knobs (gicp.h:119-130) and where the per-iteration trace goes (device memory, nullable)
  int max_iterations, max_inner_iterations;
  double rotation_epsilon, transformation_epsilon;
  lh_gicp_trace* trace;
  int bfgs_quad_curv, pad_b;   // lh_gicp_params::bfgs_quad_curv
};

struct SweepJob {  // dynamic per-launch part
  int slot;
  int pad;         // 1: the pair's first sweep, right after the seed pass (`cold`: certificate and neighbour record are not read)
  float T[12];     // transformation_ (row-major 3x4 float)
};
struct SweepArgs {
  int njobs;
  int bpj;         // workgroups per job
  int max_depth;   // deepest target tree among the jobs (sizes the LDS traversal stack)
  int pad;
  int span;        // source points per walk row / per k_walk wave (set by launch_sweep_fused)
  int refill;      // k_walk: idle lanes that trigger a refill from the walker queue
  float cert_rel;  // relative safety margin of the certificate test (set by the launch functions)
  int pad2;
  SweepJob job[MAX_JOBS];
};

struct SolveArgs {   // k_solve: one workgroup (one wave) per listed slot
  int njobs;
  int slot[MAX_JOBS];
};

struct CostJob {
  int slot;
  int out_offset;  // offset (in doubles) of this job's partial sums in the output buffer
  float T[12];
};
struct CostArgs {
  int njobs;
  int pad;         // k_moments_final: bit j set = job j's rows come from the fused sweep (one row per sweep_fused_wg() points instead of 256)
  CostJob job[MAX_JOBS];
};

// ---- K2: index build -------------------------------------------------------------------------------
size_t sort_temp_bytes(int n);
// keys/vals double buffers: on return sorted keys are in keys_out / vals_out
void sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                    uint32_t* vals_out, int n, int end_bit, hipStream_t s);

// ---- K2 batched: the indexes of several clouds are built by the same launches (grid.y = cloud) and ONE radix sort of
// the concatenated 64-bit keys (cloud id << 32 | 30-bit Morton key); the per-pair build was launch-bound.
struct IndexDesc {
  const float4* xyz;
  float4* sorted;     // [n + LEAF_CAP]
  NodeX* nodes;       // [n] worst case (one internal node per leaf, one leaf per point)
  TreeHeader* hdr;    // the start grid (GRID_ENTRIES x int32) lies between the header and the nodes
  int32_t* pos;       // (unused: the inverse permutation was written by every build and read by nothing)
  int n, offset;      // offset = start of this cloud in the concatenated key/value arrays
  int tile0, pad;     // first tile of this cloud in the sort's histogram table (sum of segsort_tiles of the clouds before it)
};
// build scratch shared by the clouds of a batch (capacity = total points of the batch + 1)
struct TreeScratch {
  const uint64_t* keys;   // sorted (cloud id << 32 | Morton key)       [total]
  uint32_t* flag;         // leaf-start flags                           [total]
  uint32_t* lid;          // inclusive scan of the flags                [total]
  uint32_t* tsum;         // leaves per tile of 4096 sorted positions (zero between builds)   [total / 4096 + 2]
  uint32_t* toff;         // ... and their exclusive prefix
  uint64_t* lkey;         // key of a leaf's first point (+ sentinel)   [leaves + 1]
  uint32_t* lstart;       // first sorted position of a leaf (+ sentinel)
  float4* lbox;           // leaf boxes, 2 x float4 (lo, hi) per leaf
  float4* a1box;          // boxes of 32 consecutive leaves
  float4* a2box;          // boxes of 1024 consecutive leaves
  float4* ibox;           // box of every binary node (its leaf range), 2 x float4   [leaves]
  int32_t* ichild;        // binary children of internal node i [2]: >= 0 internal, < 0 leaf ~index
  int32_t* irange;        // covered leaf range [2]
  int32_t* iparent;       // binary parent of internal node i (undefined for a root)
  int32_t* icom;          // leading bits of the 30-bit key shared by node i's range (key_common)
  int total;
};
constexpr size_t TREE_SCRATCH_BYTES_PER_POINT = 4 + 4 + 8 + 4 + 32 + 2 + 32 + 8 + 8 + 4 + 4 + 1;  // flag lid lkey lstart lbox a1+a2 ibox ichild irange iparent icom tsum
constexpr int MAX_INDEX_BATCH = 64;
// clouds of at most this many points are indexed by ONE launch of one workgroup each (lh_index_small.hip): LOCUS's operating point is ~3 000
// points per scan, where the general build's 13 launches are 122 us of launch latency around microseconds of work
constexpr int SMALL_INDEX_MAX_N = 4096;
void launch_index_small(const IndexDesc* descs_dev, int n_clouds, const TreeScratch& t, hipStream_t s);
// the build's own sort (lh_radix.hip): segmented 3 x 10-bit LSD radix sort of the 30-bit keys, every cloud inside its segment
int segsort_tiles(int n);
size_t segsort_hist_elems(long total_points, int n_clouds);
// kv_a: the (key, index) pairs of all clouds, concatenated in cloud order (clobbered); kv_b: scratch of the same size
void segsort_pairs(const IndexDesc* descs, int n_clouds, int max_n, uint64_t* kv_a, uint64_t* kv_b, uint64_t* keys_out, uint32_t* vals_out, uint32_t* hist,
                   hipStream_t s);
size_t sort64_temp_bytes(int n);
void sort_pairs_u64(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                    uint32_t* vals_out, int n, int end_bit, hipStream_t s);
void launch_index_bbox_init(uint32_t* bbox_enc /*[MAX_INDEX_BATCH][8]*/, hipStream_t s);   // once per allocation; every build leaves the slots reset
// pairs: (30-bit key, index) packed in 8 bytes, the segmented sort's input; keys / vals (nullable): the same as (cloud << 32 | key) + index arrays
void launch_index_keys(const IndexDesc* descs, int n_clouds, int max_n, uint32_t* bbox_enc /*[n_clouds][8]*/, uint64_t* pairs, uint64_t* keys,
                       uint32_t* vals, hipStream_t s);
// leaf-start flags + leaves per tile -> tile offsets -> running leaf numbers (t.lid), sorted points, inverse permutation, leaf records
void launch_index_leaves(const IndexDesc* descs, int n_clouds, const TreeScratch& t, const uint32_t* vals_sorted, uint32_t* bbox_enc, hipStream_t s);
// ... and everything after: stage 0: box tables (per leaf, per 32, per 1024 leaves), 1: radix hierarchy + one box per binary node,
// 2: 4-ary nodes + headers (separate so they can be timed)
void launch_index_trees(const IndexDesc* descs, int n_clouds, int max_n, const TreeScratch& t, hipStream_t s, int stage);

// ---- K4 / K5 ---------------------------------------------------------------------------------------
void launch_sweep(const PairDesc* descs, SweepArgs& a, int max_n, hipStream_t s);
// cost_mode 1: sweep + 74-moment reduction in one kernel (one partial per 256-point workgroup), then the final sum
// states == nullptr: host-driven loop (the jobs' transforms come with the launch); otherwise the pairs' device states (k_solve's)
// normals_only: no job of the launch uses k-NN covariances (recompute_*_cov) -> the rank-one Mahalanobis form
// split_mask: bit j set = job j is swept by the two-launch form (k_late + k_walk; needs covariances from normals, guess = I and
// wmask: one 64-bit walker mask per wave of 64 source points, mask_stride words per slot, zero-initialised); the others by the fused kernel
void launch_sweep_fused(const PairDesc* descs, SweepArgs& a, uint32_t split_mask, int max_n, double* partials_dev, int partials_stride,
                        const OuterState* states, bool normals_only, unsigned long long* wmask, int mask_stride, hipStream_t s);
int sweep_walk_span();    // source points per walk row (LH_WALK_SPAN, default 512)
int sweep_split_from();   // first outer iteration (0-based) swept in two launches (LH_SPLIT_FROM)
bool sweep_coop(bool normals_only);   // the fused jobs of such a launch are swept by k_sweep_coop: one partial row per 256 points (CostArgs::pad bit clear)
int sweep_greedy_flag(int sweep_index); // SweepJob::pad flag of a pair's sweep_index-th sweep (k_sweep_coop's greedy descent)
int sweep_fused_wg();      // source points per partial row of the FUSED sweep (its workgroup size: 256, or 64 with one wave per workgroup)
inline int sweep_rows(int n) { return (n + sweep_fused_wg() - 1) / sweep_fused_wg() + (n + sweep_walk_span() - 1) / sweep_walk_span(); }   // partial rows of one job at most: one per workgroup of the sweep + the walk rows
void launch_moments_final(const PairDesc* descs, const CostArgs& a, double* partials_dev, int partials_stride, double* out, const OuterState* states,
                          unsigned long long* wmask, int mask_stride, hipStream_t s);
// the BFGS solve + convergence test of one outer iteration, on the device (cost_mode 1): reads the FINAL_CHUNKS x MOM_ROW chunk
// sums k_moments_final left at chunks[slot * chunk_stride], updates states[slot]
void launch_solve(const PairDesc* descs, const SolveArgs& a, const double* chunks, int chunk_stride, OuterState* states, hipStream_t s);
constexpr int FUSED_CHUNK = 64;   // one 74-double partial per WAVE of the fused sweep
constexpr int FINAL_CHUNKS = 8;   // the final sum leaves FINAL_CHUNKS x 74 chunk sums per job for the host to add (in chunk order)
// cold-start helper: the nearest point of the leaf nearest to every 4th source point, written as the warm-start candidate of its group
void launch_seed(const PairDesc* descs, SweepArgs& a, int max_n, hipStream_t s);
constexpr int SEED_GROUP = 16;   // (4 until round 5: see k_seed)
__host__ __device__ inline int cost_blocks(int n) { return (n + COST_CHUNK - 1) / COST_CHUNK; }
// one pass of the cost functor: per-block sums in partials_dev[slot * partials_stride + block * COST_NSUM + k], then their sum in block
// order in out[job.out_offset + k] (k < COST_NSUM; pinned host memory for the host-driven loop)
void launch_cost(const PairDesc* descs, const CostArgs& a, int max_n, double* partials_dev, int partials_stride, double* out, hipStream_t s);
// second-order moments of the cost about T0 = job.T (see lh_bfgs.hpp MomentModel): per-block partials on the device,
// then one workgroup per job sums them in block order into out[job.out_offset + k], k < MOM_NSUM
void launch_moments(const PairDesc* descs, const CostArgs& a, int max_n, double* partials_dev, int partials_stride, double* out,
                    hipStream_t s);
inline int mom_blocks(int n) { return (n + MOM_CHUNK - 1) / MOM_CHUNK; }

// ---- K6 / K7 / misc ---------------------------------------------------------------------------------
void launch_transform(const float4* in_xyz, const float4* in_nrm, int n, const float* T12, float4* out_xyz, float4* out_nrm,
                      hipStream_t s);
// xyz transformed, normals / intensity copied unchanged (pcl::transformPointCloud on a PointXYZINormal cloud)
void launch_transform_copy(const float4* in_xyz, const float4* in_nrm, const float* in_int, int n, const float* T12, float4* out_xyz,
                           float4* out_nrm, float* out_int, hipStream_t s);
struct XformJob {
  const float4* in_xyz; const float4* in_nrm; const float* in_int;
  float4* out_xyz; float4* out_nrm; float* out_int;
  int n, pad;
  float T[12];
};
constexpr int MAX_XFORM_JOBS = 32;   // 32 x 104 B of launch arguments
struct XformBatchArgs { int njobs, pad; XformJob job[MAX_XFORM_JOBS]; };
void launch_transform_copy_batch(const XformBatchArgs& a, int max_n, hipStream_t s);
void launch_fill_i32(int32_t* p, int n, int32_t v, hipStream_t s);
// raw = a device copy of a host point array (n x stride bytes): its float fields into the path's arrays (nrm / inten nullable)
void launch_unpack_view(const void* raw, int n, uint32_t stride, uint32_t off_xyz, uint32_t off_normal, uint32_t off_intensity, uint32_t off_curvature,
                        float4* xyz, float4* nrm, float* inten, hipStream_t s);
// ungated 1-NN of T*q against a tree; T12 may be null (identity)
void launch_nn1(const float4* q, int nq, const float* T12, TreeView tree, int32_t* idx, float* d2, hipStream_t s);
// instrumentation: per-query visit counts of a cold 1-NN search; stats[0..4] = sum nodes, sum leaves, sum over waves of
// the per-wave max (nodes+leaves), number of waves, max (nodes+leaves) of any query
void launch_nn1_stats(const float4* q, int nq, const float* T12, TreeView tree, const float4* tgt_xyz,
                      const int32_t* cand /*nullable: warm-start candidates*/, unsigned long long* stats, hipStream_t s);
// deterministic double sum of the float d2 of the queries with idx >= 0, and their number (fitness): partials[ceil(n/1024)][2]
void launch_sum_f32(const float* v, const int32_t* idx, int n, double* partials, hipStream_t s);
inline int sum_blocks(int n) { return (n + 1023) / 1024; }

// ---- K3 ----------------------------------------------------------------------------------------------
void launch_knn(const float4* q, int nq, TreeView tree, int k, int32_t* idx, float* d2, hipStream_t s);
// k-NN covariance (gicp.hpp:85-154) -> 6 planes of n_pad doubles
void launch_knn_cov(const float4* xyz, int n, int n_pad, TreeView tree, int k, double eps, double* cov6, hipStream_t s);
// NormalEstimationOMP k-NN restated (normal_computation.cc:26-59): out = (nx,ny,nz,curvature)
void launch_knn_normals(const float4* xyz, int n, TreeView tree, int k, float4* out_nrm, hipStream_t s);
void launch_radius_normals(const float4* xyz, int n, TreeView tree, float radius, float4* out_nrm, hipStream_t s);
// The block k-NN search (lh_knn_block.hpp): every point of a cloud against its OWN cloud, one wave per 64 consecutive points of the
// Morton-sorted array, any number of clouds per launch.  One descriptor per cloud (device memory); the consumer is chosen by `mode`.
struct KnnCloudDesc {
  const float4* pts;        // Morton-sorted points (x, y, z, original index) + LEAF_CAP pads
  const NodeX* nodes;
  const TreeHeader* hdr;
  const float4* xyz;        // the cloud in its original order (the neighbours' coordinates are accumulated from here)
  float *sx, *sy, *sz;      // the sorted points' coordinates as three arrays [n + LEAF_CAP] (filled by the launch itself: a chunk of 8 candidates
                            // is then three scalar loads whose register PAIRS feed packed-f32 instructions, two candidates each)
  float4* nrm;              // KNN_MODE_NORMALS: (nx, ny, nz, curvature) per point                [n]
  double* cov6;             // KNN_MODE_COV: 6 planes of n_pad doubles (gicp.hpp:85-154)
  int32_t* idx;             // KNN_MODE_RAW: the k neighbour indices / squared distances per point [n * k]
  float* d2;
  int n, n_pad;
};
enum { KNN_MODE_NORMALS = 0, KNN_MODE_COV = 1, KNN_MODE_RAW = 2 };
constexpr int KNN_BLOCK_MAX_K = 32;   // larger k: the one-query-per-lane kernels above
// redo_cnt: one uint32 (reset by the launch), redo: room for every point of the batch -- the queries the block search hands to the
// one-query-per-lane search (ties at the k-th distance, pathological blocks); both launches are queued on `s`
void launch_knn_block(const KnnCloudDesc* descs_dev, int n_clouds, int max_n, int k, int mode, double eps, uint32_t* redo_cnt, uint2* redo,
                      hipStream_t s);
void launch_finite_normal_flags(const float4* nrm, int n, uint32_t* flags, hipStream_t s);
void launch_compact(const uint32_t* incl, int n, const float4* xyz, const float4* nrm, const float* inten, float4* oxyz, float4* onrm,
                    float* ointen, hipStream_t s);

void launch_crop_flags(const float4* xyz, int n, const float* mn, const float* mx, float c, float s, int negative, uint32_t* flags, hipStream_t st);

// ---- NDT ---------------------------------------------------------------------------------------
struct NdtFrame;
constexpr int NDT_ROW = 44;   // a partial row: score, gradient[6], hessian[36], one pad
struct NdtVoxelRaw { double sum[3]; double cov[6]; float cen[3]; int count; };   // raw per-voxel sums (xx xy xz yy yz zz)
void launch_ndt_voxel_stats(const float4* xyz, const uint32_t* keys, const uint32_t* vals, const uint32_t* heads, const uint32_t* rank_incl,
                            int n, NdtVoxelRaw* out, hipStream_t s);
void launch_ndt_finish_cells(const NdtVoxelRaw* raw, int n_vox, int min_points, double eig_mult, double* mean, double* icov, float4* cen,
                             uint32_t* flags, hipStream_t s);
void launch_ndt_compact_cells(const uint32_t* incl, int n_vox, const double* mean, const double* icov, const float4* cen, double* omean,
                              double* oicov, float4* ocen, hipStream_t s);
// one evaluation: per-wave rows [ceil(n/256)*4][NDT_ROW] -> FINAL_CHUNKS chunk sums (the host adds them in chunk order)
void launch_ndt_derivs(const float4* src, int n, TreeView cells, const double* mean, const double* icov, const NdtFrame& f, int hessian_only,
                       double* rows_dev, double* out_chunks, hipStream_t s);

// ---- local map ------------------------------------------------------------------------------
void launch_map_keys(const float4* xyz, int n, double inv_res, uint64_t* keys, uint32_t* vals /*nullable*/, hipStream_t s);
void launch_map_accept(const uint64_t* skeys, const uint32_t* svals, int n, const uint64_t* map_keys, int m, uint32_t* accept, hipStream_t s);
void launch_box_flags(const float4* xyz, int n, float cx, float cy, float cz, float half, uint32_t* flags, hipStream_t s);
void launch_map_compact(const uint32_t* incl, int n, const float4* xyz, const float4* nrm, const float* inten, double inv_res, int dst_off,
                        float4* oxyz, float4* onrm, float* ointen, uint64_t* okeys, hipStream_t s);

// ---- K8 ----------------------------------------------------------------------------------------------
// stage 1: per-block partial sums (x,y,z float-in-double) for the centroid; stage 2 etc. are in lh_api.hip
void launch_ap(const float4* qnorm, int n, const float4* ref_nrm, const int64_t* corr, double* partials /*blocks*21*/, hipStream_t s);
inline int ap_blocks(int n) { return (n + 1023) / 1024; }

// ---- K1 (pcl::VoxelGrid, custom_voxel_grid.cc:76-87) ------------------------------------------------------
// bbox over finite points passing the pass-through limits on one axis (ordered-uint encoding, 6 values)
void launch_voxel_bbox(const float4* xyzi, int n, int limit_axis, float lo, float hi, uint32_t* bbox_enc, hipStream_t s);
struct VoxelGridDesc { float inv_leaf; int limit_axis; float lo, hi; int minb[3]; int mul[3]; };
// key = linear voxel index (0xFFFFFFFF for rejected points), val = point index
void launch_voxel_keys(const float4* xyzi, int n, VoxelGridDesc g, uint32_t* keys, uint32_t* vals, hipStream_t s);
// heads[i] = 1 where a new voxel starts in the sorted key array
void launch_voxel_heads(const uint32_t* keys, int n, uint32_t* heads, hipStream_t s);
// one thread per voxel head: sequential float centroid of the segment in sorted (= input) order
// (nrm / out_nrm non-null: the PointXYZINormal flavour -- normals summed and normalised, curvature averaged)
void launch_voxel_centroids(const float4* xyzi, const float4* nrm, const uint32_t* keys, const uint32_t* vals, const uint32_t* heads,
                            const uint32_t* rank_incl, int n, float4* out, float4* out_nrm, uint32_t out_cap, hipStream_t s, const float* in_inten = nullptr,
                            float* out_inten = nullptr);
size_t sort_keys64_temp_bytes(int n);
void sort_keys_u64(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, int n, hipStream_t s);
size_t scan_temp_bytes(int n);
void inclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, int n, hipStream_t s);

}  // namespace lh
