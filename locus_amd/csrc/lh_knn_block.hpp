// lh_knn_block.hpp -- K3, the block k-NN search: per-lane pieces shared by k_knn_block (lh_kernels.hip) and the host model that
// checks the algorithm against an exhaustive search (tools/model/knn_block_model.cpp).
//
// The k nearest neighbours of every point of a cloud IN ITS OWN cloud (pcl::NormalEstimation, normal_computation.cc:35-36; gicp.hpp:
// 85-154) are found by one WAVE per block of 64 consecutive points of the Morton-sorted array -- 64 queries that sit next to each
// other in space and therefore want (nearly) the same candidates:
//   * lane = query.  The candidates are read at wave-uniform addresses (scalar loads: every lane sees the same 8-point chunk) and each
//     lane keeps, in registers, only the K smallest squared distances it has seen, as sorted 32-bit keys: a chunk's 8 keys are sorted by a
//     19-comparator network and merged into the list by a pruned bitonic merge (lh_knn_net.hpp: 96 v_min_u32 / v_max_u32 for K = 20) --
//     no (distance, index) pairs, no insertion chains, no divergence.
//   * pass 1: the block's own stretch of the sorted array (the window: the 64 points and 32 on either side) gives every lane a first
//     bound; then the wave walks the cloud's tree ONCE for all its lanes (a uniform stack of (child, box) entries): a child is visited if
//     ANY lane's ball (its current k-th distance) reaches the child's box (the child most lanes want is visited first), leaves are
//     clipped against the window (no point is offered twice) and merged.  Every chunk in which some lane found a key <= its bound is
//     remembered (<= 192 chunks).  The stack and the chunk list are wave-uniform: they live in the lanes of a few vector registers.
//   * pass 2: with tau = the lane's exact k-th smallest distance, the remembered chunks are read once more and every candidate with
//     d <= tau leaves its sorted position in the lane's column of an LDS table -- exactly k entries unless the cloud holds ties at tau.
//   * the k (d, original index) pairs are then formed from the table, sorted as 64-bit keys (ascending distance, lowest index first:
//     the order nearestKSearch returns and the moments are accumulated in) and handed to the consumer (normal / covariance / raw list).
// Ties at the k-th distance (more than k candidates within tau: about one query per 100 k-point lidar scan, everywhere in lattices) are
// settled by the wave itself: the lanes concerned run a (d2, index) insertion list over the remembered chunks.  A block that cannot be
// finished at all -- a full chunk list or stack, an infinite bound with k points available -- goes to a redo list served by the
// one-query-per-lane search (tree_search + KnnRegCollector).  Every result is the exact (d2, index)-lexicographic k-NN set whatever
// the data look like.
// Exactness of the pruning is the tree's usual argument: boxd2_q is a lower bound of the float distance of every point in the box, a
// lane's bound only ever shrinks, and a subtree is dropped only when its bound EXCEEDS every lane's current k-th distance.
#pragma once
#include "lh_device.hpp"
#include "lh_knn_net.hpp"

namespace lh {

constexpr int KNN_BLOCK_Q = 64;      // queries per wave
constexpr int KNN_WIN_SIDE = 32;     // sorted positions on either side of the block that belong to the window
constexpr int KNN_ACC_CAP = 192;     // remembered chunks per block
constexpr int KNN_STACK_CAP = 64;    // the walk's stack: one lane of four vector registers per entry (observed depth <= 27 on lidar scans; a deeper walk goes to the redo list)
constexpr uint32_t KNN_KEY_INF = 0x7f800000u;

// reasons a lane goes to the redo list (instrumentation only)
enum { KNN_FAIL_TIES = 1, KNN_FAIL_CHUNKS = 2, KNN_FAIL_STACK = 4, KNN_FAIL_INF = 8 };

// chunk reference: (first sorted position << 4) | (count - 1), count 1..8 -- the positive form of a leaf reference
LH_HD uint32_t knn_chunk_ref(uint32_t first, int cnt) { return (first << 4) | (uint32_t)(cnt - 1); }

// the part of [first, first + cnt) outside the window [w0, w1): a leaf is a run of <= 8 sorted positions and the window is longer than
// any leaf, so what is left is a prefix or a suffix (or nothing: false)
LH_HD bool knn_clip_chunk(int first, int cnt, int w0, int w1, int& f2, int& c2) {
  int lo = first, hi = first + cnt;
  if (lo >= w0 && hi <= w1) return false;
  if (lo < w0 && hi > w0) hi = w0;        // head sticks out below the window
  else if (lo < w1 && hi > w1) lo = w1;   // tail sticks out above
  f2 = lo; c2 = hi - lo;
  return c2 > 0;
}

// the 8 keys of a chunk for one query: float bits of (dx^2 + dy^2) + dz^2 (d2f's operation order), +INF past the chunk's count
LH_HD void knn_chunk_keys(float qx, float qy, float qz, const float4* p, int cnt, uint32_t* B) {
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const float x = e < cnt ? p[e].x : inf_f();   // (uniform select; INF - q = INF, INF^2 + finite = INF: no NaN for finite queries)
    B[e] = f2u(d2f(qx, qy, qz, x, p[e].y, p[e].z));
  }
}
LH_HD uint32_t knn_min8(const uint32_t* B) {
  uint32_t a = B[0] < B[1] ? B[0] : B[1], b = B[2] < B[3] ? B[2] : B[3], c = B[4] < B[5] ? B[4] : B[5], d = B[6] < B[7] ? B[6] : B[7];
  a = a < b ? a : b; c = c < d ? c : d;
  return a < c ? a : c;
}
// The list of a lane: K keys ascending.  A search for k < K neighbours starts with K - k PHANTOM keys of 0 in front of the +INF
// fill: they can never be displaced (no key is smaller), so the K-th smallest key of the list is always the k-th smallest REAL key seen so
// far -- the lane's bound is L[K - 1] for every k, without a run-time index into the registers.
template <int K>
LH_HD void knn_list_init(uint32_t* L, int k) {
#pragma unroll
  for (int j = 0; j < K; j++) L[j] = j < K - k ? 0u : KNN_KEY_INF;
}

}  // namespace lh
