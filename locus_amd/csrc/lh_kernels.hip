// lh_kernels.hip -- hand-written HIP kernels of the GICP hot path for gfx950 (MI355X, wave64).
// Compiled with -ffp-contract=off: float/double expressions keep one rounding per operation so that the
// results can be compared bit-for-bit (indices) / to 1e-12 (doubles) with the CPU oracle.
//
// Kernel <-> reference loop:
//   k_bbox / k_morton / k_gather_sorted / k_leaf_level / k_level_up   K2  tree_->setInputCloud (initCompute)
//   k_sweep                                                           K4  gicp.hpp:464-498 (+K3' fused, gicp.hpp:81-82)
//   k_cost                                                            K5  gicp.hpp:362-402
//   k_transform                                                       K6  gicp.hpp:440,586
//   k_nn1 / k_sum_f32                                                 K7  getFitnessScore; PointCloudLocalization.cc:327-336
//   k_knn / k_knn_cov / k_knn_normals                                 K3  gicp.hpp:85-154; normal_computation.cc:26-59
//   k_ap                                                              K8  PointCloudLocalization.cc:723-750
#include <cstdlib>

#include "lh_kernels.hpp"
#include "lh_launch.hpp"
#include "lh_ndt.hpp"
#include "lh_terms.hpp"

namespace lh {

// ===== K2: index build ===================================================================================
__device__ __forceinline__ uint32_t enc_ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(uint32_t e) {
  uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  return __uint_as_float(u);
}

// ----- batched index build -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bbox_init_b(uint32_t* bbox, int n_clouds) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_clouds * 8) return;
  int slot = t & 7;
  bbox[t] = slot < 3 ? 0xffffffffu : 0u;
}
__global__ void __launch_bounds__(256) k_bbox_b(const IndexDesc* __restrict__ descs, uint32_t* bbox) {
  const IndexDesc d = descs[blockIdx.y];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int stride = gridDim.x * blockDim.x;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < d.n; i += 4 * stride) {   // four independent loads in flight
    float4 p[4];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = d.xyz[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      lo[0] = fminf(lo[0], p[k].x); hi[0] = fmaxf(hi[0], p[k].x);
      lo[1] = fminf(lo[1], p[k].y); hi[1] = fmaxf(hi[1], p[k].y);
      lo[2] = fminf(lo[2], p[k].z); hi[2] = fmaxf(hi[2], p[k].z);
    }
  }
  for (; i < d.n; i += stride) {
    float4 p = d.xyz[i];
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  __shared__ float sm[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { sm[threadIdx.x >> 6][a] = lo[a]; sm[threadIdx.x >> 6][3 + a] = hi[a]; }
  }
  __syncthreads();
  uint32_t* bb = bbox + blockIdx.y * 8;
  if (threadIdx.x < 3)
    atomicMin(&bb[threadIdx.x], enc_ordered(fminf(fminf(sm[0][threadIdx.x], sm[1][threadIdx.x]), fminf(sm[2][threadIdx.x], sm[3][threadIdx.x]))));
  else if (threadIdx.x < 6)
    atomicMax(&bb[threadIdx.x], enc_ordered(fmaxf(fmaxf(sm[0][threadIdx.x], sm[1][threadIdx.x]), fmaxf(sm[2][threadIdx.x], sm[3][threadIdx.x]))));
}
// (key, index) leave as ONE 8-byte pair, the form the segmented sort keeps between its passes (lh_radix.hip); keys64 / vals32 (nullable)
// are the same as separate arrays for the LH_SORT=check / generic debug paths
__global__ void __launch_bounds__(256) k_key_b(const IndexDesc* __restrict__ descs, const uint32_t* __restrict__ bbox,
                                               uint2* __restrict__ pairs, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const IndexDesc d = descs[blockIdx.y];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* bb = bbox + blockIdx.y * 8;
  const bool grid_on = d.n >= GRID_MIN_POINTS;
  if (i == 0) {  // the grid the node boxes are quantised on (also for empty clouds: block 0 always runs)
    const float lo[3] = {dec_ordered(bb[0]), dec_ordered(bb[1]), dec_ordered(bb[2])};
    const float hi[3] = {dec_ordered(bb[3]), dec_ordered(bb[4]), dec_ordered(bb[5])};
    const bool ok = quant_frame(lo, hi, d.hdr);
    d.hdr->grid_on = (grid_on && ok) ? 1 : 0;   // (its tables are filled by k_nodex_b, the last launch of this build; a cloud with a
                                                // non-finite point has no usable key grid: its walks start at the root)
  }
  if (grid_on) {   // the start grid behind the header: every cell empty until k_nodex_b says otherwise
    int32_t* grid = reinterpret_cast<int32_t*>(d.hdr + 1);
    for (int k = i; k < GRID_ENTRIES; k += gridDim.x * 256) grid[k] = GRID_EMPTY;
  }
  if (i >= d.n) return;
  float4 p = d.xyz[i];
  // 30 bits (7.8 cm cells on an 80 m scene) are enough: 16 bits/axis (spatial_key48) leaves the visit counts unchanged
  uint32_t k = spatial_key30(p.x, p.y, p.z, dec_ordered(bb[0]), dec_ordered(bb[1]), dec_ordered(bb[2]), dec_ordered(bb[3]),
                             dec_ordered(bb[4]), dec_ordered(bb[5]));
  pairs[d.offset + i] = make_uint2(k, (uint32_t)(d.offset + i));
  if (keys) {
    keys[d.offset + i] = ((uint64_t)blockIdx.y << 32) | k;
    vals[d.offset + i] = (uint32_t)(d.offset + i);
  }
}
// --- leaves: the largest key-prefix cell around every sorted position with <= LEAF_CAP points --------------------------
// One workgroup per tile of LEAF_TILE sorted positions.  k_leafcell_b: the leaf-start flags and the tile's number of leaves;
// (every k_leaves_b workgroup adds the counts of the tiles below its own: no scan launch); k_leaves_b: the same tile again -- the running leaf
// number of every position (lid = inclusive scan of the flags), the sorted points (x, y, z, original index) gathered through the
// sorted index, the inverse permutation, and for every flagged position the leaf's record.  (Round 2: flags, a library scan with
// its own init kernel, gather, leaf records = five launches and two more round trips through the flags.)
constexpr int LEAF_TILE = 4096, LEAF_PER_THREAD = LEAF_TILE / 256;
__device__ __forceinline__ uint32_t block_exclusive_u32(uint32_t mine, uint32_t* wsum /*[4] shared*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t b0 = inc - mine;
  for (int w = 0; w < wave; w++) b0 += wsum[w];
  if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return b0;
}
__global__ void __launch_bounds__(256) k_leafcell_b(TreeScratch t) {   // one sorted position per thread
  __shared__ uint64_t win[256 + 2 * LEAF_CAP];   // the workgroup's keys + LEAF_CAP on either side: every thread looks at keys[g - 8 .. g + 8]
  __shared__ uint32_t wcnt[4];
  const int64_t g0 = (int64_t)blockIdx.x * 256, g = g0 + threadIdx.x;
  for (int k = threadIdx.x; k < 256 + 2 * LEAF_CAP; k += 256) {
    const int64_t q = g0 - LEAF_CAP + k;
    win[k] = (q >= 0 && q < t.total) ? t.keys[q] : 0ull;   // (positions outside the array are never compared: leafcell_flag checks the bounds)
  }
  __syncthreads();
  uint32_t f = 0;
  if (g < t.total) {
    f = leafcell_flag(win + LEAF_CAP - g0, t.total, g);     // same function, the window addressed like the whole array
    t.flag[g] = f;
  }
  const uint32_t c = (uint32_t)__popcll(__ballot(f != 0u));
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = c;
  __syncthreads();
  // leaves per tile of LEAF_TILE positions (16 workgroups each): integer adds, order-free.  tsum is zero when a build starts (k_boxes_b
  // of the previous build left it so)
  if (threadIdx.x == 0) atomicAdd(&t.tsum[(blockIdx.x * 256) / LEAF_TILE], wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
}
__global__ void __launch_bounds__(256) k_leaves_b(const IndexDesc* __restrict__ descs, TreeScratch t, const uint32_t* __restrict__ vals, uint32_t* bbox_next) {
  __shared__ uint32_t cnt[LEAF_PER_THREAD][4];   // leaves per (round, wave) of the tile, then their exclusive prefix in (round, wave) order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  // (tile, round, thread) order: consecutive lanes hold consecutive sorted positions -> every load and store of a round is coalesced
  uint32_t f[LEAF_PER_THREAD];
  unsigned long long bal[LEAF_PER_THREAD];
#pragma unroll
  for (int r = 0; r < LEAF_PER_THREAD; r++) {
    const int64_t g = (int64_t)blockIdx.x * LEAF_TILE + r * 256 + threadIdx.x;
    f[r] = g < t.total ? t.flag[g] : 0u;
    bal[r] = __ballot(f[r] != 0u);
    if (lane == 0) cnt[r][wave] = (uint32_t)__popcll(bal[r]);
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // 64 (round, wave) counts -> exclusive prefix, one wave
    const uint32_t v = (&cnt[0][0])[threadIdx.x];
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    (&cnt[0][0])[threadIdx.x] = inc - v;
  }
  __syncthreads();
  // The tile's first leaf id = the leaves of all earlier tiles: every workgroup adds the (<= total / 4096) counts below its own itself, in
  // parallel -- a separate one-workgroup scan launch (k_tile_offsets, round 2-5: 5 us + its launch gap on the build's critical path) for
  // a sum that 256 threads form in a microsecond.  Integer adds: order-free.  The counts are zeroed for the next build by k_boxes_b.
  __shared__ uint32_t tb_part[4];
  {
    uint32_t s = 0;
    for (int tt = threadIdx.x; tt < (int)blockIdx.x; tt += 256) s += t.tsum[tt];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) tb_part[wave] = s;
  }
  __syncthreads();
  const uint32_t tile_base = (tb_part[0] + tb_part[1]) + (tb_part[2] + tb_part[3]);
  int cur = -1;
  const float4* xyz = nullptr; float4* sorted = nullptr; int d_off = 0, d_n = 0;
#pragma unroll 4
  for (int r = 0; r < LEAF_PER_THREAD; r++) {
    const int64_t g = (int64_t)blockIdx.x * LEAF_TILE + r * 256 + threadIdx.x;
    if (g >= t.total) continue;
    const uint32_t lid = tile_base + cnt[r][wave] + (uint32_t)__popcll(bal[r] & below) + f[r];   // inclusive scan of the flags
    t.lid[g] = lid;
    const uint64_t key = t.keys[g];
    const int cloud = (int)(key >> 32);
    if (cloud != cur) {   // (a tile holds one cloud, two at a boundary)
      const IndexDesc& d = descs[cloud];
      xyz = d.xyz; sorted = d.sorted; d_off = d.offset; d_n = d.n; cur = cloud;
    }
    const uint32_t j = vals[g] - (uint32_t)d_off;            // original index of the point at sorted position g
    const uint32_t i = (uint32_t)(g - d_off);                 // its position inside the cloud's own sorted array
    const float4 p = xyz[j];
    sorted[i] = make_float4(p.x, p.y, p.z, __uint_as_float(j));
    if ((int)i == d_n - 1) {                                  // padding behind the last leaf (scan_leaf loads LEAF_CAP entries)
#pragma unroll
      for (int e = 1; e <= LEAF_CAP; e++) sorted[i + e] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0x7fffffffu));
    }
    if (f[r]) {                                               // leaf L = lid - 1 starts here
      t.lkey[lid - 1u] = key;
      t.lstart[lid - 1u] = (uint32_t)g;
    }
    if (g == t.total - 1) {                                   // sentinel: leaf L's points are [lstart[L], lstart[L+1])
      t.lstart[lid] = (uint32_t)t.total;
      t.lkey[lid] = ~0ull;
    }
  }
  // the bounding-box slots of this batch have been consumed (k_key_b): reset them for the NEXT build of this scratch set (builds on one
  // set are serialised), which saves every build its own init launch
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < MAX_INDEX_BATCH * 8; k += 256) bbox_next[k] = (k & 7) < 3 ? 0xffffffffu : 0u;
}
// --- Karras' binary radix tree over the leaf keys of the WHOLE batch (nodes that join two clouds are never used) -------
// ... and, in the same thread, the box of the node: the min/max over its contiguous range of leaves, read off three tables --
// lbox (one box per leaf), a1box (per 32 consecutive leaves), a2box (per 1024) -- written by earlier launches: no arrival counters,
// no agent-scope fences (a __threadfence() per tree level costs ~3.5 us on this chip because the per-XCD L2s are not coherent:
// the climbing version took 3.5 ms per batch).  ONE range box per binary node; the 4-ary nodes are assembled from them afterwards
// (round 2 computed four range boxes per 4-ary node in k_nodex_b: twice the table reads, and the launch lasted as long as the
// thread of a root, ~560 dependent table entries).
struct Box6 { float lx, ly, lz, hx, hy, hz; };
__device__ __forceinline__ Box6 box_empty() { return Box6{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}; }
__device__ __forceinline__ void box_merge(Box6& b, const float4* __restrict__ tab, int idx) {  // table entry = 2 x float4
  float4 lo = tab[2 * (size_t)idx], hi = tab[2 * (size_t)idx + 1];
  b.lx = fminf(b.lx, lo.x); b.ly = fminf(b.ly, lo.y); b.lz = fminf(b.lz, lo.z);
  b.hx = fmaxf(b.hx, hi.x); b.hy = fmaxf(b.hy, hi.y); b.hz = fmaxf(b.hz, hi.z);
}
__device__ __forceinline__ void box_store(float4* tab, int idx, const Box6& b) {
  tab[2 * (size_t)idx] = make_float4(b.lx, b.ly, b.lz, 0.f);
  tab[2 * (size_t)idx + 1] = make_float4(b.hx, b.hy, b.hz, 0.f);
}
// The three box tables in ONE launch (round 6; k_leafbox_b + two k_chunkbox_b launches before): a workgroup takes 1 024 consecutive leaves in
// four rounds of 256 -- a thread forms a leaf's box from its (<= LEAF_CAP) points in one round of loads (the sorted array is padded, the
// count masks) -> lbox; 32 consecutive lanes hold 32 consecutive leaves, five shuffle steps give their union -> a1box; the workgroup's 32
// chunk boxes meet in LDS -> a2box.  min / max are exact and order-free: the tables are the separate launches' byte for byte.
__global__ void __launch_bounds__(256) k_boxes_b(const IndexDesc* __restrict__ descs, TreeScratch t, int tiles) {
  __shared__ float cb[32][6];
  {  // the tile counts of k_leafcell_b have been consumed by k_leaves_b: zero for the next build of this scratch set
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid < tiles) t.tsum[gid] = 0u;
  }
  const int n_leaves = (int)t.lid[t.total - 1];
  const int base = blockIdx.x * 1024;
  if (base >= n_leaves) return;   // (uniform)
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    const int L = base + r * 256 + (int)threadIdx.x;
    Box6 b = box_empty();
    if (L < n_leaves) {
      const int cloud = (int)(t.lkey[L] >> 32);
      const IndexDesc d = descs[cloud];
      const uint32_t s0 = t.lstart[L], s1 = t.lstart[L + 1];
      const float4* pp = d.sorted + (s0 - (uint32_t)d.offset);
      float4 q[LEAF_CAP];
#pragma unroll
      for (int e = 0; e < LEAF_CAP; e++) q[e] = pp[e];
#pragma unroll
      for (int e = 0; e < LEAF_CAP; e++)
        if (s0 + (uint32_t)e < s1) {
          b.lx = fminf(b.lx, q[e].x); b.ly = fminf(b.ly, q[e].y); b.lz = fminf(b.lz, q[e].z);
          b.hx = fmaxf(b.hx, q[e].x); b.hy = fmaxf(b.hy, q[e].y); b.hz = fmaxf(b.hz, q[e].z);
        }
      box_store(t.lbox, L, b);
      const int a_c = (int)t.lid[d.offset] - 1, b_c = (int)t.lid[d.offset + d.n - 1] - 1;
      if (a_c == b_c) {  // the whole cloud is one leaf: no internal node will write the header
        d.hdr->root = leaf_ref(0u, d.n);
        d.hdr->n_leaves = 1;
      }
    }
    // a1: the union of the 32 leaves of a half wave (lanes past the last leaf hold the empty box)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      b.lx = fminf(b.lx, __shfl_down(b.lx, off, 32)); b.ly = fminf(b.ly, __shfl_down(b.ly, off, 32)); b.lz = fminf(b.lz, __shfl_down(b.lz, off, 32));
      b.hx = fmaxf(b.hx, __shfl_down(b.hx, off, 32)); b.hy = fmaxf(b.hy, __shfl_down(b.hy, off, 32)); b.hz = fmaxf(b.hz, __shfl_down(b.hz, off, 32));
    }
    if ((threadIdx.x & 31) == 0) {
      const int c = (base + r * 256 + (int)threadIdx.x) >> 5;   // chunk of 32 leaves
      if (c * 32 < n_leaves) box_store(t.a1box, c, b);
      float* w = cb[r * 8 + (threadIdx.x >> 5)];
      w[0] = b.lx; w[1] = b.ly; w[2] = b.lz; w[3] = b.hx; w[4] = b.hy; w[5] = b.hz;
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {   // a2: the union of the workgroup's 32 chunks (empty boxes where the cloud batch ends)
    const float* w = cb[threadIdx.x];
    Box6 b{w[0], w[1], w[2], w[3], w[4], w[5]};
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      b.lx = fminf(b.lx, __shfl_down(b.lx, off, 32)); b.ly = fminf(b.ly, __shfl_down(b.ly, off, 32)); b.lz = fminf(b.lz, __shfl_down(b.lz, off, 32));
      b.hx = fmaxf(b.hx, __shfl_down(b.hx, off, 32)); b.hy = fmaxf(b.hy, __shfl_down(b.hy, off, 32)); b.hz = fmaxf(b.hz, __shfl_down(b.hz, off, 32));
    }
    if (threadIdx.x == 0) box_store(t.a2box, blockIdx.x, b);
  }
}
// box of the leaves [a, e): entries fetched four at a time, independent loads in flight together, merged afterwards (min / max are
// exact and order-free, so the boxes do not depend on how they are gathered)
__device__ __forceinline__ void box_merge_run(Box6& b, const float4* __restrict__ tab, int first, int last) {
  constexpr int W = 8;   // (one box per thread now: registers for eight entries in flight)
  // every batch issues its W loads together; a short batch repeats its last entry (min / max do not mind) -- a one-by-one tail was a
  // chain of dependent loads, and most nodes of a tree have ranges of two to seven leaves: ALL of their entries went through it
  for (int l = first; l < last; l += W) {
    float4 lo[W], hi[W];
#pragma unroll
    for (int k = 0; k < W; k++) {
      const int idx = l + k < last ? l + k : last - 1;
      lo[k] = tab[2 * (size_t)idx]; hi[k] = tab[2 * (size_t)idx + 1];
    }
#pragma unroll
    for (int k = 0; k < W; k++) {
      b.lx = fminf(b.lx, lo[k].x); b.ly = fminf(b.ly, lo[k].y); b.lz = fminf(b.lz, lo[k].z);
      b.hx = fmaxf(b.hx, hi[k].x); b.hy = fmaxf(b.hy, hi[k].y); b.hz = fmaxf(b.hz, hi[k].z);
    }
  }
}
__device__ __forceinline__ Box6 range_box(const TreeScratch& t, int a, int e) {
  Box6 b = box_empty();
  if (e - a <= 64) {
    box_merge_run(b, t.lbox, a, e);
    return b;
  }
  int a1 = (a + 31) & ~31, e1 = e & ~31;
  box_merge_run(b, t.lbox, a, a1);
  box_merge_run(b, t.lbox, e1, e);
  int c0 = a1 >> 5, c1 = e1 >> 5;
  if (c1 - c0 <= 64) {
    box_merge_run(b, t.a1box, c0, c1);
    return b;
  }
  int c0a = (c0 + 31) & ~31, c1a = c1 & ~31;
  box_merge_run(b, t.a1box, c0, c0a);
  box_merge_run(b, t.a1box, c1a, c1);
  box_merge_run(b, t.a2box, c0a >> 5, c1a >> 5);
  return b;
}
__global__ void __launch_bounds__(256) k_radix_b(TreeScratch t) {
  const int n_leaves = (int)t.lid[t.total - 1];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_leaves - 1) return;
  int left, right, lo, hi, delta;
  radix_node(t.lkey, n_leaves, i, left, right, lo, hi, &delta);
  t.icom[i] = key_common(delta);
  t.ichild[2 * i] = left;
  t.ichild[2 * i + 1] = right;
  t.irange[2 * i] = lo;
  t.irange[2 * i + 1] = hi;
  if (left >= 0) t.iparent[left] = i;    // (k_nodex_b climbs these to find a node's depth)
  if (right >= 0) t.iparent[right] = i;
  if ((t.lkey[lo] >> 32) == (t.lkey[hi] >> 32)) box_store(t.ibox, i, range_box(t, lo, hi + 1));   // (a node that joins two clouds has no box)
}
// --- 4-ary nodes: every binary node of a cloud adopts its grandchildren (a leaf child stays a child) -------------------
__global__ void __launch_bounds__(256) k_nodex_b(const IndexDesc* __restrict__ descs, TreeScratch t) {
  const int n_leaves = (int)t.lid[t.total - 1];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // (no early return before the start grid's entries are made: a long run of table cells is written by the whole wave)
  bool valid = i < n_leaves - 1;
  int lo = 0, hi = 0, cloud = 0;
  if (valid) {
    lo = t.irange[2 * i]; hi = t.irange[2 * i + 1];
    cloud = (int)(t.lkey[lo] >> 32);
    valid = cloud == (int)(t.lkey[hi] >> 32);   // a node that joins two clouds is not part of any cloud's tree
  }
  IndexDesc d = descs[0];
  int a_c = 0, b_c = 0, com_i = 0, lc = -1, rc = -1, com_l = 30, com_r = 30;
  bool is_root = false, grid_on = false;
  if (valid) {
    d = descs[cloud];
    a_c = (int)t.lid[d.offset] - 1; b_c = (int)t.lid[d.offset + d.n - 1] - 1;
    is_root = lo == a_c && hi == b_c;
    com_i = t.icom[i];
    grid_on = d.n >= GRID_MIN_POINTS;
    lc = t.ichild[2 * i]; rc = t.ichild[2 * i + 1];
    com_l = lc >= 0 ? t.icom[lc] : 30; com_r = rc >= 0 ? t.icom[rc] : 30;
  }
  auto child_ref = [&](int c) -> int32_t {   // the reference the walk uses: cloud-local node index / leaf reference
    return c < 0 ? leaf_ref(t.lstart[~c] - (uint32_t)d.offset, (int)(t.lstart[~c + 1] - t.lstart[~c])) : c - a_c;
  };
  {   // the start grid's entries (grid_child_cells, lh_device.hpp): every binary node offers its two children.  A child's cells are
      // written by its own thread when they are few (nearly always: one); a leaf in a large empty region -- an outlier that stretched
      // the key grid can own thousands of cells -- is entered by the whole wave (one thread doing it made a 100 k-point build with three
      // 4-km outliers take 0.52 instead of 0.16 ms)
    int32_t* grid = reinterpret_cast<int32_t*>(d.hdr + 1);
    const bool offer = valid && grid_on && com_i < 3 * GRID_FINEST;   // (a node inside a cell of the finest table offers nothing)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int c = side ? rc : lc;
      uint32_t key = 0; int32_t ref = 0;
      if (offer) { key = (uint32_t)t.lkey[c < 0 ? ~c : t.irange[2 * c]] & 0x3fffffffu; ref = child_ref(c); }
#pragma unroll
      for (int l = GRID_FINEST; l >= 3; l--) {
        uint32_t lo3[3] = {0, 0, 0}, hi3[3] = {0, 0, 0};
        const bool app = offer && grid_child_cells(l, com_i, c < 0, side ? com_r : com_l, key, lo3, hi3);
        const uint32_t nx = hi3[0] - lo3[0] + 1, ny = hi3[1] - lo3[1] + 1, nz = hi3[2] - lo3[2] + 1, cnt = nx * ny * nz;
        if (app && cnt <= 8)
          for (uint32_t x = lo3[0]; x <= hi3[0]; x++)
            for (uint32_t y = lo3[1]; y <= hi3[1]; y++)
              for (uint32_t z = lo3[2]; z <= hi3[2]; z++) grid[grid_index(l, (int)x, (int)y, (int)z)] = ref;
        unsigned long long big = __ballot(app && cnt > 8);
        while (big) {   // (wave-uniform loop)
          const int src = __ffsll((long long)big) - 1;
          big &= big - 1;
          const uint32_t bx = (uint32_t)__shfl((int)lo3[0], src, 64), by = (uint32_t)__shfl((int)lo3[1], src, 64), bz = (uint32_t)__shfl((int)lo3[2], src, 64);
          const uint32_t by_n = (uint32_t)__shfl((int)ny, src, 64), bz_n = (uint32_t)__shfl((int)nz, src, 64), bcnt = (uint32_t)__shfl((int)cnt, src, 64);
          const int32_t bref = __shfl(ref, src, 64);
          const unsigned long long gp = (unsigned long long)(uintptr_t)grid;
          int32_t* bgrid = reinterpret_cast<int32_t*>((uintptr_t)(((unsigned long long)(uint32_t)__shfl((int)(gp >> 32), src, 64) << 32) |
                                                                  (uint32_t)__shfl((int)(uint32_t)gp, src, 64)));
          for (uint32_t k = (uint32_t)lane; k < bcnt; k += 64u) {
            const uint32_t z = k % bz_n, y = (k / bz_n) % by_n, x = k / (bz_n * by_n);
            bgrid[grid_index(l, (int)(bx + x), (int)(by + y), (int)(bz + z))] = bref;
          }
        }
      }
    }
    if (valid && grid_on && is_root) grid_fill_root(grid, com_i, (uint32_t)t.lkey[lo] & 0x3fffffffu, i - a_c);
  }
  if (!valid) return;
  // A 4-ary node adopts its GRANDchildren, so the walk only ever reaches the binary nodes at even depth below the cloud's root -- or
  // below a CELL ROOT of the start grid (a node that lies inside a table cell while its parent does not: walks start there too, so
  // it is a 4-ary node whatever its depth, and its parent keeps it as a child instead of adopting its children).  The other nodes
  // would be built (a 64-byte store each) and never read.  The distance to the nearest such ancestor comes from climbing the
  // parent links: ~8 dependent reads.
  {
    int depth = 0, j = i, cj = com_i;
    while (!(t.irange[2 * j] == a_c && t.irange[2 * j + 1] == b_c)) {
      const int pj = t.iparent[j], cp = t.icom[pj];
      if (grid_on && grid_depth(cj) > grid_depth(cp)) break;   // j is a cell root
      j = pj; cj = cp; depth++;
    }
    if (depth & 1) return;
  }
  const TreeHeader fr = *d.hdr;   // the quantisation frame, written by k_key_b (an earlier launch)
  // the (<= 4) children and their boxes: a leaf's from the leaf table, an internal node's from the node table k_radix_b left
  int cref[4] = {0, 0, 0, 0}, cnt = 0;
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int c = side ? rc : lc;
    const bool cell_root = c >= 0 && grid_on && grid_depth(side ? com_r : com_l) > grid_depth(com_i);
    if (c < 0 || cell_root) cref[cnt++] = c;
    else { cref[cnt++] = t.ichild[2 * c]; cref[cnt++] = t.ichild[2 * c + 1]; }
  }
  NodeX nd;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    nd.lo_xy[k] = 0xffffffffu; nd.hi_xy[k] = 0u; nd.z_lohi[k] = 0xffffffffu;
    nd.child[k] = NO_CHILD;
  }
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (k < cnt) {
      const int ref = cref[k];
      const float4* tab = ref < 0 ? t.lbox : t.ibox;
      const int idx = ref < 0 ? ~ref : ref;
      const float4 blo = tab[2 * (size_t)idx], bhi = tab[2 * (size_t)idx + 1];
      quant_box(fr, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k]);
      nd.child[k] = child_ref(ref);
    }
  d.nodes[i - a_c] = nd;
  if (is_root) {
    d.hdr->root = i - a_c;
    d.hdr->n_leaves = b_c - a_c + 1;
  }
}

void launch_index_bbox_init(uint32_t* bbox, hipStream_t s) {   // once, when the slots are allocated; afterwards every build resets them for the next
  hipLaunchKernelGGL(k_bbox_init_b, dim3((MAX_INDEX_BATCH * 8 + 255) / 256), dim3(256), 0, s, bbox, MAX_INDEX_BATCH);
}
void launch_index_keys(const IndexDesc* descs, int n_clouds, int max_n, uint32_t* bbox, uint64_t* pairs, uint64_t* keys, uint32_t* vals, hipStream_t s) {
  // (the bounding-box slots were reset by the previous build's k_leaves_b, or by the allocation)
  int blocks = (max_n + 255) / 256;
  hipLaunchKernelGGL(k_bbox_b, dim3(blocks > 48 ? 48 : blocks, n_clouds), dim3(256), 0, s, descs, bbox);   // (every workgroup ends in six atomics on its cloud's slots: few, fat workgroups)
  hipLaunchKernelGGL(k_key_b, dim3(blocks, n_clouds), dim3(256), 0, s, descs, bbox, reinterpret_cast<uint2*>(pairs), keys, vals);
}
void launch_index_leaves(const IndexDesc* descs, int n_clouds, const TreeScratch& t, const uint32_t* vals_sorted, uint32_t* bbox, hipStream_t s) {
  const int tiles = (t.total + LEAF_TILE - 1) / LEAF_TILE;
  hipLaunchKernelGGL(k_leafcell_b, dim3((t.total + 255) / 256), dim3(256), 0, s, t);
  hipLaunchKernelGGL(k_leaves_b, dim3(tiles), dim3(256), 0, s, descs, t, vals_sorted, bbox);   // (forms its own tile offset: no scan launch in between)
  (void)n_clouds;
}
void launch_index_trees(const IndexDesc* descs, int n_clouds, int max_n, const TreeScratch& t, hipStream_t s, int stage) {
  int blocks = (t.total + 255) / 256;  // upper bound of the leaf count; the kernels read the real one from lid[total-1]
  if (stage == 0) {        // box tables: per leaf, per 32 leaves, per 1024 leaves -- one launch, 1 024 leaves per workgroup (leaves <= points)
    const int tiles = (t.total + LEAF_TILE - 1) / LEAF_TILE;
    const int nb = (t.total + 1023) / 1024, nz = (tiles + 255) / 256;
    hipLaunchKernelGGL(k_boxes_b, dim3(nb > nz ? nb : nz), dim3(256), 0, s, descs, t, tiles);
  } else if (stage == 1) { // hierarchy + one box per binary node
    hipLaunchKernelGGL(k_radix_b, dim3(blocks), dim3(256), 0, s, t);
  } else {                 // 4-ary nodes + headers
    hipLaunchKernelGGL(k_nodex_b, dim3(blocks), dim3(256), 0, s, descs, t);
  }
}

// ===== K4: NN + Mahalanobis sweep ==========================================================================

// K2': seeds for a cold sweep.  One thread per group of SEED_GROUP consecutive source points descends the tree to the leaf nearest to
// the group's first point and hands that leaf's nearest point to the whole group as warm-start candidate (any target point is a
// valid candidate, so exactness is untouched; consecutive lidar returns are spatial neighbours, so the bound is tight).  The descent
// starts at the query's own cell of the start grid (lh_device.hpp) and ONLY the candidates are written: the sweep that follows is
// launched cold (SweepJob::pad) and reads neither certificates nor neighbour records.  Per 32 pairs: an exact search per seed (groups
// of 8) 220 us; the descent from the root with groups of 4, certificates and records written, 120 us; now 39-48 us.
__global__ void __launch_bounds__(256) k_seed(const PairDesc* __restrict__ descs, SweepArgs a) {
  int jb, blk;
  if (!xcd_job_map(a.njobs, a.bpj, jb, blk)) return;
  const SweepJob& job = a.job[jb];  // (a cold pair's transformation_ is the identity in both loop flavours: job.T carries it)
  const PairDesc d = descs[job.slot];
  const int group = a.pad;  // seed group size (runtime so it can be tuned)
  int g = blk * 256 + threadIdx.x;
  int i = g * group;
  if (i >= d.n) return;
  float4 p = d.src[i];
  float qx, qy, qz;
  xform_pt(job.T, p.x, p.y, p.z, qx, qy, qz);
  TreeView tv{d.tgt_sorted, d.tgt_nodes, d.tgt_hdr, d.m};
  Nn1Collector col{INFINITY, 0x7fffffff};
  tree_descend<Nn1Collector, true>(tv, qx, qy, qz, col);   // the nearest point of the nearest leaf (below the query's own grid cell): no stack, no backtracking, a third of an exact cold search
  const int j = (col.bi == 0x7fffffff) ? -1 : col.bi;
  // only the candidate: the sweep that follows is launched `cold` (SweepJob::pad = the group size) and reads neither certificate nor record.
  // The whole group gets j; the cold sweep itself lets point e of the group choose between j and j + e (sweep_point).
  for (int e = 0; e < group; e++)
    if (i + e < d.n) d.prev_nn[i + e] = j;
}

static int seed_group_size() {
  static const int seed_group = []() { const char* e = getenv("LH_SEED_GROUP"); int v = e ? atoi(e) : SEED_GROUP; return v < 1 ? 1 : v; }();
  return seed_group;
}
void launch_seed(const PairDesc* descs, SweepArgs& a, int max_n, hipStream_t s) {
  const int seed_group = seed_group_size();
  a.pad = seed_group;
  int groups = (max_n + seed_group - 1) / seed_group;
  a.bpj = (groups + 255) / 256;
  hipLaunchKernelGGL(k_seed, dim3(xcd_grid(a.njobs, a.bpj)), dim3(256), 0, s, descs, a);
}

// transformation_ of a job as 12 row-major floats in SCALAR registers: from the launch arguments (host-driven loop: the host
// solved the last iteration) or from the pair's device state (device-driven loop: k_solve did).  The state is read through a
// uniform address and pinned to SGPRs with readfirstlane -- twelve VGPRs would cost the fused sweep its 6 waves per SIMD.
// Returns false when the pair's loop has already ended (the launch covers it only because the host has not looked yet).
__device__ __forceinline__ bool job_transform(const SweepJob& job, const OuterState* __restrict__ states, float* T) {
  if (!states) {
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = job.T[k];
    return true;
  }
  // The state was written by an EARLIER launch (k_solve) and is read-only here, at a wave-uniform address: read it through the
  // constant address space so that the loads are scalar (s_load_dword*: SGPR results, their own counter -- the wave's vector
  // loads of the point data are neither delayed by them nor waited for with them).
  typedef const __attribute__((address_space(4))) OuterState* ConstState;
  ConstState st = (ConstState)(uintptr_t)(states + job.slot);
  if (st->done) return false;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) T[r * 4 + c] = st->T[c * 4 + r];
  return true;
}

// The neighbour a search ended with, or -1 for "none": a query finds no neighbour when it is not finite (NaN: nothing compares
// closer; Inf: every distance is +inf) -- pcl::KdTreeFLANN's point_representation_->isValid(query) -- and the reference then
// sets `failure` and gives the whole alignment up (gicp.hpp:471-478, 504-506).  Such a point adds NO_NN_MARK to the
// correspondence count of its sweep (every other sum untouched): integers stay exact in a double, the mark survives every
// fixed-order reduction and the SUM hook of the source-sharded pair, and the solver (lh_bfgs.hpp) sees it with the first count.
__device__ __forceinline__ int nn_index(int bi, float bd) { return (bi == 0x7fffffff || !(bd < INFINITY)) ? -1 : bi; }

// one source point of a sweep: exact NN (warm start + certificate) and, if gated in, M = (R C1 R^T + C2)^-1
struct SweepPoint {
  float4 p;      // source point
  float4 tgt;    // matched target point
  double M[9];
  int j;         // target index or -1
  bool matched;
  bool searched; // the certificate did not cover this query: the tree was walked
  bool nonn;     // the query has NO nearest neighbour (a non-finite query point): searchForNeighbors false, gicp.hpp:471-478
};
// kRank1 (the fused sweep of cost_mode 1, both covariances from normals): C = I - (1-eps) n n^T / |n|^2 is a rank-one update
// of the identity, so   R C1 R^T + C2 = (R R^T + I) - k1 u u^T - k2 v v^T,   u = R n1, v = n2, k = (1-eps) / |n|^2
// -- the same matrix in exact arithmetic for ANY R (float-rounded rotations included), formed with 60 instead of 150 double
// operations, no square roots, and inverted as a symmetric 3x3 (six cofactors).  Its rounding differs from the reference's
// order of operations (normalise, two 3x3x3 products) in the last bits of M, which is why only cost_mode 1 -- a bit-different
// evaluation of the cost anyway -- uses it; k_sweep (cost_mode 0, the debug entry points) keeps the reference order.
// `cold` (wave-uniform: the pair's FIRST sweep, after the seed pass): prev_nn holds a seed -- any target point, a bound and nothing more --
// while the certificate and the neighbour record of this workspace slot are a former pair's: neither is read, every point searches,
// and the record is (re)written whatever the search finds.  (The seed pass used to write a zero certificate and the record for every
// point: 52 B per source point of stores, 110 us per 32 pairs, for one sweep's use.)
// The neighbour record of a source point -- position and normal of target point prev_nn[i] -- as two planes of packed 12-byte triples
// (PairDesc::rec: positions at rec, normals at rec + 3 n_pad): a wave's load is 768 contiguous bytes, and the late sweeps, which stream the
// record of every point once per iteration and are bound by exactly that stream, read 24 bytes per point instead of two float4.
struct __attribute__((packed, aligned(4))) Pk3 { float x, y, z; };
__device__ __forceinline__ float4 rec_pos(const PairDesc& d, int i) { const Pk3 v = gld(reinterpret_cast<const Pk3*>(d.rec) + i); return make_float4(v.x, v.y, v.z, 0.f); }
__device__ __forceinline__ float4 rec_nrm(const PairDesc& d, int i) { const Pk3 v = gld(reinterpret_cast<const Pk3*>(d.rec + 3 * (size_t)d.n_pad) + i); return make_float4(v.x, v.y, v.z, 0.f); }
__device__ __forceinline__ void rec_put(const PairDesc& d, int i, const float4& t, const float4& tn) {
  gst(reinterpret_cast<Pk3*>(d.rec) + i, Pk3{t.x, t.y, t.z});
  gst(reinterpret_cast<Pk3*>(d.rec + 3 * (size_t)d.n_pad) + i, Pk3{tn.x, tn.y, tn.z});
}
template <bool kRank1 = false, int kStride = 256>
__device__ __forceinline__ void sweep_point(const PairDesc& d, const float* __restrict__ T, int i, uint64_t* stack, SweepPoint& o, float cm, int cold = 0) {
  // first round of loads: everything whose address only depends on i goes out together (the certificate and, in the fused
  // kernel, the source normal as well: each was its own dependent memory round behind the candidate before, and the late
  // sweeps are bound by exactly that chain -- a workgroup lives for two memory latencies instead of three)
  o.p = gld(d.src + i);
  int w = gld(d.prev_nn + i);
  // `cold` = the seed pass's group size (the pair's FIRST sweep).  Point e of a group may start from target point w + e instead of the group's
  // shared seed w (round 5): source and target are scans of the same kind of sensor, consecutive returns of a ring are consecutive in both, so
  // the neighbour of the e-th next source point is near the e-th next target point.  ANY target point is a valid candidate -- no result depends
  // on it, only how tight the cold walk's first bound is -- and the nearer of the two is taken, so a target in some other order (a voxelised
  // cloud, a map) loses nothing.  Host model (tools/model/seed_model.cpp, bench pair seed 10): groups of 16 with w + e walk as little (17.3 wave
  // iterations per 64 queries) as groups of 4 sharing w (17.8; groups of 16 sharing w: 19.2) -- a quarter of the seed pass's descents.
  int w_alt = -1;
  float4 t_alt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cold > 1 && w >= 0) {
    const int e = i % cold;
    if (e > 0) {
      w_alt = min(d.m - 1, w + e);
      t_alt = gld(d.tgt_xyz + w_alt);
    }
  }
  float4 cq = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!cold) cq = gld(d.cert + i);   // only meaningful when w >= 0; the buffer always holds n entries
  float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f), tn = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (kRank1) {
    nn = gld(d.src_nrm + i);
    // ... and so does the candidate itself: rec[i] holds the position and normal of target point prev_nn[i] (kept in step with
    // prev_nn by every writer), so the gather through w -- a second, dependent memory round -- is gone from every sweep
    if (!cold) {
      t = rec_pos(d, i);
      tn = rec_nrm(d, i);
    } else if (w >= 0)
      t = gld(d.tgt_xyz + w);   // the seed: its position only (it is a bound, not yet a neighbour)
  }
  float qx, qy, qz;
  xform_pt(T, o.p.x, o.p.y, o.p.z, qx, qy, qz);  // gicp.hpp:469
  TreeView tv{d.tgt_sorted, d.tgt_nodes, d.tgt_hdr, d.m};
  Nn1CertCollector col{INFINITY, 0x7fffffff, INFINITY};
  bool need_search = true;
  if (w_alt >= 0) {   // cold sweep: the nearer of the group's seed and its e-th successor (ties: the seed)
    const float4 t0 = kRank1 ? t : gld(d.tgt_xyz + w);
    if (d2f(qx, qy, qz, t_alt.x, t_alt.y, t_alt.z) < d2f(qx, qy, qz, t0.x, t0.y, t0.z)) {
      w = w_alt;
      if constexpr (kRank1) t = t_alt;
    }
  }
  if (w >= 0) {
    // warm start: last sweep's neighbour is a valid candidate => tight initial bound, still exact
    if constexpr (!kRank1) {
      t = gld(d.tgt_xyz + w);
      // its normal is fetched in the same round: when the certificate holds (every late sweep) the neighbour is w and the
      // normal would otherwise be a third dependent memory level; after a walk both are re-read, so neither stays live across it
      if (d.tgt_nrm) tn = gld(d.tgt_nrm + w);
    }
    col.bd = d2f(qx, qy, qz, t.x, t.y, t.z);
    col.bi = w;
    // certificate from the last full search at query position cq: every other target point was at squared distance
    // >= cq.w from cq, so it is at distance >= sqrt(cq.w) - |q - cq| from q.  If the candidate is strictly closer
    // (1e-5 relative margin >> float rounding of the d2 evaluations), the traversal cannot change the result.
    // (float arithmetic: the raw hardware square root is good to 1 ulp ~ 1e-7 relative, two orders below the 1e-5 margins)
    float e = sqrt_bound(d2f(qx, qy, qz, cq.x, cq.y, cq.z));
    float dw = sqrt_bound(col.bd), lo = sqrt_bound(cq.w);
    if (!cold && dw * (1.0f + cm) + e * (1.0f + cm) + 1e-12f < lo * (1.0f - cm)) need_search = false;
  }
  o.searched = need_search;
  if (need_search) {
    tree_search<Nn1CertCollector, true>(tv, qx, qy, qz, col, stack, kStride);
    // (a search that found nothing -- a non-finite query -- leaves a certificate that can never hold: k_late tests it without reading prev_nn)
    gst(d.cert + i, make_float4(qx, qy, qz, nn_index(col.bi, col.bd) >= 0 ? col.lb : -1.0f));
    if (d.stats) atomicAdd(&d.stats[0], 1ull);  // instrumentation only (lh_gicp_debug_sweep): contended atomics
  }
  if (d.stats) atomicAdd(&d.stats[1], 1ull);
  int j = nn_index(col.bi, col.bd);
  o.nonn = j < 0;
  if (j != w || cold) gst(d.prev_nn + i, j);   // (cold: w may be the group seed's successor, not what prev_nn holds)
  if (need_search && j >= 0) {
    t = gld(d.tgt_xyz + j);
    if (d.tgt_nrm) tn = gld(d.tgt_nrm + j);
    if constexpr (kRank1) nn = gld(d.src_nrm + i);   // re-read after a walk, so that the normal is not live across it
    if ((j != w || cold) && d.rec) rec_put(d, i, t, tn);   // the record follows prev_nn
  }
  o.j = j;
  o.matched = j >= 0 && (double)col.bd < d.corr_dist2;  // gicp.hpp:483
  if (o.matched) {
    // transform_R = double(transformation_) * double(guess), top-left 3x3 (gicp.hpp:450-460); the k = 3 term is T(i,3)*0
    double R[9];
    if (d.guess_identity) {  // T * I: every product with an off-diagonal 0 vanishes and x * 1.0 + 0.0 = x exactly
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R[r * 3 + cc] = (double)T[r * 4 + cc];
    } else {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
          R[r * 3 + cc] = (((double)T[r * 4 + 0] * d.guess3[0 * 3 + cc] + (double)T[r * 4 + 1] * d.guess3[1 * 3 + cc]) +
                           (double)T[r * 4 + 2] * d.guess3[2 * 3 + cc]) + (double)T[r * 4 + 3] * 0.0;
    }
    if constexpr (kRank1) {  // (the launch guarantees that neither cloud of any job carries k-NN covariances)
      double G[6], M6[6];
      {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int cc = r; cc < 3; cc++) {   // R R^T + I, like pose_of_transform (R = double(T) when the guess is the identity)
            const double rr = __builtin_fma(R[r * 3 + 2], R[cc * 3 + 2], __builtin_fma(R[r * 3 + 1], R[cc * 3 + 1], R[r * 3 + 0] * R[cc * 3 + 0]));
            G[q++] = r == cc ? rr + 1.0 : rr;
          }
      }
      maha_rank1(R, 3, G, 1.0 - d.gicp_eps, nn, tn, M6);
      o.M[0] = M6[0]; o.M[1] = M6[1]; o.M[2] = M6[2];
      o.M[3] = M6[1]; o.M[4] = M6[3]; o.M[5] = M6[4];
      o.M[6] = M6[2]; o.M[7] = M6[4]; o.M[8] = M6[5];
      o.tgt = t;
      return;
    }
    double C1[9], C2[9];
    if (d.src_cov6) {
      double s6[6];
#pragma unroll
      for (int k = 0; k < 6; k++) s6[k] = gld(d.src_cov6 + (size_t)k * d.src_cov_pad + i);
      sym6_to_mat9(s6, C1);
    } else {
      float4 nn = gld(d.src_nrm + i);
      cov_from_normal(nn.x, nn.y, nn.z, d.gicp_eps, C1);
    }
    if (d.tgt_cov6) {
      double s6[6];
#pragma unroll
      for (int k = 0; k < 6; k++) s6[k] = gld(d.tgt_cov6 + (size_t)k * d.m_pad + j);
      sym6_to_mat9(s6, C2);
    } else {
      cov_from_normal(tn.x, tn.y, tn.z, d.gicp_eps, C2);
    }
    mahalanobis(R, C1, C2, o.M);  // gicp.hpp:488-493
    o.tgt = t;
  }
}

__global__ void __launch_bounds__(256) k_sweep(const PairDesc* __restrict__ descs, SweepArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_stack[];  // [entries][256]
  int jb, blk;
  if (!xcd_job_map(a.njobs, a.bpj, jb, blk)) return;
  const SweepJob& job = a.job[jb];
  const PairDesc d = descs[job.slot];
  int i = blk * 256 + threadIdx.x;
  if (i >= d.n) return;
  SweepPoint sp;
  sweep_point(d, job.T, i, lds_stack + threadIdx.x, sp, a.cert_rel, job.pad);
  float4 c = make_float4(0.f, 0.f, 0.f, __int_as_float(sp.nonn ? -2 : -1));   // -1: gated out; -2: no neighbour at all (the failure path)
  if (sp.matched) {
    d.maha6[(size_t)0 * d.n_pad + i] = sp.M[0];
    d.maha6[(size_t)1 * d.n_pad + i] = sp.M[1];
    d.maha6[(size_t)2 * d.n_pad + i] = sp.M[2];
    d.maha6[(size_t)3 * d.n_pad + i] = sp.M[4];
    d.maha6[(size_t)4 * d.n_pad + i] = sp.M[5];
    d.maha6[(size_t)5 * d.n_pad + i] = sp.M[8];
    c = make_float4(sp.tgt.x, sp.tgt.y, sp.tgt.z, __int_as_float(sp.j));
  }
  d.corr[i] = c;
}

// K4 + K5' fused (cost_mode 1): the Mahalanobis matrix and the correspondence never leave registers -- each workgroup
// reduces the 74 second-order moments of its 256 points through LDS (the traversal stack's region is reused) and writes
// one 74-double partial.  Late GICP iterations, where certificates skip the tree walk, become a pure stream of the
// source cloud: ~90 B/point instead of 152 B (sweep) + 80 B (moments pass).
__device__ __forceinline__ double mom_value(int k, const double* M6, const double* Ma, double aMa, const double* pt, const double* pp) {
  if (k == 0) return aMa;
  if (k < 13) return Ma[(k - 1) >> 2] * pt[(k - 1) & 3];
  if (k < 73) return M6[(k - 13) / 10] * pp[(k - 13) % 10];
  return 1.0;
}
// the moment contributions of one matched point, added to acc[0..73] (acc must be zero-initialised by the caller)
__device__ __forceinline__ void moments_of_point(const float* __restrict__ T, const SweepPoint& sp, double* M6, double* Ma, double& aMa, double* pt, double* pp) {
  pt[0] = (double)sp.p.x; pt[1] = (double)sp.p.y; pt[2] = (double)sp.p.z; pt[3] = 1.0;
  double T0[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T0[k] = (double)T[k];
  const double m6[6] = {sp.M[0], sp.M[1], sp.M[2], sp.M[4], sp.M[5], sp.M[8]};
  const double p3[3] = {pt[0], pt[1], pt[2]};
  double ma[3];
  resid_terms(T0, m6, p3, sp.tgt, ma, aMa);
#pragma unroll
  for (int k = 0; k < 6; k++) M6[k] = m6[k];
  Ma[0] = ma[0]; Ma[1] = ma[1]; Ma[2] = ma[2];
  int t = 0;
#pragma unroll
  for (int cc = 0; cc < 4; cc++)
#pragma unroll
    for (int ee = cc; ee < 4; ee++) pp[t++] = pt[cc] * pt[ee];
}

// Per-wave reduction of 74 per-lane values, no workgroup barriers inside the loop: after ONE barrier (all lanes are done
// with their traversal stacks) each wave owns a private 4-KB slice [8 values][64 lanes] of the LDS region; a wave's DS
// operations execute in order, so write -> fence -> read needs no s_barrier.  64 lanes -> 8 strided partial sums -> 3
// shuffles.  `value(k)` yields the lane's k-th of NV values; the row (ROW doubles) is completed with `extra` and zeros.
// sum of a double with its neighbour lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: two v_mov_dpp + one v_add_f64, no LDS
__device__ __forceinline__ double pair_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
template <int NV, int ROW, class F>
__device__ __forceinline__ void wave_reduce_row(uint64_t* lds_base, double* out, double extra, F value) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Lane pairs are added in registers first (DPP), so only the even lanes go through LDS: 32 entries per value.  Row stride
  // 40 doubles (not 32): the 16 lanes the LDS serves per cycle read two 64-B pieces of two DIFFERENT rows, and 320-B rows put
  // those on complementary bank halves (with power-of-two rows they collided: SQ_LDS_BANK_CONFLICT was 26 % of the LDS-active
  // cycles).  A second DPP step (quads, 16 entries per value) measured slightly slower: the VALU is the tighter resource then.
  constexpr int RS = 40;
  double* wst = reinterpret_cast<double*>(lds_base) + wave * (8 * RS);
  const int v_of = lane >> 3, part = lane & 7;
  __syncthreads();
#pragma unroll
  for (int g = 0; g < (NV + 7) / 8; g++) {
#pragma unroll
    for (int v = 0; v < 8; v++) {
      int k = g * 8 + v;
      if (k < NV) {
        double s2 = pair_sum(value(k));
        if (!(lane & 1)) wst[v * RS + (lane >> 1)] = s2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) s += wst[v_of * RS + jj * 8 + part];
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) s += __shfl_down(s, off, 8);
    if (part == 0 && g * 8 + v_of < NV) out[g * 8 + v_of] = s;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < ROW - NV) out[NV + lane] = lane == 0 ? extra : 0.0;  // the rest of the row: one extra value (e.g. tree walks), zeros
}

// ---- the 74 moments of a wave as a Gram matrix on the FP64 matrix pipe -----------------------------------------------------
// The 74 sums of a wave's 64 points are D = sum_i a_i b_i^T with
//   a_i = (M00 M01 M02 M11 | M12 M22 Ma0 Ma1 | Ma2 aMa live 0)        b_i = (x y z 1 | xx xy xz yy | yz zz 0 0)
// cut into 4x4 tiles: H = M6 x all ten columns (row tiles R0, R1 x column tiles C0, C1, C2), B = Ma x (x y z 1) (R1, R2 x C0),
// c0 = aMa * 1 and the count = live * 1 (R2 x C0): seven tiles.  v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4x4x4 blocks
// per instruction in 20 cycles (measured, tools/micro/mfma_f64_rate.hip; the 16x16x4 form takes 64 and 57 % of its 16x16 tile is
// unused here): the four blocks take four different groups of four points of the same tile, so 64 points cost 7 x 4 = 28
// instructions = 560 cycles of the matrix pipe instead of 16 x 64 = 1024.  Lane map (probed, tools/micro/mfma_f64_4x4_map.hip):
// A lane 16 k + 4 b + i = A_b[i][k], B lane 16 k + 4 b + j = B_b[k][j], D lane 16 i + 4 b + j = D_b[i][j].
// Operands go through LDS once (lane = point -> lane = (k, block, element)), half a wave at a time: 32 rows of GRAM_RS doubles.
// A lane stages a (11 values), its point (x, y, z) and a 1.0; the reading lane forms the products of b itself -- the same
// products the writer would have made, without ten more doubles per lane in registers and in LDS.  Wave-synchronous: no barrier.
struct GramAcc { double t[7]; };   // R0C0 R0C1 R0C2 R1C0 R1C1 R1C2 R2C0, summed over the block's point groups
constexpr int GRAM_RS = 15;        // doubles per staged point; odd => conflict-free column writes
__device__ __forceinline__ void gram_zero(GramAcc& g) {
#pragma unroll
  for (int k = 0; k < 7; k++) g.t[k] = 0.0;
}
// `on` = the lane has a matched point (otherwise it stages zeros)
__device__ __forceinline__ void gram_accumulate(double* wl, const double (&av)[11], const float4& p, bool on, GramAcc& g, bool mark = false) {
  const int lane = threadIdx.x & 63;
  const int kk = lane >> 4, b = (lane >> 2) & 3, ij = lane & 3;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if ((lane >> 5) == half) {
      double* row = wl + (lane & 31) * GRAM_RS;
#pragma unroll
      for (int e = 0; e < 11; e++) row[e] = av[e];
      row[11] = on ? (double)p.x : 0.0;
      row[12] = on ? (double)p.y : 0.0;
      row[13] = on ? (double)p.z : 0.0;
      row[14] = (on || mark) ? 1.0 : 0.0;   // mark: a no-neighbour lane -- av[10] = NO_NN_MARK reaches the count, its (maybe NaN) point nothing
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 2; s++) {  // sixteen points per step: block b takes points 16 s + 4 b .. + 3
      const double* row = wl + (16 * s + 4 * b + kk) * GRAM_RS;
      const double A0 = row[ij], A1 = row[4 + ij], A2 = (ij < 3) ? row[8 + ij] : 0.0;
      // the lane's operands of b are fetched from the staged row at per-lane offsets (x y z 1 sit at 11..14) rather than picked out of
      // all four by compare-and-select: sixteen v_cndmask per step were a seventh of the late sweep's vector instructions
      const double u = row[11 + ij];                                       // C0: x y z 1
      const double B1 = row[ij == 3 ? 12 : 11] * row[ij == 3 ? 12 : 11 + ij];   // C1: xx xy xz yy
      const double yz = row[ij == 1 ? 13 : 12] * row[13];
      const double B2 = ij < 2 ? yz : 0.0;                                 // C2: yz zz 0 0
      g.t[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, u, g.t[0], 0, 0, 0);
      g.t[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, B1, g.t[1], 0, 0, 0);
      g.t[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, B2, g.t[2], 0, 0, 0);
      g.t[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, u, g.t[3], 0, 0, 0);
      g.t[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, B1, g.t[4], 0, 0, 0);
      g.t[5] = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, B2, g.t[5], 0, 0, 0);
      g.t[6] = __builtin_amdgcn_mfma_f64_4x4x4f64(A2, u, g.t[6], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}
// the four blocks' tiles are added in a fixed tree ((b0 + b1) + (b2 + b3)) and written to the wave's 76-double row:
// S[0] = c0, S[1 + 4 r + c] = B[r][c], S[13 + 10 m + q] = H[m][q] (q: xx xy xz x yy yz y zz z 1), S[73] = count, S[74] = walks
__device__ __forceinline__ void gram_store(const GramAcc& g, double* out, double walks) {
  const int lane = threadIdx.x & 63, i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3;
  double v[7];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    double s = g.t[k];
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    v[k] = s;
  }
  if (b == 0) {
    const int q0 = j == 0 ? 3 : (j == 1 ? 6 : (j == 2 ? 8 : 9));   // C0: x y z 1
    const int q1 = j == 3 ? 4 : j;                                  // C1: xx xy xz yy
    const int q2 = j == 0 ? 5 : 7;                                  // C2: yz zz
    // R0: M00 M01 M02 M11 = M6[i]
    out[13 + 10 * i + q0] = v[0];
    out[13 + 10 * i + q1] = v[1];
    if (j < 2) out[13 + 10 * i + q2] = v[2];
    // R1: M12 M22 Ma0 Ma1
    if (i < 2) {
      out[13 + 10 * (4 + i) + q0] = v[3];
      out[13 + 10 * (4 + i) + q1] = v[4];
      if (j < 2) out[13 + 10 * (4 + i) + q2] = v[5];
    } else {
      out[1 + 4 * (i - 2) + j] = v[3];
    }
    // R2: Ma2 aMa live 0
    if (i == 0) out[1 + 8 + j] = v[6];
    if (i == 1 && j == 3) out[0] = v[6];
    if (i == 2 && j == 3) out[73] = v[6];
  }
  if (lane < MOM_ROW - MOM_NSUM) out[MOM_NSUM + lane] = lane == 0 ? walks : 0.0;
}

// One row per WORKGROUP: every wave leaves its 76-double row in its own (now free) staging region, and after a barrier -- at the very
// end, the waves have nothing left to do -- 76 threads add the four in a fixed tree.  A row per wave was 9.5 bytes of stores per
// source point and made the final sum read 34 MB per 32-pair launch (10 us of every iteration).
__device__ __forceinline__ void wg_row_sum(const double* lds_rows, int wave_stride, double* __restrict__ out) {
  __syncthreads();
  const int t = threadIdx.x;
  if (t < MOM_ROW) out[t] = (lds_rows[t] + lds_rows[wave_stride + t]) + (lds_rows[2 * wave_stride + t] + lds_rows[3 * wave_stride + t]);
}

// one point per thread (78 VGPRs, 6 waves per SIMD).  A variant with 4 points per thread and the 74 moments accumulated in
// registers (239 VGPRs, 2 waves per SIMD) pays the reduction once per 4 points but was slower even on certificate-only
// sweeps (142 vs 128 us): with so few waves the dependent src -> neighbour gathers are no longer hidden.
// kNormals: every job's covariances come from stored normals (the production configuration) -> rank-one Mahalanobis path
// FUSED_WG threads per workgroup.  256 (round 1-2): four waves share one LDS region and meet at a barrier before the reduction, so a
// workgroup lives as long as its slowest wave (the walks of a wave end after 15-25 steps: the other three wait, parked, holding
// their registers and LDS) and leaves one row per 256 points.  64 (default since round 3; -DLH_FUSED_WG=256 restores the other): every
// wave is its own workgroup -- no barrier, its slot is free the moment its own walks are over -- and leaves one row per 64 points.
// Measured: the three all-walk sweeps 590 / 386 / 340 -> 552 / 371 / 326 us per 32 pairs, +2.3 % scan-pairs/s.
#ifndef LH_FUSED_WG
#define LH_FUSED_WG 64
#endif
constexpr int FUSED_WG = LH_FUSED_WG;
static_assert(FUSED_WG == 256 || FUSED_WG == 64, "LH_FUSED_WG");
template <bool kNormals>
#ifndef LH_FUSED_WAVES
#define LH_FUSED_WAVES 6
#endif
__global__ void __launch_bounds__(FUSED_WG) __attribute__((amdgpu_waves_per_eu(LH_FUSED_WAVES))) k_sweep_fused(const PairDesc* __restrict__ descs, SweepArgs a, double* __restrict__ partials,
                                                     int partials_stride, const OuterState* __restrict__ states) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_stack[];  // [entries][FUSED_WG], later reused as the Gram staging rows
  int jb, blk;
  if (!xcd_job_map(a.njobs, a.bpj, jb, blk)) return;
  const SweepJob& job = a.job[jb];
  const PairDesc d = descs[job.slot];
  if (blk * FUSED_WG >= d.n) return;  // whole workgroup out of range (uniform)
  float T[12];
  if (!job_transform(job, states, T)) return;  // device-driven loop: this pair has already converged
  int i = blk * FUSED_WG + threadIdx.x;
  SweepPoint sp;
  sp.matched = false;
  sp.searched = false;
  sp.nonn = false;
  // LH_EXP_LANES (timing experiments only, results are wrong; tools/ab_lanes.sh): 1 = only every 2nd lane searches, 2 = only every 4th --
  // does a sweep's time follow the number of LANES that walk (request-bound) or the number of WAVES that do (bound per wave step)?
  // Measured (round 3): a quarter of the lanes -> 25 / 13 / 14 % less time in the three all-walk sweeps: per wave step.
  const bool exp_skip = a.pad2 != 0 && (threadIdx.x & (a.pad2 == 1 ? 1 : 3)) != 0;
  if (i < d.n && !exp_skip) sweep_point<kNormals, FUSED_WG>(d, T, i, lds_stack + threadIdx.x, sp, a.cert_rel, job.pad);
  double M6[6] = {0, 0, 0, 0, 0, 0}, Ma[3] = {0, 0, 0}, pt[4] = {0, 0, 0, 0}, pp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, aMa = 0.0;
  if (sp.matched) moments_of_point(T, sp, M6, Ma, aMa, pt, pp);
  const double live = sp.matched ? 1.0 : (sp.nonn ? NO_NN_MARK : 0.0);
  const int walks = __popcll(__ballot(sp.searched));
  double* out = partials + (size_t)job.slot * partials_stride + (size_t)blk * MOM_ROW;   // one row per workgroup
  {
    // the 74 moments of the wave on the matrix pipe (gram_accumulate above); the staging rows alias the traversal stacks
    const int wave = threadIdx.x >> 6;
    const double av[11] = {M6[0], M6[1], M6[2], M6[3], M6[4], M6[5], Ma[0], Ma[1], Ma[2], aMa, live};
    GramAcc acc;
    gram_zero(acc);
    if constexpr (FUSED_WG > 64) __syncthreads();  // every lane of the workgroup is done with its traversal stack
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    double* wl = reinterpret_cast<double*>(lds_stack) + wave * (32 * GRAM_RS);
    gram_accumulate(wl, av, sp.p, sp.matched, acc, sp.nonn);
    if constexpr (FUSED_WG > 64) {
      gram_store(acc, wl, (double)walks);
      wg_row_sum(reinterpret_cast<double*>(lds_stack), 32 * GRAM_RS, out);
    } else
      gram_store(acc, out, (double)walks);   // the wave's row straight to its place
  }
  // every job has one more row per `span` source points (k_walk's, below); a job swept by this kernel leaves them zero, so that the
  // final sum adds the same rows in the same order whichever way the sweep was launched
  if (a.span > 0) {
    const int bps = a.span / FUSED_WG;
    if (blk % bps == 0)
      for (int k = threadIdx.x; k < MOM_ROW; k += FUSED_WG)
        partials[(size_t)job.slot * partials_stride + ((size_t)((d.n + FUSED_WG - 1) / FUSED_WG) + blk / bps) * MOM_ROW + k] = 0.0;
  }
}

// ===== the all-walk sweeps as a COOPERATIVE search (round 6; cost_mode 1, covariances from normals) ===========================
// k_sweep_fused walks one query per lane: a wave lives for the LONGEST of its 64 walks and alternates node and leaf steps by vote --
// 27-35 of 64 lanes busy per vector instruction, 10.7 (warm) to 17.6 (cold) wave steps for 4.6 to 6.7 steps of work per query
// (tools/model/grid_start_model.cpp).  Here the unit of work is an ITEM (query, subtree), not a lane's whole walk, and the 256 queries
// of a workgroup share it (tools/model/item_stack_model.cpp prices the variants -- a wave-shared walk of the k-NN block-search kind
// costs 55-80 full-wave visits per 64 queries, a shared stack without the first descent explodes when the warm bound is loose):
//   phase 1  every lane: loads, transform, warm candidate / seed, certificate test (sweep_point's first half), then -- in the pair's
//            first two sweeps, where the candidate is a loose bound -- a GREEDY DESCENT from the query's own start-grid cell to one
//            leaf and a scan of it: no stack, no sort, no vote, every lane busy; the bound is tight afterwards.  Then the start items
//            of the exact search: the start cell grid_start chooses for that bound and the neighbour cells the ball still reaches.
//   phase 2  level-synchronous rounds over the workgroup's item queues in LDS: a LEAF round (every queued leaf is scanned by whichever
//            lane its position in the queue falls to; the query's (d2, index) slot takes a 64-bit atomic minimum, its certificate
//            bound a 32-bit one), then a NODE round (four child boxes against the query's bound of the moment; survivors are appended
//            to the next round's queues with one LDS atomic per wave and kind).  Every step is of ONE kind with (nearly) 64 busy lanes.
// Exactness: a subtree is dropped only when its box is farther than a real candidate's distance, a leaf is always scanned, and the
// minimum over (d2 bits << 32 | index) IS the nearest point with the lowest index -- the rule of every other search here, whatever the
// order of the visits.  The certificate bound is min(d2 of every examined point but the winner, box distance of every dropped subtree):
// each contribution is an exact value of a set that does not depend on timing (bounds only change in leaf rounds, are only read for
// pruning in node rounds, and a barrier separates the two), so neighbours, certificates and sums are bitwise reproducible.
// A queue that overflows (a non-finite query walks the whole tree; a degenerate cloud) sets a flag and the WHOLE workgroup redoes its
// points with sweep_point's per-lane walk -- the decision depends only on item counts, which are deterministic too.
// MEASURED (MI355X, 32 x 100 k points, sweeps 0 / 1 / 2; docs/NOTEBOOK_r6.md section 1): 2 699 / 2 108 / 1 610 vector instructions per wave at
// 44-49 active lanes against k_sweep_fused's 3 072 / 2 614 / 2 093 at 31-35 -- and 453 / 287 / 237 us against 395 / 288 / 222: a quarter
// fewer instructions buy nothing, because a workgroup's life is its chain of dependent memory round trips (loads, greedy descent, start
// cells, a node and a leaf round per tree level, the final gather: ~16, each behind a barrier), as long as a lane's walk was.  NOT the
// default (LH_SWEEP_COOP=1 selects it); it passes the same exact-neighbour and whole-alignment tests (tests/test_gpu_coop.py).
#ifndef LH_COOP_NCAP
#define LH_COOP_NCAP 640
#endif
#ifndef LH_COOP_LCAP
#define LH_COOP_LCAP 768
#endif
#ifndef LH_COOP_WAVES
#define LH_COOP_WAVES 5
#endif
constexpr int COOP_NCAP = LH_COOP_NCAP, COOP_LCAP = LH_COOP_LCAP;
// LDS atomics on address-space-3 pointers (ds_add_rtn_u32 / ds_min_u32 / ds_min_rtn_u64; through a generic pointer they would be flat atomics)
typedef __attribute__((address_space(3))) uint32_t* LdsU32;
typedef __attribute__((address_space(3))) unsigned long long* LdsU64;
__device__ __forceinline__ uint32_t lds_add(LdsU32 p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_min(LdsU32 p, uint32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned long long lds_min64(LdsU64 p, unsigned long long v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
constexpr int COOP_GREEDY_BIT = 1 << 30;        // SweepJob::pad: the sweep starts with the greedy descent (its candidate is a loose bound)
constexpr int COOP_MAX_POINTS = 1 << 24;        // a leaf item keeps 24 bits of sorted position beside the 8-bit query slot
constexpr size_t COOP_OFF_KEY = 0, COOP_OFF_GQ = 2048, COOP_OFF_Q = 6144, COOP_OFF_NQ = 10240, COOP_OFF_LQ = COOP_OFF_NQ + 2 * 8 * (size_t)COOP_NCAP,
                 COOP_OFF_CTR = COOP_OFF_LQ + 2 * 4 * (size_t)COOP_LCAP, COOP_LDS_BYTES = COOP_OFF_CTR + 32;
static_assert(COOP_LDS_BYTES >= (size_t)LDS_STACK * 256 * 8, "the fallback's traversal stacks alias the queues");
static_assert(COOP_LDS_BYTES >= 4 * 32 * GRAM_RS * 8, "the Gram staging rows alias the queues");
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LH_COOP_WAVES))) k_sweep_coop(const PairDesc* __restrict__ descs, SweepArgs a, double* __restrict__ partials,
                                                                                       int partials_stride, const OuterState* __restrict__ states) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_stack[];
  typedef __attribute__((address_space(3))) unsigned char* LdsBytes;
  LdsBytes const lds = (LdsBytes)lds_stack;
  auto* const skey = (__attribute__((address_space(3))) unsigned long long*)(lds + COOP_OFF_KEY);   // per query: d2 bits << 32 | index
  typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));   // (HIP's vector classes have no address-space-qualified assignment)
  typedef float F32x4 __attribute__((ext_vector_type(4)));
  auto* const sgq = (__attribute__((address_space(3))) U32x4*)(lds + COOP_OFF_GQ);                  // per query: GridQuery
  auto* const sq = (__attribute__((address_space(3))) F32x4*)(lds + COOP_OFF_Q);                    // per query: x y z, bits of the certificate bound
  auto lb_of = [&](int q) { return (LdsU32)(lds + COOP_OFF_Q + 16 * (size_t)q + 12); };
  auto* const nq = (__attribute__((address_space(3))) unsigned long long*)(lds + COOP_OFF_NQ);      // node items [2][NCAP]: (key & ~255 | query) << 32 | node
  auto* const lq = (__attribute__((address_space(3))) uint32_t*)(lds + COOP_OFF_LQ);                // leaf items [2][LCAP]: query << 24 | sorted position
  auto* const ctr = (__attribute__((address_space(3))) uint32_t*)(lds + COOP_OFF_CTR);              // [0..1] node counts, [2..3] leaf counts, [4] overflow
  int jb, blk;
  if (!xcd_job_map(a.njobs, a.bpj, jb, blk)) return;
  const SweepJob& job = a.job[jb];
  const PairDesc d = descs[job.slot];
  if (blk * 256 >= d.n) return;  // (uniform)
  float T[12];
  if (!job_transform(job, states, T)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = blk * 256 + tid;
  const int cold = job.pad & 0xffff;
  const bool greedy = (job.pad & COOP_GREEDY_BIT) != 0;
  if (tid < 8) ctr[tid] = 0u;
  TreeView tv{d.tgt_sorted, d.tgt_nodes, d.tgt_hdr, d.m};
  TreeHeader h;
  h.root = gld(&tv.hdr->root);
  h.org[0] = gld(&tv.hdr->org[0]); h.org[1] = gld(&tv.hdr->org[1]); h.org[2] = gld(&tv.hdr->org[2]);
  h.inv = gld(&tv.hdr->inv); h.scl2 = gld(&tv.hdr->scl2);
  const bool grid_on = gld(&tv.hdr->grid_on) != 0;
  const float scl2 = h.scl2;
  const float INF = INFINITY;
  // ---- phase 1a: sweep_point's first half -- the candidate and the certificate test
  bool need_search = false;
  int w = -1;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  Nn1CertCollector col{INF, 0x7fffffff, INF};
  if (i < d.n) {
    const float4 p = gld(d.src + i);
    w = gld(d.prev_nn + i);
    int w_alt = -1;
    float4 t_alt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cold > 1 && w >= 0) {
      const int e = i % cold;
      if (e > 0) {
        w_alt = min(d.m - 1, w + e);
        t_alt = gld(d.tgt_xyz + w_alt);
      }
    }
    float4 cq = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 t = cq;
    if (!cold) {
      cq = gld(d.cert + i);
      t = rec_pos(d, i);
    } else if (w >= 0)
      t = gld(d.tgt_xyz + w);
    xform_pt(T, p.x, p.y, p.z, qx, qy, qz);  // gicp.hpp:469
    need_search = true;
    if (w_alt >= 0 && d2f(qx, qy, qz, t_alt.x, t_alt.y, t_alt.z) < d2f(qx, qy, qz, t.x, t.y, t.z)) {
      w = w_alt;
      t = t_alt;
    }
    if (w >= 0) {
      col.bd = d2f(qx, qy, qz, t.x, t.y, t.z);
      col.bi = w;
      const float e = sqrt_bound(d2f(qx, qy, qz, cq.x, cq.y, cq.z));
      const float dw = sqrt_bound(col.bd), lo = sqrt_bound(cq.w);
      if (!cold && dw * (1.0f + a.cert_rel) + e * (1.0f + a.cert_rel) + 1e-12f < lo * (1.0f - a.cert_rel)) need_search = false;
    }
  }
  __syncthreads();   // the counters are zero
  // ---- phase 1b: greedy descent (loose candidates only), start items
  if (need_search) {
    const GridQuery gq = grid_query(h, qx, qy, qz);
    if (greedy) {
      int32_t r = h.root;
      if (grid_on) {   // from the query's own level-5 cell if that holds anything (tree_descend<.., true>)
        const float ksc = gld(&tv.hdr->key_sc);
        const int cx = (int)fminf(fmaxf((qx - h.org[0]) * ksc, 0.0f), 1023.0f) >> 5, cy = (int)fminf(fmaxf((qy - h.org[1]) * ksc, 0.0f), 1023.0f) >> 5,
                  cz = (int)fminf(fmaxf((qz - h.org[2]) * ksc, 0.0f), 1023.0f) >> 5;
        const int32_t g = gld(tv.grid() + grid_index(5, cx, cy, cz));
        if (g != GRID_EMPTY) r = g;
      }
      while (r >= 0) {
        const NodeX& nd = tv.nodes[r];
        const uint4 ba = gload16<uint4>(nd.lo_xy);
        const uint4 bb = gload16<uint4>(nd.hi_xy);
        const uint4 bc = gload16<uint4>(nd.z_lohi);
        const int4 ch = gload16<int4>(nd.child);
        float d0 = boxd2_q(gq, ba.x, bb.x, bc.x, scl2), d1 = boxd2_q(gq, ba.y, bb.y, bc.y, scl2);
        float d2 = boxd2_q(gq, ba.z, bb.z, bc.z, scl2), d3 = boxd2_q(gq, ba.w, bb.w, bc.w, scl2);
        d1 = ch.y == NO_CHILD ? INF : d1;
        d2 = ch.z == NO_CHILD ? INF : d2;
        d3 = ch.w == NO_CHILD ? INF : d3;
        float dm = d0; int32_t rm = ch.x;
        if (d1 < dm) { dm = d1; rm = ch.y; }
        if (d2 < dm) { dm = d2; rm = ch.z; }
        if (d3 < dm) { dm = d3; rm = ch.w; }
        r = rm;
      }
      scan_leaf(tv, r, qx, qy, qz, col);
    }
    auto push = [&](uint32_t key, int32_t ref) {   // a start item (one LDS atomic per item: at most eight per query, once)
      if (ref >= 0) {
        const uint32_t pos = lds_add(&ctr[0], 1u);
        if (pos < (uint32_t)COOP_NCAP) nq[pos] = ((unsigned long long)((key & ~255u) | (uint32_t)tid) << 32) | (uint32_t)ref;
        else ctr[4] = 1u;
      } else {
        const uint32_t pos = lds_add(&ctr[2], 1u);
        if (pos < (uint32_t)COOP_LCAP) lq[pos] = ((uint32_t)tid << 24) | ((uint32_t)~ref >> 4);
        else ctr[4] = 1u;
      }
    };
    int32_t ref = h.root;
    if (col.bd < INF && grid_on) {
      const int32_t g = grid_start(h.org, gld(&tv.hdr->key_sc), gld(&tv.hdr->key_inv), tv.grid(), qx, qy, qz, col, push);
      if (g != GRID_USE_ROOT) ref = g;
    }
    if (ref != GRID_EMPTY) push(0u, ref);
    skey[tid] = ((unsigned long long)f2u(col.bd) << 32) | (uint32_t)col.bi;
    sgq[tid] = U32x4{gq.up_xy, gq.dn_xy, gq.z, f2u(gq.e2)};
    sq[tid] = F32x4{qx, qy, qz, col.lb};
  }
  __syncthreads();   // the start items, the queries' slots
  // ---- phase 2: rounds
  bool fallback = d.m + LEAF_CAP > COOP_MAX_POINTS;   // (a leaf item could not hold the sorted position: per-lane walks)
  int par = 0;
  for (;;) {
    const uint32_t nN = ctr[par], nL = ctr[2 + par];
    if (fallback || ctr[4] != 0u) { fallback = true; break; }
    if (nN == 0u && nL == 0u) break;
    auto* const lqc = lq + par * COOP_LCAP;
    for (uint32_t c0 = (uint32_t)wave * 64u; c0 < nL; c0 += 256u) {   // leaf round
      const uint32_t k = c0 + (uint32_t)lane;
      if (k < nL) {
        const uint32_t it = lqc[k];
        const int qid = (int)(it >> 24);
        const F32x4 qq = sq[qid];
        const unsigned long long snap = skey[qid];
        Nn1CertCollector c{u2f((uint32_t)(snap >> 32)), (int)(uint32_t)snap, INF};
        const float4* p = tv.pts + (it & 0xffffffu);
        float4 v[LEAF_CAP];
#pragma unroll
        for (int e = 0; e < LEAF_CAP; e++) v[e] = gload16<float4>(p + e);
#pragma unroll
        for (int e = 0; e < LEAF_CAP; e++) c.offer(d2f(qq.x, qq.y, qq.z, v[e].x, v[e].y, v[e].z), (int)f2u(v[e].w));
        const unsigned long long mine = ((unsigned long long)f2u(c.bd) << 32) | (uint32_t)c.bi;
        float contrib = c.lb;
        if (mine < snap) {
          const unsigned long long old = lds_min64(&skey[qid], mine);
          if (old != mine) contrib = fminf(contrib, u2f((uint32_t)((old > mine ? old : mine) >> 32)));   // the loser of the two is a runner-up
        }
        if (contrib < INF) lds_min(lb_of(qid), f2u(contrib));
      }
    }
    __syncthreads();   // the bounds of this round's node tests are final; everybody has read the counts
    if (tid == 0) { ctr[par] = 0u; ctr[2 + par] = 0u; }   // (pushed to again only after the next barrier but one)
    auto* const nqc = nq + par * COOP_NCAP;
    auto* const nqn = nq + (par ^ 1) * COOP_NCAP;
    auto* const lqn = lq + (par ^ 1) * COOP_LCAP;
    for (uint32_t c0 = (uint32_t)wave * 64u; c0 < nN; c0 += 256u) {   // node round
      const uint32_t k = c0 + (uint32_t)lane;
      bool v0 = false, v1 = false, v2 = false, v3 = false;
      int4 ch = make_int4(NO_CHILD, NO_CHILD, NO_CHILD, NO_CHILD);
      float d0 = INF, d1 = INF, d2 = INF, d3 = INF;
      int qid = 0;
      if (k < nN) {
        const unsigned long long it = nqc[k];
        const uint32_t hi = (uint32_t)(it >> 32);
        qid = (int)(hi & 255u);
        const float key = u2f(hi & ~255u);
        const float bd = u2f((uint32_t)(skey[qid] >> 32));
        if (key <= bd) {
          const U32x4 g4 = sgq[qid];
          const GridQuery gq{g4.x, g4.y, g4.z, u2f(g4.w)};
          const NodeX& nd = tv.nodes[(int32_t)(uint32_t)it];
          const uint4 ba = gload16<uint4>(nd.lo_xy);
          const uint4 bb = gload16<uint4>(nd.hi_xy);
          const uint4 bc = gload16<uint4>(nd.z_lohi);
          ch = gload16<int4>(nd.child);
          d0 = boxd2_q(gq, ba.x, bb.x, bc.x, scl2);
          d1 = boxd2_q(gq, ba.y, bb.y, bc.y, scl2);
          d2 = boxd2_q(gq, ba.z, bb.z, bc.z, scl2);
          d3 = boxd2_q(gq, ba.w, bb.w, bc.w, scl2);
          d1 = ch.y == NO_CHILD ? INF : d1;
          d2 = ch.z == NO_CHILD ? INF : d2;
          d3 = ch.w == NO_CHILD ? INF : d3;
          v0 = d0 <= bd && d0 < INF; v1 = d1 <= bd && d1 < INF; v2 = d2 <= bd && d2 < INF; v3 = d3 <= bd && d3 < INF;
          const float sk = fminf(fminf(v0 ? INF : d0, v1 ? INF : d1), fminf(v2 ? INF : d2, v3 ? INF : d3));   // the dropped children (Collector::skip)
          if (sk < INF) lds_min(lb_of(qid), f2u(sk));
        } else
          lds_min(lb_of(qid), f2u(key));   // dropped at the pop
      }
      // survivors -> the next round's queues: one atomic per wave and kind, positions by ballot
      const bool n0 = v0 && ch.x >= 0, n1 = v1 && ch.y >= 0, n2 = v2 && ch.z >= 0, n3 = v3 && ch.w >= 0;
      const bool l0 = v0 && ch.x < 0, l1 = v1 && ch.y < 0, l2 = v2 && ch.z < 0, l3 = v3 && ch.w < 0;
      const unsigned long long mn0 = __ballot(n0), mn1 = __ballot(n1), mn2 = __ballot(n2), mn3 = __ballot(n3);
      const unsigned long long ml0 = __ballot(l0), ml1 = __ballot(l1), ml2 = __ballot(l2), ml3 = __ballot(l3);
      const uint32_t tn = (uint32_t)(__popcll(mn0) + __popcll(mn1) + __popcll(mn2) + __popcll(mn3));
      const uint32_t tl = (uint32_t)(__popcll(ml0) + __popcll(ml1) + __popcll(ml2) + __popcll(ml3));
      uint32_t bn = 0u, bl = 0u;
      if (lane == 0) {
        if (tn) bn = lds_add(&ctr[par ^ 1], tn);
        if (tl) bl = lds_add(&ctr[2 + (par ^ 1)], tl);
      }
      bn = (uint32_t)__builtin_amdgcn_readfirstlane((int)bn);
      bl = (uint32_t)__builtin_amdgcn_readfirstlane((int)bl);
      if (bn + tn > (uint32_t)COOP_NCAP || bl + tl > (uint32_t)COOP_LCAP) {
        if (lane == 0) ctr[4] = 1u;
      } else {
        auto below = [&](unsigned long long m) { return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
        auto nitem = [&](float dk, int32_t r) { return ((unsigned long long)((f2u(dk) & ~255u) | (uint32_t)qid) << 32) | (uint32_t)r; };
        auto litem = [&](int32_t r) { return ((uint32_t)qid << 24) | ((uint32_t)~r >> 4); };
        if (n0) nqn[bn + below(mn0)] = nitem(d0, ch.x);
        bn += (uint32_t)__popcll(mn0);
        if (n1) nqn[bn + below(mn1)] = nitem(d1, ch.y);
        bn += (uint32_t)__popcll(mn1);
        if (n2) nqn[bn + below(mn2)] = nitem(d2, ch.z);
        bn += (uint32_t)__popcll(mn2);
        if (n3) nqn[bn + below(mn3)] = nitem(d3, ch.w);
        if (l0) lqn[bl + below(ml0)] = litem(ch.x);
        bl += (uint32_t)__popcll(ml0);
        if (l1) lqn[bl + below(ml1)] = litem(ch.y);
        bl += (uint32_t)__popcll(ml1);
        if (l2) lqn[bl + below(ml2)] = litem(ch.z);
        bl += (uint32_t)__popcll(ml2);
        if (l3) lqn[bl + below(ml3)] = litem(ch.w);
      }
    }
    __syncthreads();   // the next round's items
    par ^= 1;
  }
  // ---- sweep_point's second half: the neighbour, its certificate, the Mahalanobis matrix
  SweepPoint sp;
  sp.matched = false;
  sp.searched = false;
  sp.nonn = false;
  sp.p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fallback) {
    __syncthreads();   // (everybody has left the loop: the traversal stacks alias the queues)
    if (i < d.n) sweep_point<true, 256>(d, T, i, lds_stack + tid, sp, a.cert_rel, cold);
  } else if (i < d.n) {
    if (need_search) {
      const unsigned long long kk = skey[tid];
      const F32x4 qq = sq[tid];
      col.bd = u2f((uint32_t)(kk >> 32));
      col.bi = (int)(uint32_t)kk;
      col.lb = qq.w;
      gst(d.cert + i, make_float4(qx, qy, qz, nn_index(col.bi, col.bd) >= 0 ? col.lb : -1.0f));
      if (d.stats) atomicAdd(&d.stats[0], 1ull);
    }
    if (d.stats) atomicAdd(&d.stats[1], 1ull);
    const int j = nn_index(col.bi, col.bd);
    sp.searched = need_search;
    sp.nonn = j < 0;
    sp.j = j;
    sp.p = gld(d.src + i);
    if (j != w || cold) gst(d.prev_nn + i, j);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f), tn = t, nn = t;
    if (j >= 0) {
      nn = gld(d.src_nrm + i);
      if (need_search) {
        t = gld(d.tgt_xyz + j);
        tn = gld(d.tgt_nrm + j);
        if ((j != w || cold) && d.rec) rec_put(d, i, t, tn);   // the record follows prev_nn
      } else {
        t = rec_pos(d, i);
        tn = rec_nrm(d, i);
      }
    }
    sp.matched = j >= 0 && (double)col.bd < d.corr_dist2;  // gicp.hpp:483
    if (sp.matched) {   // (sweep_point's kRank1 branch, expression for expression: every sweep kernel must produce the same bits for a point)
      double R[9];
      if (d.guess_identity) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int cc = 0; cc < 3; cc++) R[r * 3 + cc] = (double)T[r * 4 + cc];
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int cc = 0; cc < 3; cc++)
            R[r * 3 + cc] = (((double)T[r * 4 + 0] * d.guess3[0 * 3 + cc] + (double)T[r * 4 + 1] * d.guess3[1 * 3 + cc]) +
                             (double)T[r * 4 + 2] * d.guess3[2 * 3 + cc]) + (double)T[r * 4 + 3] * 0.0;
      }
      double G[6], M6[6];
      {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int cc = r; cc < 3; cc++) {
            const double rr = __builtin_fma(R[r * 3 + 2], R[cc * 3 + 2], __builtin_fma(R[r * 3 + 1], R[cc * 3 + 1], R[r * 3 + 0] * R[cc * 3 + 0]));
            G[q++] = r == cc ? rr + 1.0 : rr;
          }
      }
      maha_rank1(R, 3, G, 1.0 - d.gicp_eps, nn, tn, M6);
      sp.M[0] = M6[0]; sp.M[1] = M6[1]; sp.M[2] = M6[2];
      sp.M[3] = M6[1]; sp.M[4] = M6[3]; sp.M[5] = M6[4];
      sp.M[6] = M6[2]; sp.M[7] = M6[4]; sp.M[8] = M6[5];
      sp.tgt = t;
    }
  }
  // ---- the 74 moments: k_sweep_fused's tail with four waves per workgroup (one row per 256 points, like k_late's)
  double M6[6] = {0, 0, 0, 0, 0, 0}, Ma[3] = {0, 0, 0}, pt[4] = {0, 0, 0, 0}, pp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, aMa = 0.0;
  if (sp.matched) moments_of_point(T, sp, M6, Ma, aMa, pt, pp);
  const double live = sp.matched ? 1.0 : (sp.nonn ? NO_NN_MARK : 0.0);
  const int walks = __popcll(__ballot(sp.searched));
  double* out = partials + (size_t)job.slot * partials_stride + (size_t)blk * MOM_ROW;
  {
    const double av[11] = {M6[0], M6[1], M6[2], M6[3], M6[4], M6[5], Ma[0], Ma[1], Ma[2], aMa, live};
    GramAcc acc;
    gram_zero(acc);
    __syncthreads();  // every lane is done with the queues / its traversal stack
    double* wl = reinterpret_cast<double*>(lds_stack) + wave * (32 * GRAM_RS);
    gram_accumulate(wl, av, sp.p, sp.matched, acc, sp.nonn);
    gram_store(acc, wl, (double)walks);
    wg_row_sum(reinterpret_cast<double*>(lds_stack), 32 * GRAM_RS, out);
  }
  if (a.span > 0) {   // the walk rows of this job stay zero (k_moments_final adds the same rows whichever way the sweep was launched)
    const int bps = a.span >> 8;
    if (blk % bps == 0 && tid < MOM_ROW) partials[(size_t)job.slot * partials_stride + ((size_t)((d.n + 255) / 256) + blk / bps) * MOM_ROW + tid] = 0.0;
  }
}

// ===== the same sweep in two launches (cost_mode 1, covariances from normals, guess = I) =======================================
// What bounds k_sweep_fused once most certificates hold (SQ counters, profiles/): a wave that has nothing to search still carries
// the traversal's costs -- 24 KB of LDS stack per workgroup (6 waves per SIMD), a workgroup barrier in front of the reduction (its
// staging area aliases the stacks) -- and in the iterations where a few per cent of the queries still walk, nearly every wave
// executes a whole tree descent for its one or two walkers (measured 3-13 of 64 lanes active).  So the work is split by WHAT A
// POINT NEEDS:
//   k_late  every source point, one round of loads (point, normal, certificate, neighbour record), certificate test; the points
//           whose certificate holds are finished here -- Mahalanobis matrix, 74 moments, Gram reduction per wave with a
//           wave-PRIVATE staging area: no traversal stack, no workgroup barrier, 7 waves per SIMD.  The others only leave one
//           bit in the wave's 64-bit walker mask (written, not appended: deterministic).
//   k_walk  one WAVE per `span` consecutive source points (four independent waves per workgroup): looks at the span's masks
//           first and ends if nobody walks (most spans of most iterations); otherwise queues the walkers and runs their exact
//           searches with PERSISTENT LANES -- a lane whose search has ended takes the next walker from the queue (once `refill`
//           lanes are idle), so the wave's lanes stay busy whatever the fraction of walkers and however unequal the walks;
//           then refreshes neighbour, certificate and record and reduces the walkers' moments into one more row per span.
// k_moments_final adds both row sets in fixed order, so results stay bitwise reproducible and independent of batching.
// Measured (32 x 100 k points, MI355X): iterations with 25 / 7 / 2 % walkers 300 / 203 / 145 us against 310 / 255 / 170 fused; in
// the iterations where (nearly) everything walks the fused kernel is faster (its walks overlap the other waves' streaming), so
// the first LH_SPLIT_FROM (3) sweeps of a pair stay fused.  Certificate-only iterations: 62 + 24 + 10 us (k_late, k_walk, final
// sum) against 80 + 10 in isolation -- k_walk's 24 us are the latency of ONE search plus its moments (two dozen dependent memory
// round trips of a lone wave: a pair of 100 k points nearly always has one or two points whose two nearest neighbours tie to
// within the certificate's safety margin; a k_walk whose waves all end after the mask test takes 2 us); with several scheduler
// groups in flight that latency overlaps other streams' kernels and the step gets 3-6 % faster, one pair alone pays it
// (1.2 -> 1.5 ms for 20 iterations).  Tried and dropped: the walks inside k_late by the last wave of a span to finish (a device-wide counter
// needs agent-scope release/acquire = L2 write-back: 1.4 ms per launch; a workgroup-wide one keeps the workgroup's LDS and
// registers while one wave walks: walking iterations 40 % slower).
// Mahalanobis matrix (maha_rank1) and moment operands of one matched pair of points; guess = I
__device__ __forceinline__ void point_terms(const PairDesc& d, const PoseD& P, const float4& p, const float4& nn, const float4& t, const float4& tn, double (&av)[11]) {
  double M6[6], Ma[3], aMa;
  maha_rank1(P.T, 4, P.G, 1.0 - d.gicp_eps, nn, tn, M6);
  const double pt[3] = {(double)p.x, (double)p.y, (double)p.z};
  resid_terms(P.T, M6, pt, t, Ma, aMa);
#pragma unroll
  for (int k = 0; k < 6; k++) av[k] = M6[k];
  av[6] = Ma[0]; av[7] = Ma[1]; av[8] = Ma[2];
  av[9] = aMa;
  av[10] = 1.0;
}
// the pose of a pair in the device-driven loop: k_solve left it in the pair's state (scalar loads, like job_transform's)
__device__ __forceinline__ void pose_load(const SweepJob& job, const OuterState* __restrict__ states, PoseD& P) {
  typedef const __attribute__((address_space(4))) OuterState* ConstState;
  ConstState st = (ConstState)(uintptr_t)(states + job.slot);
#pragma unroll
  for (int k = 0; k < 12; k++) P.T[k] = st->poseT[k];
#pragma unroll
  for (int k = 0; k < 6; k++) P.G[k] = st->poseG[k];
  // (pinned here: left alone the compiler sinks the loads into the matched-points branch, where every wave then waits for them)
#pragma unroll
  for (int k = 0; k < 12; k++) asm volatile("" : "+s"(P.T[k]));
#pragma unroll
  for (int k = 0; k < 6; k++) asm volatile("" : "+s"(P.G[k]));
}

// kDev: launched by the device-driven loop (states != nullptr) -- the pose comes from the pair's state instead of being derived per wave
template <bool kDev>
#ifndef LH_LATE_WAVES
#define LH_LATE_WAVES 7, 8
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LH_LATE_WAVES))) k_late(const PairDesc* __restrict__ descs, SweepArgs a, double* __restrict__ partials,
                                                                                   int partials_stride, const OuterState* __restrict__ states,
                                                                                   unsigned long long* __restrict__ wmask, int mask_stride) {
  extern __shared__ __attribute__((aligned(16))) double lds_gram[];  // [4 waves][32 rows][GRAM_RS]
  int jb, blk;
  if (!xcd_job_map(a.njobs, a.bpj, jb, blk)) return;
  const SweepJob& job = a.job[jb];
  const PairDesc d = descs[job.slot];
  if (blk * 256 >= d.n) return;
  float T[12];
  if (!job_transform(job, states, T)) return;
  PoseD P;
  if constexpr (kDev) pose_load(job, states, P);   // (scalar loads, issued beside the transform's)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = blk * 256 + tid;
  bool walker = false, matched = false;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f), nn = p, t = p, tn = p;
  if (i < d.n) {
    // ONE round of loads: the point, its normal, the certificate of the last search and the neighbour's record -- 72 bytes per point
    // (the neighbour's index is not needed: a point without a neighbour carries a negative bound in its certificate)
    p = gld(d.src + i);
    const float4 cq = gld(d.cert + i);
    nn = gld(d.src_nrm + i);
    t = rec_pos(d, i);
    tn = rec_nrm(d, i);
    float qx, qy, qz;
    xform_pt(T, p.x, p.y, p.z, qx, qy, qz);  // gicp.hpp:469
    // the certificate test of sweep_point: the neighbour of the last search is provably still the nearest
    const float bd = d2f(qx, qy, qz, t.x, t.y, t.z);
    const float e = sqrt_bound(d2f(qx, qy, qz, cq.x, cq.y, cq.z));
    const float dw = sqrt_bound(bd), lo = sqrt_bound(cq.w);
    // (one test, no branch: a short-circuit on the sign made the compiler fetch the certificate in two dependent rounds)
    const bool ok = (int)(cq.w >= 0.0f) & (int)(dw * (1.0f + a.cert_rel) + e * (1.0f + a.cert_rel) + 1e-12f < lo * (1.0f - a.cert_rel));
    walker = !ok;
    matched = ok && (double)bd < d.corr_dist2;  // gicp.hpp:483
  }
  const unsigned long long mask = __ballot(walker);
  if (lane == 0) {
    wmask[(size_t)job.slot * mask_stride + blk * 4 + wave] = mask;
  }
  {  // the span's walk row starts as zeros (k_walk overwrites it if the span has walkers)
    const int bps = a.span >> 8;
    if (blk % bps == 0 && tid < MOM_ROW)
      partials[(size_t)job.slot * partials_stride + ((size_t)((d.n + 255) / 256) + blk / bps) * MOM_ROW + tid] = 0.0;
  }
  double* out = partials + (size_t)job.slot * partials_stride + (size_t)blk * MOM_ROW;   // one row per workgroup
  double* myrow = lds_gram + wave * (32 * GRAM_RS);
  if (__ballot(matched) == 0ull) {  // nothing to add (every point of this wave walks: the first sweeps): a zero row, no math, no MFMA
    myrow[lane] = 0.0;
    if (lane < MOM_ROW - 64) myrow[64 + lane] = (lane == MOM_NSUM - 64) ? (double)__popcll(mask) : 0.0;
  } else {
    double av[11];
    if (matched) {   // gicp.hpp:488-498
      if constexpr (!kDev) pose_from_T(T, P);
      point_terms(d, P, p, nn, t, tn, av);
    } else {
#pragma unroll
      for (int k = 0; k < 11; k++) av[k] = 0.0;
    }
    GramAcc acc;
    gram_zero(acc);
    gram_accumulate(myrow, av, p, matched, acc);
    gram_store(acc, myrow, (double)__popcll(mask));
  }
  wg_row_sum(lds_gram, 32 * GRAM_RS, out);
}

constexpr int WALK_STACK = 8;        // traversal-stack entries a lane keeps in LDS (warm walks rarely go deeper; the rest spills to private memory)
constexpr int WALK_SPAN_MAX = 512;   // source points per k_walk wave, at most
// Four INDEPENDENT waves per workgroup, each with its own span (no barrier anywhere): one-wave workgroups cost 25 us per launch in
// workgroup dispatch alone (6 000 of them, each with its LDS allocation), whether or not anybody walks.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) k_walk(const PairDesc* __restrict__ descs, SweepArgs a, double* __restrict__ partials, int partials_stride,
                                             const OuterState* __restrict__ states, const unsigned long long* __restrict__ wmask, int mask_stride) {
  __shared__ __attribute__((aligned(16))) uint64_t lds_stack_all[4][WALK_STACK * 64];  // per wave [entry][lane]; after the walks: the Gram staging rows (32 x 15 doubles)
  __shared__ uint16_t queue_all[4][WALK_SPAN_MAX];
  static_assert(WALK_STACK * 64 >= 32 * GRAM_RS, "the staging rows alias the stack");
  int jb, blk4;
  if (!xcd_job_map(a.njobs, a.bpj, jb, blk4)) return;
  const SweepJob& job = a.job[jb];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int span = a.span;
  const int blk = blk4 * 4 + wv;   // this wave's span
  const int wps = span >> 6;       // mask words per span (<= 8)
  // The masks FIRST, straight from the launch arguments (slot, stride): most spans of most iterations have no walker, and their
  // waves must not pay the chain descriptor -> state -> transform before they find out (24 us per launch when they did).
  unsigned long long m = 0ull;
  if (lane < wps && blk * wps + lane < mask_stride) m = wmask[(size_t)job.slot * mask_stride + blk * wps + lane];
  if (__ballot(m != 0ull) == 0ull) return;  // nobody in this span needs a search: k_late left a zero row
  uint64_t* const lds_stack = lds_stack_all[wv];
  uint16_t* const queue = queue_all[wv];
  const PairDesc d = descs[job.slot];
  if (blk * span >= d.n) return;
  float T[12];
  if (!job_transform(job, states, T)) return;
  const int n_words = ((d.n + 255) / 256) * 4;
  if (blk * wps + lane >= n_words) m = 0ull;   // (words past this pair's last workgroup may be a former, larger pair's)
  const int cnt = __popcll(m);
  int pre = cnt;   // inclusive prefix over the first 8 lanes
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    int v = __shfl_up(pre, off);
    if (lane >= off) pre += v;
  }
  const int total = __builtin_amdgcn_readlane(pre, 7);   // (lanes >= wps hold empty masks)
  pre -= cnt;
  if (total == 0) return;
  double* out = partials + (size_t)job.slot * partials_stride + ((size_t)(n_words >> 2) + blk) * MOM_ROW;   // behind the pair's workgroup rows
  for (int k = 0; k < wps; k++) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m, k), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m >> 32), k);
    const unsigned long long mk = ((unsigned long long)hi << 32) | lo;
    const int pk = __builtin_amdgcn_readlane(pre, k);
    if ((mk >> lane) & 1ull) queue[pk + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)(k * 64 + lane);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const int i0 = blk * span;
  {
    // ---- the walks: tree_search's loop (lh_device.hpp) as a state machine, so that a lane can start its next query while
    // the others are in the middle of theirs.  Same visiting rule, same collector => same neighbours and certificates.
    TreeView tv{d.tgt_sorted, d.tgt_nodes, d.tgt_hdr, d.m};
    TreeHeader h;
    h.root = gld(&tv.hdr->root);
    h.org[0] = gld(&tv.hdr->org[0]); h.org[1] = gld(&tv.hdr->org[1]); h.org[2] = gld(&tv.hdr->org[2]);
    h.inv = gld(&tv.hdr->inv); h.scl2 = gld(&tv.hdr->scl2);
    const int32_t root = h.root;
    const bool grid_on = gld(&tv.hdr->grid_on) != 0;
    const float key_sc = gld(&tv.hdr->key_sc), key_inv = gld(&tv.hdr->key_inv);
    const int32_t DONE = NO_CHILD;
    WalkStack<WALK_STACK> stk(lds_stack + lane, 64);
    Nn1CertCollector col{INFINITY, 0x7fffffff, INFINITY};
    GridQuery gq{0u, 0u, 0u, 0.f};
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int qi = -1;
    int32_t ref = DONE;
    int head = 0;
    const int refill = a.refill;
    for (;;) {
      const bool idle = ref == DONE;
      const unsigned long long im = __ballot(idle);
      const int nidle = __popcll(im);
      if (nidle == 64 || (head < total && nidle >= refill)) {
        if (idle) {
          if (qi >= 0) {  // the search that has just ended: its neighbour and certificate
            const int i = i0 + qi;
            const int j = nn_index(col.bi, col.bd);
            gst(d.cert + i, make_float4(qx, qy, qz, j >= 0 ? col.lb : -1.0f));
            gst(d.prev_nn + i, j);
            qi = -1;
          }
          const int slot = head + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
          if (slot < total) {
            qi = queue[slot];
            const int i = i0 + qi;
            const float4 p = gld(d.src + i);
            const int w = gld(d.prev_nn + i);
            xform_pt(T, p.x, p.y, p.z, qx, qy, qz);  // gicp.hpp:469
            col = Nn1CertCollector{INFINITY, 0x7fffffff, INFINITY};
            if (w >= 0) {  // warm start: the previous neighbour is a valid candidate => tight initial bound, still exact
              const float4 t0 = rec_pos(d, i);
              col.bd = d2f(qx, qy, qz, t0.x, t0.y, t0.z);
              col.bi = w;
            }
            gq = grid_query(h, qx, qy, qz);
            stk.sp = 0;
            ref = root;
            if (grid_on && col.bd < INFINITY) {   // start at the query's own cell of the start grid (grid_start, lh_device.hpp), like tree_search<.., true>
              const int32_t g = grid_start(h.org, key_sc, key_inv, tv.grid(), qx, qy, qz, col, [&](uint32_t key, int32_t r) { stk.push(key, r); });
              if (g != GRID_USE_ROOT) ref = (g == GRID_EMPTY) ? stk.pop(col) : g;
            }
          }
        }
        head = min(total, head + nidle);
        if (__ballot(ref != DONE) == 0ull) break;
      }
      while (ref >= 0 && ref != DONE) ref = node_visit(tv.nodes[ref], gq, h.scl2, col, stk);
      if (ref < 0) {
        scan_leaf(tv, ref, qx, qy, qz, col);
        ref = stk.pop(col);
      }
    }
  }
  // ---- the walkers' moments: their new neighbours are in prev_nn (written by lanes of this wave: workgroup-scope visibility)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  GramAcc acc;
  gram_zero(acc);
  for (int base = 0; base < total; base += 64) {
    const int r = base + lane;
    bool matched = false, nonn = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), nn = p, t = p, tn = p;
    if (r < total) {
      const int i = i0 + queue[r];
      p = gld(d.src + i);
      nn = gld(d.src_nrm + i);
      const int j = gld(d.prev_nn + i);
      nonn = j < 0;   // the search found nothing (a non-finite query): gicp.hpp:471-478
      if (j >= 0) {
        t = gld(d.tgt_xyz + j);
        tn = gld(d.tgt_nrm + j);
        rec_put(d, i, t, tn);   // the record follows prev_nn
        float qx, qy, qz;
        xform_pt(T, p.x, p.y, p.z, qx, qy, qz);
        matched = (double)d2f(qx, qy, qz, t.x, t.y, t.z) < d.corr_dist2;  // gicp.hpp:483 (the same float the search ended with)
      }
    }
    // the transform's doubles must be re-made from the scalar floats in every round: hoisted out of the loop they would cost 40
    // vector registers (and the kernel three of its seven waves per SIMD)
    float Tl[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      Tl[k] = T[k];
      asm volatile("" : "+s"(Tl[k]));
    }
    double av[11];
    if (matched) {
      PoseD P;
      pose_from_T(Tl, P);
      point_terms(d, P, p, nn, t, tn, av);
    } else {
#pragma unroll
      for (int k = 0; k < 11; k++) av[k] = 0.0;
      if (nonn) av[10] = NO_NN_MARK;
    }
    gram_accumulate(reinterpret_cast<double*>(lds_stack), av, p, matched, acc, nonn);
  }
  gram_store(acc, out, 0.0);   // (the late rows carry the walker counts)
}

// relative safety margin of the certificate test (sweep_point, k_late): the float evaluations of the three distances in it are
// each within 3e-7 relative of the true values, so 1e-5 leaves a factor of 30
static float cert_margin() {
  static const float m = []() { const char* e = getenv("LH_CERT_REL"); float v = e ? (float)atof(e) : 1e-5f; return (v >= 1e-6f && v <= 1e-3f) ? v : 1e-5f; }();
  return m;
}
static void split_jobs(const SweepArgs& a, uint32_t split_mask, SweepArgs& f, SweepArgs& sp) {
  f = a; sp = a;
  f.njobs = 0; sp.njobs = 0;
  for (int j = 0; j < a.njobs; j++) {
    if ((split_mask >> j) & 1u) sp.job[sp.njobs++] = a.job[j];
    else f.job[f.njobs++] = a.job[j];
  }
}
int sweep_walk_span() {
  static const int span = []() { const char* e = getenv("LH_WALK_SPAN"); int v = e ? atoi(e) : 512; v = v >= 512 ? 512 : 256; return v; }();
  return span;
}
int sweep_fused_wg() { return FUSED_WG; }
// the all-walk sweeps of cost_mode 1 with covariances from normals: k_sweep_fused, or k_sweep_coop with LH_SWEEP_COOP=1 (measured slower:
// see the comment above the kernel; kept selectable for A/B runs and held to the same parity tests)
bool sweep_coop(bool normals_only) {
  static const bool on = []() { const char* e = getenv("LH_SWEEP_COOP"); return e ? atoi(e) != 0 : false; }();
  return on && normals_only;
}
// a pair's first sweeps start with the greedy descent: their candidate (a seed; a neighbour found before the first, largest step) is a loose bound
int sweep_coop_greedy_until() {
  static const int until = []() { const char* e = getenv("LH_COOP_GREEDY"); int v = e ? atoi(e) : 2; return v < 0 ? 0 : v; }();
  return until;
}
int sweep_greedy_flag(int sweep_index) { return sweep_index < sweep_coop_greedy_until() ? COOP_GREEDY_BIT : 0; }
int sweep_split_from() {
  static const int from = []() { const char* e = getenv("LH_SPLIT_FROM"); int v = e ? atoi(e) : 3; return v < 0 ? 0 : v; }();
  return from;
}

void launch_sweep_fused(const PairDesc* descs, SweepArgs& a, uint32_t split_mask, int max_n, double* partials_dev, int partials_stride,
                        const OuterState* states, bool normals_only, unsigned long long* wmask, int mask_stride, hipStream_t s) {
  static const int refill = []() { const char* e = getenv("LH_WALK_REFILL"); int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
  a.span = sweep_walk_span();
  a.cert_rel = cert_margin();
  const bool coop = sweep_coop(normals_only);
  for (int j = 0; j < a.njobs; j++) {
    const int flags = coop ? (a.job[j].pad & COOP_GREEDY_BIT) : 0;   // (only k_sweep_coop knows the bit)
    a.job[j].pad = ((a.job[j].pad & 0xffff) ? seed_group_size() : 0) | flags;   // a cold job: the sweep is told the seed pass's group size (sweep_point)
  }
  static const int exp_lanes = []() { const char* e = getenv("LH_EXP_LANES"); return e ? atoi(e) : 0; }();
  a.pad2 = exp_lanes;
  a.refill = refill;
  SweepArgs f, sp;
  split_jobs(a, wmask ? split_mask : 0u, f, sp);
  if (f.njobs > 0) {
    size_t lds = stack_lds_bytes(f.max_depth, FUSED_WG);
    if (lds < (FUSED_WG / 64) * 32 * GRAM_RS * sizeof(double)) lds = (FUSED_WG / 64) * 32 * GRAM_RS * sizeof(double);
    if (coop) {
      f.bpj = (max_n + 255) / 256;
      hipLaunchKernelGGL(k_sweep_coop, dim3(xcd_grid(f.njobs, f.bpj)), dim3(256), COOP_LDS_BYTES, s, descs, f, partials_dev, partials_stride, states);
    } else {
    f.bpj = (max_n + FUSED_WG - 1) / FUSED_WG;
    if (normals_only) hipLaunchKernelGGL(k_sweep_fused<true>, dim3(xcd_grid(f.njobs, f.bpj)), dim3(FUSED_WG), lds, s, descs, f, partials_dev, partials_stride, states);
    else hipLaunchKernelGGL(k_sweep_fused<false>, dim3(xcd_grid(f.njobs, f.bpj)), dim3(FUSED_WG), lds, s, descs, f, partials_dev, partials_stride, states);
    }
  }
  if (sp.njobs > 0) {
    sp.bpj = (max_n + 255) / 256;
    if (states)
      hipLaunchKernelGGL(k_late<true>, dim3(xcd_grid(sp.njobs, sp.bpj)), dim3(256), sizeof(double) * 4 * 32 * GRAM_RS, s, descs, sp, partials_dev, partials_stride, states,
                         wmask, mask_stride);
    else
      hipLaunchKernelGGL(k_late<false>, dim3(xcd_grid(sp.njobs, sp.bpj)), dim3(256), sizeof(double) * 4 * 32 * GRAM_RS, s, descs, sp, partials_dev, partials_stride, states,
                         wmask, mask_stride);
    sp.bpj = ((max_n + sp.span - 1) / sp.span + 3) / 4;
    hipLaunchKernelGGL(k_walk, dim3(xcd_grid(sp.njobs, sp.bpj)), dim3(256), 0, s, descs, sp, partials_dev, partials_stride, states, wmask, mask_stride);
  }
}
void launch_sweep(const PairDesc* descs, SweepArgs& a, int max_n, hipStream_t s) {
  a.cert_rel = cert_margin();
  for (int j = 0; j < a.njobs; j++)
    if (a.job[j].pad) a.job[j].pad = seed_group_size();
  a.bpj = (max_n + 255) / 256;
  hipLaunchKernelGGL(k_sweep, dim3(xcd_grid(a.njobs, a.bpj)), dim3(256), stack_lds_bytes(a.max_depth, 256), s, descs, a);
}

// ===== K5: cost / gradient reduction =======================================================================
__global__ void __launch_bounds__(256) k_cost(const PairDesc* __restrict__ descs, CostArgs a, double* __restrict__ out, int out_stride) {
  const CostJob& job = a.job[blockIdx.y];
  const PairDesc d = descs[job.slot];
  int base = blockIdx.x * COST_CHUNK;
  if (base >= d.n) return;
  double acc[COST_NSUM];
#pragma unroll
  for (int k = 0; k < COST_NSUM; k++) acc[k] = 0.0;
#pragma unroll
  for (int r = 0; r < COST_CHUNK / 256; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < d.n) {
      float4 c = d.corr[i];
      if (__float_as_int(c.w) >= 0) {
        float4 p = d.src[i];
        float px, py, pz;
        xform_pt(job.T, p.x, p.y, p.z, px, py, pz);                       // gicp.hpp:382
        double r0 = (double)(px - c.x), r1 = (double)(py - c.y), r2 = (double)(pz - c.z);  // :384 (float subtract)
        double m00 = d.maha6[(size_t)0 * d.n_pad + i], m01 = d.maha6[(size_t)1 * d.n_pad + i],
               m02 = d.maha6[(size_t)2 * d.n_pad + i], m11 = d.maha6[(size_t)3 * d.n_pad + i],
               m12 = d.maha6[(size_t)4 * d.n_pad + i], m22 = d.maha6[(size_t)5 * d.n_pad + i];
        double t0 = (m00 * r0 + m01 * r1) + m02 * r2;                     // temp = M*res  :386
        double t1 = (m01 * r0 + m11 * r1) + m12 * r2;
        double t2 = (m02 * r0 + m12 * r1) + m22 * r2;
        acc[0] += (r0 * t0 + r1 * t1) + r2 * t2;                          // :388
        acc[1] += t0; acc[2] += t1; acc[3] += t2;                         // :392
        double p0 = p.x, p1 = p.y, p2 = p.z;                              // :393-394 (base_transformation_ = I)
        acc[4] += p0 * t0; acc[5] += p0 * t1; acc[6] += p0 * t2;          // :396
        acc[7] += p1 * t0; acc[8] += p1 * t1; acc[9] += p1 * t2;
        acc[10] += p2 * t0; acc[11] += p2 * t1; acc[12] += p2 * t2;
      }
      // the correspondence count; a query without a neighbour (index -2, gicp.hpp:471-478) adds the failure mark instead: see nn_index
      acc[13] += __float_as_int(c.w) >= 0 ? 1.0 : (__float_as_int(c.w) == -2 ? NO_NN_MARK : 0.0);
    }
  }
  // fixed-shape reduction: wave64 shuffle tree, then the 4 waves in order => bitwise reproducible
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < COST_NSUM; k++) acc[k] += __shfl_down(acc[k], off, 64);
  }
  __shared__ double sm[4][COST_NSUM];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < COST_NSUM; k++) sm[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < COST_NSUM) {
    double v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    out[(size_t)job.slot * out_stride + blockIdx.x * COST_NSUM + threadIdx.x] = v;
  }
}
// The block sums of a job added in BLOCK ORDER, one after the other -- the order the host used while the partial sums still crossed
// PCIe (22 KB per pair and evaluation, 90 k host additions per 32-pair launch: the reference-arithmetic mode was bound by exactly
// that): the same 14 numbers, bit for bit, now 112 bytes per pair in pinned host memory.
__global__ void __launch_bounds__(64) k_cost_final(const PairDesc* __restrict__ descs, CostArgs a, const double* __restrict__ part, int part_stride,
                                                   double* __restrict__ out) {
  const CostJob& job = a.job[blockIdx.x];
  const int k = threadIdx.x;
  if (k >= COST_NSUM) return;
  const int nb = cost_blocks(descs[job.slot].n);
  const double* p = part + (size_t)job.slot * part_stride + k;
  double s = 0.0;
  int b = 0;
  for (; b + 8 <= nb; b += 8) {   // eight loads in flight, added in order
    double v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = p[(size_t)(b + e) * COST_NSUM];
#pragma unroll
    for (int e = 0; e < 8; e++) s += v[e];
  }
  for (; b < nb; b++) s += p[(size_t)b * COST_NSUM];
  out[job.out_offset + k] = s;
}

void launch_cost(const PairDesc* descs, const CostArgs& a, int max_n, double* partials_dev, int partials_stride, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_cost, dim3(cost_blocks(max_n), a.njobs), dim3(256), 0, s, descs, a, partials_dev, partials_stride);
  hipLaunchKernelGGL(k_cost_final, dim3(a.njobs), dim3(64), 0, s, descs, a, (const double*)partials_dev, partials_stride, out);
}

// ===== K5': second-order moments of the cost about T0 =====================================================
// res(T) = a + D*[p;1] with a = T0*p - q (double), D = T - T0 (3x4).  Then
//   sum res^T M res = c0 + 2 d.B + d.(H d),  B[r][c] = sum (M a)_r pt_c,  H[(r,s)][(c,e)] = sum M_rs pt_c pt_e,  pt = (p,1)
// and sum (M res)_r pt_c = B[r][c] + (H d)[r][c] gives the translation gradient and the rotation-gradient matrix R of
// gicp.hpp:392-396.  74 doubles per workgroup, fixed-shape reduction.
__global__ void __launch_bounds__(256) k_moments(const PairDesc* __restrict__ descs, CostArgs a, double* __restrict__ partials,
                                                 int partials_stride) {
  const CostJob& job = a.job[blockIdx.y];
  const PairDesc d = descs[job.slot];
  int base = blockIdx.x * MOM_CHUNK;
  if (base >= d.n) return;
  double acc[MOM_NSUM];
#pragma unroll
  for (int k = 0; k < MOM_NSUM; k++) acc[k] = 0.0;
  double T0[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T0[k] = (double)job.T[k];
  // software-pipelined: the next point's 80 bytes are requested before the current point's ~110 double FMAs, so the
  // single resident wave per SIMD (148 accumulator VGPRs) still overlaps HBM latency with arithmetic
  struct Raw { float4 c, p; double m[6]; bool ok, nonn; };
  auto load = [&](int i) {
    Raw r;
    r.ok = false;
    r.nonn = false;
    r.c = make_float4(0.f, 0.f, 0.f, 0.f); r.p = r.c;
#pragma unroll
    for (int k = 0; k < 6; k++) r.m[k] = 0.0;
    if (i < d.n) {
      r.c = d.corr[i];
      r.ok = __float_as_int(r.c.w) >= 0;
      r.nonn = __float_as_int(r.c.w) == -2;
      r.p = d.src[i];
      if (r.ok) {
#pragma unroll
        for (int k = 0; k < 6; k++) r.m[k] = d.maha6[(size_t)k * d.n_pad + i];
      }
    }
    return r;
  };
  Raw cur = load(base + threadIdx.x);
#pragma unroll 1
  for (int r = 0; r < MOM_CHUNK / 256; r++) {
    Raw nxt = load(base + (r + 1) * 256 + threadIdx.x < base + MOM_CHUNK ? base + (r + 1) * 256 + threadIdx.x : d.n);
    if (cur.ok) {
      double pt[4] = {(double)cur.p.x, (double)cur.p.y, (double)cur.p.z, 1.0};
      double a0 = (((T0[0] * pt[0] + T0[1] * pt[1]) + T0[2] * pt[2]) + T0[3]) - (double)cur.c.x;
      double a1 = (((T0[4] * pt[0] + T0[5] * pt[1]) + T0[6] * pt[2]) + T0[7]) - (double)cur.c.y;
      double a2 = (((T0[8] * pt[0] + T0[9] * pt[1]) + T0[10] * pt[2]) + T0[11]) - (double)cur.c.z;
      const double* M6 = cur.m;
      double Ma[3] = {(M6[0] * a0 + M6[1] * a1) + M6[2] * a2, (M6[1] * a0 + M6[3] * a1) + M6[4] * a2,
                      (M6[2] * a0 + M6[4] * a1) + M6[5] * a2};
      acc[0] += (a0 * Ma[0] + a1 * Ma[1]) + a2 * Ma[2];
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) acc[1 + rr * 4 + cc] += Ma[rr] * pt[cc];
      double pp[10];
      {
        int t = 0;
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
#pragma unroll
          for (int ee = cc; ee < 4; ee++) pp[t++] = pt[cc] * pt[ee];
      }
#pragma unroll
      for (int rs = 0; rs < 6; rs++)
#pragma unroll
        for (int ce = 0; ce < 10; ce++) acc[13 + rs * 10 + ce] += M6[rs] * pp[ce];
      acc[73] += 1.0;
    } else if (cur.nonn) {
      acc[73] += NO_NN_MARK;
    }
    cur = nxt;
  }
  // workgroup reduction through LDS in rounds of 16 values: [16][256] staging, fixed tree 256 -> 16 -> 1 per value
  // (a 74-value shuffle tree costs ~900 ds_bpermute per wave; this costs ~5 x 48 LDS ops per thread)
  __shared__ double st[16][256];
  __shared__ double st2[16][16];
  const int tid = threadIdx.x;
#pragma unroll
  for (int g = 0; g < (MOM_NSUM + 15) / 16; g++) {
#pragma unroll
    for (int v = 0; v < 16; v++)
      if (g * 16 + v < MOM_NSUM) st[v][tid] = acc[g * 16 + v];
    __syncthreads();
    {
      int v = tid >> 4, part = tid & 15;
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; j++) s += st[v][j * 16 + part];
      st2[v][part] = s;
    }
    __syncthreads();
    if (tid < 16 && g * 16 + tid < MOM_NSUM) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; j++) s += st2[tid][j];
      partials[(size_t)job.slot * partials_stride + (size_t)blockIdx.x * MOM_ROW + g * 16 + tid] = s;
    }
    __syncthreads();
  }
  if (tid < MOM_ROW - MOM_NSUM) partials[(size_t)job.slot * partials_stride + (size_t)blockIdx.x * MOM_ROW + MOM_NSUM + tid] = 0.0;
}

// final sum of a job's per-wave / per-workgroup partial rows (MOM_ROW doubles each), in two fixed-order stages: workgroup
// (chunk c, job) adds the rows of chunk c (4 strided sub-sums per column, combined in sub order) and writes MOM_ROW chunk
// sums into the job's slot of the pinned host buffer; the host adds the FINAL_CHUNKS chunk sums in chunk order => bitwise
// reproducible.  (One workgroup per job took 23 us per sweep -- 15 % of a late sweep: 32 workgroups cannot pull 30 MB quickly.)
// rows of a job = ceil(n / ppb) * rpb  (ppb points per workgroup of the producing kernel, rpb rows per workgroup)
constexpr int FINAL_SUB = 4;
__global__ void __launch_bounds__(FINAL_SUB * MOM_ROW) k_moments_final(const PairDesc* __restrict__ descs, CostArgs a, const double* __restrict__ partials,
                                                                      int partials_stride, int ppb, int rpb, int extra_ppr, double* __restrict__ out,
                                                                      const OuterState* __restrict__ states, unsigned long long* __restrict__ wmask, int mask_stride) {
  const CostJob& job = a.job[blockIdx.y];
  if (states && states[job.slot].done) return;  // device-driven loop: the pair's sweep did not run either
  const int c = blockIdx.x;
  int n = descs[job.slot].n;
  if ((a.pad >> blockIdx.y) & 1) ppb = FUSED_WG;   // this job's rows were left by the fused sweep (a.pad: one bit per job)
  int nb = ((n + ppb - 1) / ppb) * rpb;
  if (extra_ppr > 0) nb += (n + extra_ppr - 1) / extra_ppr;   // the fused / split sweep's walk rows (one per extra_ppr source points)
  int v = threadIdx.x % MOM_ROW, sub = threadIdx.x / MOM_ROW;
  int per = (nb + FINAL_CHUNKS - 1) / FINAL_CHUNKS;
  int b0 = c * per, b1 = min(nb, b0 + per);
  const double* p = partials + (size_t)job.slot * partials_stride + v;
  double s = 0.0;
  int b = b0 + sub;
  for (; b + 7 * FINAL_SUB < b1; b += 8 * FINAL_SUB) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = p[(size_t)(b + k * FINAL_SUB) * MOM_ROW];
#pragma unroll
    for (int k = 0; k < 8; k++) s += t[k];
  }
  for (; b < b1; b += FINAL_SUB) s += p[(size_t)b * MOM_ROW];
  __shared__ double sm[FINAL_SUB][MOM_ROW];
  sm[sub][v] = s;
  __syncthreads();
  if (sub == 0) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < FINAL_SUB; k++) tot += sm[k][v];
    out[job.out_offset + c * MOM_ROW + v] = tot;
  }
}

void launch_moments(const PairDesc* descs, const CostArgs& a, int max_n, double* partials_dev, int partials_stride, double* out,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_moments, dim3(mom_blocks(max_n), a.njobs), dim3(256), 0, s, descs, a, partials_dev, partials_stride);
  hipLaunchKernelGGL(k_moments_final, dim3(FINAL_CHUNKS, a.njobs), dim3(FINAL_SUB * MOM_ROW), 0, s, descs, a, partials_dev, partials_stride, MOM_CHUNK, 1, 0, out,
                     (const OuterState*)nullptr, (unsigned long long*)nullptr, 0);
}
void launch_moments_final(const PairDesc* descs, const CostArgs& a, double* partials_dev, int partials_stride, double* out, const OuterState* states,
                          unsigned long long* wmask, int mask_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_moments_final, dim3(FINAL_CHUNKS, a.njobs), dim3(FINAL_SUB * MOM_ROW), 0, s, descs, a, partials_dev, partials_stride, 256, 1, sweep_walk_span(), out, states,
                     wmask, mask_stride);
}

// ===== the solve of one outer iteration on the device (cost_mode 1) =========================================
// One wave per pair runs the part of computeTransformation's loop body that follows the sweep (gicp.hpp:518-568): the whole BFGS
// solve on the pair's 74-moment model, the convergence test, and the update of the pair's state (transformation_, iteration
// count, done flag) that the next sweep launch reads -- the host is not in the loop.  The code is lh_bfgs.hpp's, the same
// templates the host instantiates for the source-sharded pair, with PortableMath (lh_math.hpp): identical bits on both sides.
// Every lane executes the (uniform) control flow; the model lives in LDS.
__global__ void __launch_bounds__(64) k_solve(const PairDesc* __restrict__ descs, SolveArgs a, const double* __restrict__ chunks, int chunk_stride,
                                              OuterState* __restrict__ states) {
  const int slot = a.slot[blockIdx.x];
  OuterState* sp = states + slot;
  if (sp->done) return;
  __shared__ MomentModel mom;
  const int lane = threadIdx.x;
  const double* part = chunks + (size_t)slot * chunk_stride;  // FINAL_CHUNKS x MOM_ROW chunk sums of this iteration's sweep
  for (int k = lane; k < MOM_NSUM; k += 64) {
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < FINAL_CHUNKS; ch++) s += part[ch * MOM_ROW + k];  // chunk order, like the host: bitwise the same sums
    mom.S[k] = s;
  }
  if (lane < 16) mom.T0[lane] = sp->T[lane];  // the transform the sweep used
  __syncthreads();
  for (int e = lane; e < 144; e += 64) mom.H12[e] = mom.S[MomentModel::h_index(e / 12, e % 12)];
  __syncthreads();
  const PairDesc* d = descs + slot;
  OuterParams P{d->max_iterations, d->max_inner_iterations, d->rotation_epsilon, d->transformation_epsilon, d->bfgs_quad_curv};
  OuterState s = *sp;
  typedef MomentPass<PortableMath> Pass;
  typedef CostEval<Pass, PortableMath> Fn;
  Pass pass;
  pass.mom = &mom;
  Fn fn;
  fn.pass = pass;
  const int before = s.passes, it = s.iter;
  outer_step<Fn, PortableMath>(&fn, P, &s);
  s.corr_sum += mom.count();
  {  // the next sweep's transform as doubles (k_late reads them with scalar loads)
    float T12[12];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) T12[r * 4 + c] = s.T[c * 4 + r];
    pose_of_transform(T12, s.poseT, s.poseG);
  }
  if (lane == 0) {
    *sp = s;
    lh_gicp_trace* tr = d->trace;
    if (tr && s.status == 0 && it < LH_MAX_TRACE) {  // the per-iteration trace of lh_gicp_align (parity tests)
#pragma unroll
      for (int k = 0; k < 16; k++) tr->T[it][k] = s.T[k];
      tr->n_corr[it] = s.n_corr_last;
      tr->n_passes[it] = s.passes - before;
      tr->n_inner[it] = s.n_inner;
      tr->f_end[it] = s.f_end;
      tr->delta[it] = s.delta;
      tr->n_iters = it + 1;
    }
  }
}
void launch_solve(const PairDesc* descs, const SolveArgs& a, const double* chunks, int chunk_stride, OuterState* states, hipStream_t s) {
  hipLaunchKernelGGL(k_solve, dim3(a.njobs), dim3(64), 0, s, descs, a, chunks, chunk_stride, states);
}

// ===== K6 / misc ===========================================================================================
struct T12 { float v[12]; };
__global__ void __launch_bounds__(256) k_transform(const float4* __restrict__ in_xyz, const float4* __restrict__ in_nrm, int n, T12 T,
                                                   float4* __restrict__ out_xyz, float4* __restrict__ out_nrm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in_xyz[i];
  float x, y, z;
  xform_pt(T.v, p.x, p.y, p.z, x, y, z);
  out_xyz[i] = make_float4(x, y, z, 1.0f);
  if (in_nrm && out_nrm) {
    float4 nn = in_nrm[i];
    xform_nrm(T.v, nn.x, nn.y, nn.z, x, y, z);
    out_nrm[i] = make_float4(x, y, z, nn.w);
  }
}
void launch_transform(const float4* in_xyz, const float4* in_nrm, int n, const float* T12p, float4* out_xyz, float4* out_nrm,
                      hipStream_t s) {
  T12 T;
  for (int k = 0; k < 12; k++) T.v[k] = T12p[k];
  hipLaunchKernelGGL(k_transform, dim3((n + 255) / 256), dim3(256), 0, s, in_xyz, in_nrm, n, T, out_xyz, out_nrm);
}

// align()'s output cloud (gicp.hpp:586, pcl::transformPointCloud(*input_, output, final_transformation_)): xyz transformed, every
// other field of the point copied -- one launch per pair
__global__ void __launch_bounds__(256) k_transform_copy(const float4* __restrict__ in_xyz, const float4* __restrict__ in_nrm, const float* __restrict__ in_int,
                                                        int n, T12 T, float4* __restrict__ out_xyz, float4* __restrict__ out_nrm, float* __restrict__ out_int) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in_xyz[i];
  float x, y, z;
  xform_pt(T.v, p.x, p.y, p.z, x, y, z);
  out_xyz[i] = make_float4(x, y, z, 1.0f);
  if (in_nrm && out_nrm) out_nrm[i] = in_nrm[i];
  if (in_int && out_int) out_int[i] = in_int[i];
}
// the same for all pairs that retire together: one launch, grid.y = pair
__global__ void __launch_bounds__(256) k_transform_copy_batch(XformBatchArgs a) {
  const XformJob& j = a.job[blockIdx.y];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= j.n) return;
  float4 p = j.in_xyz[i];
  float x, y, z;
  xform_pt(j.T, p.x, p.y, p.z, x, y, z);
  j.out_xyz[i] = make_float4(x, y, z, 1.0f);
  if (j.in_nrm && j.out_nrm) j.out_nrm[i] = j.in_nrm[i];
  if (j.in_int && j.out_int) j.out_int[i] = j.in_int[i];
}
void launch_transform_copy_batch(const XformBatchArgs& a, int max_n, hipStream_t s) {
  hipLaunchKernelGGL(k_transform_copy_batch, dim3((max_n + 255) / 256, a.njobs), dim3(256), 0, s, a);
}
void launch_transform_copy(const float4* in_xyz, const float4* in_nrm, const float* in_int, int n, const float* T12p, float4* out_xyz,
                           float4* out_nrm, float* out_int, hipStream_t s) {
  T12 T;
  for (int k = 0; k < 12; k++) T.v[k] = T12p[k];
  hipLaunchKernelGGL(k_transform_copy, dim3((n + 255) / 256), dim3(256), 0, s, in_xyz, in_nrm, in_int, n, T, out_xyz, out_nrm, out_int);
}

// A host point array (pcl::PointXYZI / PointXYZINormal / any stride with float fields, lh_cloud_view) copied to the device AS IT IS and
// taken apart here into the path's layout (xyz1 | normal + curvature | intensity): the host never touches the points (a scalar repack
// loop + three staged copies cost ~0.45 ms per 100 k-point scan: the PCIe-inclusive rate was bound by the host, not by PCIe).
__global__ void __launch_bounds__(256) k_unpack_view(const unsigned char* __restrict__ raw, int n, uint32_t stride, uint32_t off_xyz, uint32_t off_normal,
                                                     uint32_t off_intensity, uint32_t off_curvature, float4* __restrict__ xyz, float4* __restrict__ nrm,
                                                     float* __restrict__ inten) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* p = raw + (size_t)i * stride;
  auto f = [&](uint32_t off) { float v; __builtin_memcpy(&v, p + off, 4); return v; };   // (fields are 4-byte aligned in every PCL point type; memcpy keeps odd strides legal)
  xyz[i] = make_float4(f(off_xyz), f(off_xyz + 4), f(off_xyz + 8), 1.0f);
  if (nrm) nrm[i] = make_float4(f(off_normal), f(off_normal + 4), f(off_normal + 8), off_curvature != 0xffffffffu ? f(off_curvature) : 0.0f);
  if (inten) inten[i] = f(off_intensity);
}
void launch_unpack_view(const void* raw, int n, uint32_t stride, uint32_t off_xyz, uint32_t off_normal, uint32_t off_intensity, uint32_t off_curvature,
                        float4* xyz, float4* nrm, float* inten, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack_view, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned char*)raw, n, stride, off_xyz, off_normal, off_intensity, off_curvature,
                     xyz, nrm, inten);
}

__global__ void __launch_bounds__(256) k_fill_i32(int32_t* p, int n, int32_t v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill_i32(int32_t* p, int n, int32_t v, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, s, p, n, v);
}

__global__ void __launch_bounds__(256) k_nn1(const float4* __restrict__ q, int nq, T12 T, int has_T, TreeView tv,
                                             int32_t* __restrict__ idx, float* __restrict__ d2) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_stack[];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  float4 p = q[i];
  float x = p.x, y = p.y, z = p.z;
  if (has_T) xform_pt(T.v, p.x, p.y, p.z, x, y, z);
  Nn1Collector col{INFINITY, 0x7fffffff};
  tree_search<Nn1Collector, true>(tv, x, y, z, col, lds_stack + threadIdx.x, 256);   // (a descent below the query's own grid cell for a bound, then the walk from there)
  idx[i] = nn_index(col.bi, col.bd);   // a non-finite query has no neighbour (pcl::KdTreeFLANN: isValid(query))
  d2[i] = col.bd;
}
void launch_nn1(const float4* q, int nq, const float* T12p, TreeView tree, int32_t* idx, float* d2, hipStream_t s) {
  T12 T;
  for (int k = 0; k < 12; k++) T.v[k] = T12p ? T12p[k] : 0.f;
  hipLaunchKernelGGL(k_nn1, dim3((nq + 255) / 256), dim3(256), stack_lds_bytes(0, 256), s, q, nq, T, T12p ? 1 : 0, tree, idx, d2);
}

__global__ void __launch_bounds__(256) k_nn1_stats(const float4* __restrict__ q, int nq, T12 T, int has_T, TreeView tv,
                                                   const float4* __restrict__ tgt_xyz, const int32_t* __restrict__ cand,
                                                   unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_stack[];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int nodes = 0, leaves = 0;
  if (i < nq) {
    float4 p = q[i];
    float x = p.x, y = p.y, z = p.z;
    if (has_T) xform_pt(T.v, p.x, p.y, p.z, x, y, z);
    Nn1CountCollector col{INFINITY, 0x7fffffff, 0, 0};
    for (int l = 0; l < MAX_DEPTH; l++) col.per_level[l] = 0;
    if (cand && cand[i] >= 0) {
      int w = cand[i];
      float4 t = tgt_xyz[w];
      col.bd = d2f(x, y, z, t.x, t.y, t.z);
      col.bi = w;
    }
    tree_search(tv, x, y, z, col, lds_stack + threadIdx.x, 256);
    nodes = col.nodes; leaves = col.leaves;
    for (int l = 0; l < MAX_DEPTH; l++) atomicAdd(&stats[8 + l], (unsigned long long)col.per_level[l]);
  }
  int tot = nodes + leaves, mx = tot, sn = nodes, sl = leaves;
  for (int off = 32; off > 0; off >>= 1) {
    mx = max(mx, __shfl_down(mx, off, 64));
    sn += __shfl_down(sn, off, 64);
    sl += __shfl_down(sl, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&stats[0], (unsigned long long)sn);
    atomicAdd(&stats[1], (unsigned long long)sl);
    atomicAdd(&stats[2], (unsigned long long)mx);
    atomicAdd(&stats[3], 1ull);
    atomicMax(&stats[4], (unsigned long long)mx);
  }
}
void launch_nn1_stats(const float4* q, int nq, const float* T12p, TreeView tree, const float4* tgt_xyz, const int32_t* cand,
                      unsigned long long* stats, hipStream_t s) {
  T12 T;
  for (int k = 0; k < 12; k++) T.v[k] = T12p ? T12p[k] : 0.f;
  hipLaunchKernelGGL(k_nn1_stats, dim3((nq + 255) / 256), dim3(256), stack_lds_bytes(0, 256), s, q, nq, T, T12p ? 1 : 0, tree, tgt_xyz, cand, stats);
}

// double sum of the float d2 of the queries that found a neighbour (idx >= 0) + their number: 1024 values per block, fixed
// tree.  partials[2 b] = sum, partials[2 b + 1] = count (getFitnessScore only averages over the matched queries)
__global__ void __launch_bounds__(256) k_sum_f32(const float* __restrict__ v, const int32_t* __restrict__ idx, int n, double* __restrict__ partials) {
  int base = blockIdx.x * 1024;
  double acc = 0.0, cnt = 0.0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n && idx[i] >= 0) { acc += (double)v[i]; cnt += 1.0; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { acc += __shfl_down(acc, off, 64); cnt += __shfl_down(cnt, off, 64); }
  __shared__ double sm[4][2];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = acc; sm[threadIdx.x >> 6][1] = cnt; }
  __syncthreads();
  if (threadIdx.x < 2) partials[2 * blockIdx.x + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}
void launch_sum_f32(const float* v, const int32_t* idx, int n, double* partials, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_f32, dim3(sum_blocks(n)), dim3(256), 0, s, v, idx, n, partials);
}

// (K3 -- k-NN lists, covariances, normals -- lives in lh_knn.hip)

// removeNaNNormalsFromPointCloud (normal_computation.cc:52-56): order-preserving compaction of points with finite normals
__global__ void __launch_bounds__(256) k_finite_normal_flags(const float4* __restrict__ nrm, int n, uint32_t* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 v = nrm[i];
  flags[i] = (isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_compact(const uint32_t* __restrict__ incl, int n, const float4* __restrict__ xyz,
                                                 const float4* __restrict__ nrm, const float* __restrict__ inten, float4* __restrict__ oxyz,
                                                 float4* __restrict__ onrm, float* __restrict__ ointen) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t here = incl[i], before = i ? incl[i - 1] : 0u;
  if (here == before) return;
  oxyz[before] = xyz[i];
  onrm[before] = nrm[i];
  if (inten) ointen[before] = inten[i];
}
void launch_finite_normal_flags(const float4* nrm, int n, uint32_t* flags, hipStream_t s) {
  hipLaunchKernelGGL(k_finite_normal_flags, dim3((n + 255) / 256), dim3(256), 0, s, nrm, n, flags);
}
void launch_compact(const uint32_t* incl, int n, const float4* xyz, const float4* nrm, const float* inten, float4* oxyz, float4* onrm,
                    float* ointen, hipStream_t s) {
  hipLaunchKernelGGL(k_compact, dim3((n + 255) / 256), dim3(256), 0, s, incl, n, xyz, nrm, inten, oxyz, onrm, ointen);
}

// ===== NDT (registration_method: ndt) ==========================================================================
// target side: per-voxel raw statistics in input order (VoxelGridCovariance::applyFilter, voxel_grid_covariance_omp_impl.hpp:
// 166-205): double sums of p and p p^T, float centroid sum, count (then k_ndt_finish_cells, one thread per voxel)
__global__ void __launch_bounds__(256) k_ndt_voxel_stats(const float4* __restrict__ xyz, const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, const uint32_t* __restrict__ heads,
                                                         const uint32_t* __restrict__ rank, int n, NdtVoxelRaw* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !heads[i]) return;
  uint32_t k = keys[i];
  NdtVoxelRaw r;
  r.count = 0;
  r.cen[0] = r.cen[1] = r.cen[2] = 0.0f;
#pragma unroll
  for (int a = 0; a < 3; a++) r.sum[a] = 0.0;
#pragma unroll
  for (int a = 0; a < 6; a++) r.cov[a] = 0.0;
  for (int j = i; j < n && keys[j] == k; j++) {  // stable radix sort => ascending point index inside a voxel
    float4 p = xyz[vals[j]];
    double x = p.x, y = p.y, z = p.z;
    r.sum[0] += x; r.sum[1] += y; r.sum[2] += z;
    r.cov[0] += x * x; r.cov[1] += x * y; r.cov[2] += x * z; r.cov[3] += y * y; r.cov[4] += y * z; r.cov[5] += z * z;
    r.cen[0] += p.x; r.cen[1] += p.y; r.cen[2] += p.z;
    r.count++;
  }
  out[rank[i] - 1u] = r;
}
void launch_ndt_voxel_stats(const float4* xyz, const uint32_t* keys, const uint32_t* vals, const uint32_t* heads, const uint32_t* rank_incl,
                            int n, NdtVoxelRaw* out, hipStream_t s) {
  hipLaunchKernelGGL(k_ndt_voxel_stats, dim3((n + 255) / 256), dim3(256), 0, s, xyz, keys, vals, heads, rank_incl, n, out);
}

// per voxel: raw sums -> cell (mean, inverse covariance, float centroid) + "becomes a cell" flag (>= min_points points)
__global__ void __launch_bounds__(256) k_ndt_finish_cells(const NdtVoxelRaw* __restrict__ raw, int n_vox, int min_points, double eig_mult,
                                                          double* __restrict__ mean, double* __restrict__ icov, float4* __restrict__ cen,
                                                          uint32_t* __restrict__ flags) {
  int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n_vox) return;
  NdtVoxelRaw r = raw[v];
  double m3[3], ic[9];
  bool cell = ndt_finish_cell(r.sum, r.cov, r.count, min_points, eig_mult, m3, ic);
  flags[v] = cell ? 1u : 0u;
  if (!cell) return;
#pragma unroll
  for (int k = 0; k < 3; k++) mean[3 * (size_t)v + k] = m3[k];
#pragma unroll
  for (int k = 0; k < 9; k++) icov[9 * (size_t)v + k] = ic[k];
  float c = (float)r.count;
  cen[v] = make_float4(r.cen[0] / c, r.cen[1] / c, r.cen[2] / c, 1.0f);
}
// keep the flagged voxels, in ascending voxel order (the order of VoxelGridCovariance's centroid cloud / kd-tree)
__global__ void __launch_bounds__(256) k_ndt_compact_cells(const uint32_t* __restrict__ incl, int n_vox, const double* __restrict__ mean,
                                                           const double* __restrict__ icov, const float4* __restrict__ cen,
                                                           double* __restrict__ omean, double* __restrict__ oicov, float4* __restrict__ ocen) {
  int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n_vox) return;
  uint32_t here = incl[v], before = v ? incl[v - 1] : 0u;
  if (here == before) return;
#pragma unroll
  for (int k = 0; k < 3; k++) omean[3 * (size_t)before + k] = mean[3 * (size_t)v + k];
#pragma unroll
  for (int k = 0; k < 9; k++) oicov[9 * (size_t)before + k] = icov[9 * (size_t)v + k];
  ocen[before] = cen[v];
}
void launch_ndt_finish_cells(const NdtVoxelRaw* raw, int n_vox, int min_points, double eig_mult, double* mean, double* icov, float4* cen,
                             uint32_t* flags, hipStream_t s) {
  hipLaunchKernelGGL(k_ndt_finish_cells, dim3((n_vox + 255) / 256), dim3(256), 0, s, raw, n_vox, min_points, eig_mult, mean, icov, cen, flags);
}
void launch_ndt_compact_cells(const uint32_t* incl, int n_vox, const double* mean, const double* icov, const float4* cen, double* omean,
                              double* oicov, float4* ocen, hipStream_t s) {
  hipLaunchKernelGGL(k_ndt_compact_cells, dim3((n_vox + 255) / 256), dim3(256), 0, s, incl, n_vox, mean, icov, cen, omean, oicov, ocen);
}

// source side: computeDerivatives (MODE 0, float point derivatives) / computeHessian (MODE 1, double).  One source point per
// thread; the radius search over the voxel CENTROIDS (KDTREE mode, voxel_grid_covariance_omp.h:433-466) walks the cells'
// radix tree and every cell inside the radius contributes on the spot; 43 doubles per wave are reduced through LDS.
template <int MODE>
struct NdtCollector {
  float r2;
  const NdtFrame* f;
  const double* mean;
  const double* icov;
  float x3[3], xt[3];
  double acc[NDT_NSUM];
  static constexpr bool kXyz = false;
  static constexpr bool kGreedy = false;
  __device__ __forceinline__ float bound() const { return r2; }
  __device__ __forceinline__ void offer(float d, int id) {
    if (!(d < r2) || id == 0x7fffffff) return;  // FLANN RadiusResultSet: strict
    const double* mu = mean + 3 * (size_t)id;
    double dd[3] = {(double)xt[0] - mu[0], (double)xt[1] - mu[1], (double)xt[2] - mu[2]};
    if (MODE == 0) ndt_term_float<true>(*f, x3, dd, icov + 9 * (size_t)id, acc);
    else if (MODE == 2) ndt_term_float<false>(*f, x3, dd, icov + 9 * (size_t)id, acc);   // line-search evaluations: score + gradient only
    else ndt_term_hessian_double(*f, x3, dd, icov + 9 * (size_t)id, acc + 7);
  }
  __device__ __forceinline__ void skip(float) {}
  __device__ __forceinline__ void count_node(int) {}
  __device__ __forceinline__ void count_leaf() {}
};
template <int MODE>
__global__ void __launch_bounds__(256) k_ndt_derivs(const float4* __restrict__ src, int n, TreeView cells, const double* __restrict__ mean,
                                                    const double* __restrict__ icov, NdtFrame f, double* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_stack[];
  const int i = blockIdx.x * 256 + threadIdx.x;
  NdtCollector<MODE> col;
  col.r2 = f.r2; col.f = &f; col.mean = mean; col.icov = icov;
#pragma unroll
  for (int k = 0; k < NDT_NSUM; k++) col.acc[k] = 0.0;
  if (i < n) {
    float4 p = src[i];
    col.x3[0] = p.x; col.x3[1] = p.y; col.x3[2] = p.z;
    xform_pt(f.T, p.x, p.y, p.z, col.xt[0], col.xt[1], col.xt[2]);  // transformPointCloud(*input_, trans_cloud, final_transformation_)
    tree_search(cells, col.xt[0], col.xt[1], col.xt[2], col, lds_stack + threadIdx.x, 256);
  }
  double* out = rows + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * NDT_ROW;
  if constexpr (MODE == 2) wave_reduce_row<7, NDT_ROW>(lds_stack, out, 0.0, [&](int k) { return col.acc[k]; });  // the hessian columns are zero
  else wave_reduce_row<NDT_NSUM, NDT_ROW>(lds_stack, out, 0.0, [&](int k) { return col.acc[k]; });
}
// fixed-order final sum of the per-wave rows: workgroup c adds chunk c into out[c][NDT_ROW]; the host adds the chunks
__global__ void __launch_bounds__(4 * NDT_ROW) k_rows_final(const double* __restrict__ rows, int n_rows, double* __restrict__ out) {
  const int c = blockIdx.x, v = threadIdx.x % NDT_ROW, sub = threadIdx.x / NDT_ROW;
  int per = (n_rows + FINAL_CHUNKS - 1) / FINAL_CHUNKS;
  int b0 = c * per, b1 = min(n_rows, b0 + per);
  double s = 0.0;
  for (int b = b0 + sub; b < b1; b += 4) s += rows[(size_t)b * NDT_ROW + v];
  __shared__ double sm[4][NDT_ROW];
  sm[sub][v] = s;
  __syncthreads();
  if (sub == 0) out[c * NDT_ROW + v] = ((sm[0][v] + sm[1][v]) + sm[2][v]) + sm[3][v];
}
void launch_ndt_derivs(const float4* src, int n, TreeView cells, const double* mean, const double* icov, const NdtFrame& f, int hessian_only,
                       double* rows_dev, double* out_chunks /*[FINAL_CHUNKS][NDT_ROW], device-visible*/, hipStream_t s) {
  size_t lds = stack_lds_bytes(0, 256);
  if (lds < 4 * 8 * 72 * sizeof(double)) lds = 4 * 8 * 72 * sizeof(double);
  int blocks = (n + 255) / 256;
  if (hessian_only) hipLaunchKernelGGL(k_ndt_derivs<1>, dim3(blocks), dim3(256), lds, s, src, n, cells, mean, icov, f, rows_dev);
  else if (f.want_h) hipLaunchKernelGGL(k_ndt_derivs<0>, dim3(blocks), dim3(256), lds, s, src, n, cells, mean, icov, f, rows_dev);
  else hipLaunchKernelGGL(k_ndt_derivs<2>, dim3(blocks), dim3(256), lds, s, src, n, cells, mean, icov, f, rows_dev);
  hipLaunchKernelGGL(k_rows_final, dim3(FINAL_CHUNKS), dim3(4 * NDT_ROW), 0, s, rows_dev, blocks * 4, out_chunks);
}

// ===== BodyFilter (body_filter.cc:27-52): pcl::CropBox with a yaw-rotated box, negative = keep what is OUTSIDE ================
// flags[i] = 1 if point i survives.  Non-finite points are dropped (CropBox skips them when keep_organized is off).
__global__ void __launch_bounds__(256) k_crop_flags(const float4* __restrict__ xyz, int n, float minx, float miny, float minz, float maxx,
                                                    float maxy, float maxz, float c, float s, int negative, uint32_t* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 p = xyz[i];
  uint32_t keep = 0;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    float lx = c * p.x + s * p.y, ly = c * p.y - s * p.x, lz = p.z;  // inverse of the box pose Rz(yaw)
    bool outside = lx < minx || ly < miny || lz < minz || lx > maxx || ly > maxy || lz > maxz;
    keep = (outside == (negative != 0)) ? 1u : 0u;
  }
  flags[i] = keep;
}
void launch_crop_flags(const float4* xyz, int n, const float* mn, const float* mx, float c, float s, int negative, uint32_t* flags, hipStream_t st) {
  hipLaunchKernelGGL(k_crop_flags, dim3((n + 255) / 256), dim3(256), 0, st, xyz, n, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], c, s, negative, flags);
}

// ===== local map: PointCloudMapper::InsertPoints / Refresh on the device =======================================
// occupancy key of a point = its voxel floor(p / resolution) packed 21 bits per axis (offset 2^20); ~0 = rejected
__device__ __forceinline__ uint64_t map_voxel_key(float4 p, double inv_res) {
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return ~0ull;
  double fx = floor((double)p.x * inv_res), fy = floor((double)p.y * inv_res), fz = floor((double)p.z * inv_res);
  const double lim = 1048575.0;
  if (fx < -lim || fx > lim || fy < -lim || fy > lim || fz < -lim || fz > lim) return ~0ull;
  uint64_t ix = (uint64_t)((long long)fx + 1048576), iy = (uint64_t)((long long)fy + 1048576), iz = (uint64_t)((long long)fz + 1048576);
  return (ix << 42) | (iy << 21) | iz;
}
__global__ void __launch_bounds__(256) k_map_keys(const float4* __restrict__ xyz, int n, double inv_res, uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = map_voxel_key(xyz[i], inv_res);
  if (vals) vals[i] = (uint32_t)i;
}
// sorted (key, input index) pairs -> accept[input index] = 1 for the FIRST point (lowest input index: the sort is stable) of
// every voxel that the map does not occupy yet (binary search in the map's sorted keys)
__global__ void __launch_bounds__(256) k_map_accept(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ svals, int n,
                                                    const uint64_t* __restrict__ map_keys, int m, uint32_t* __restrict__ accept) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t k = skeys[i];
  uint32_t ok = 0;
  if (k != ~0ull && (i == 0 || skeys[i - 1] != k)) {
    int lo = 0, hi = m;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (map_keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    ok = (lo < m && map_keys[lo] == k) ? 0u : 1u;
  }
  accept[svals[i]] = ok;
}
// Refresh (box filter around the current pose): keep points with |p - c|_inf <= half (pcl::CropBox bounds are inclusive)
__global__ void __launch_bounds__(256) k_box_flags(const float4* __restrict__ xyz, int n, float cx, float cy, float cz, float half,
                                                   uint32_t* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 p = xyz[i];
  flags[i] = (p.x >= cx - half && p.x <= cx + half && p.y >= cy - half && p.y <= cy + half && p.z >= cz - half && p.z <= cz + half) ? 1u : 0u;
}
// order-preserving compaction of (xyz, nrm, intensity) into dst arrays starting at dst_off; also emits the kept points' keys
__global__ void __launch_bounds__(256) k_map_compact(const uint32_t* __restrict__ incl, int n, const float4* __restrict__ xyz,
                                                     const float4* __restrict__ nrm, const float* __restrict__ inten, double inv_res, int dst_off,
                                                     float4* __restrict__ oxyz, float4* __restrict__ onrm, float* __restrict__ ointen,
                                                     uint64_t* __restrict__ okeys) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t here = incl[i], before = i ? incl[i - 1] : 0u;
  if (here == before) return;
  float4 p = xyz[i];
  int o = dst_off + (int)before;
  oxyz[o] = make_float4(p.x, p.y, p.z, 1.0f);
  if (onrm) onrm[o] = nrm ? nrm[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  if (ointen) ointen[o] = inten ? inten[i] : 0.0f;
  if (okeys) okeys[o] = map_voxel_key(p, inv_res);
}
void launch_map_keys(const float4* xyz, int n, double inv_res, uint64_t* keys, uint32_t* vals, hipStream_t s) {
  hipLaunchKernelGGL(k_map_keys, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, inv_res, keys, vals);
}
void launch_map_accept(const uint64_t* skeys, const uint32_t* svals, int n, const uint64_t* map_keys, int m, uint32_t* accept, hipStream_t s) {
  hipLaunchKernelGGL(k_map_accept, dim3((n + 255) / 256), dim3(256), 0, s, skeys, svals, n, map_keys, m, accept);
}
void launch_box_flags(const float4* xyz, int n, float cx, float cy, float cz, float half, uint32_t* flags, hipStream_t s) {
  hipLaunchKernelGGL(k_box_flags, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, cx, cy, cz, half, flags);
}
void launch_map_compact(const uint32_t* incl, int n, const float4* xyz, const float4* nrm, const float* inten, double inv_res, int dst_off,
                        float4* oxyz, float4* onrm, float* ointen, uint64_t* okeys, hipStream_t s) {
  hipLaunchKernelGGL(k_map_compact, dim3((n + 255) / 256), dim3(256), 0, s, incl, n, xyz, nrm, inten, inv_res, dst_off, oxyz, onrm, ointen, okeys);
}

// ===== K8: point-to-plane information matrix ===============================================================
// Ap = sum H^T H, H = [a x n, n]; 21 unique entries per block of 1024 points, fixed reduction shape
__global__ void __launch_bounds__(256) k_ap(const float4* __restrict__ qn, int n, const float4* __restrict__ ref_nrm,
                                            const int64_t* __restrict__ corr, double* __restrict__ partials) {
  double acc[21];
#pragma unroll
  for (int k = 0; k < 21; k++) acc[k] = 0.0;
  int base = blockIdx.x * 1024;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n) {
      float4 a4 = qn[i];
      float4 n4 = ref_nrm[corr[i]];
      double a0 = a4.x, a1 = a4.y, a2 = a4.z, n0 = n4.x, n1 = n4.y, n2 = n4.z;
      bool bad = (a0 != a0) || (a1 != a1) || (a2 != a2) || (n0 != n0) || (n1 != n1) || (n2 != n2);  // PointCloudLocalization.cc:742
      if (!bad) {
        double H[6] = {a1 * n2 - a2 * n1, a2 * n0 - a0 * n2, a0 * n1 - a1 * n0, n0, n1, n2};
        int t = 0;
#pragma unroll
        for (int rr = 0; rr < 6; rr++)
#pragma unroll
          for (int cc = rr; cc < 6; cc++) acc[t++] += H[rr] * H[cc];
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 21; k++) acc[k] += __shfl_down(acc[k], off, 64);
  }
  __shared__ double sm[4][21];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 21; k++) sm[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 21) partials[blockIdx.x * 21 + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}
void launch_ap(const float4* qnorm, int n, const float4* ref_nrm, const int64_t* corr, double* partials, hipStream_t s) {
  hipLaunchKernelGGL(k_ap, dim3(ap_blocks(n)), dim3(256), 0, s, qnorm, n, ref_nrm, corr, partials);
}

// ===== K1: voxel-grid centroid downsample (pcl::VoxelGrid semantics) =======================================
__device__ __forceinline__ bool voxel_accept(float4 p, int limit_axis, float lo, float hi) {
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return false;
  if (limit_axis >= 0) {
    float v = limit_axis == 0 ? p.x : (limit_axis == 1 ? p.y : p.z);
    if (v > hi || v < lo) return false;
  }
  return true;
}

__global__ void __launch_bounds__(256) k_voxel_bbox(const float4* __restrict__ xyzi, int n, int limit_axis, float lo_, float hi_, uint32_t* bbox) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = xyzi[i];
    if (voxel_accept(p, limit_axis, lo_, hi_)) {
      lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
      lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
      lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  __shared__ float sm[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { sm[threadIdx.x >> 6][a] = lo[a]; sm[threadIdx.x >> 6][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicMin(&bbox[threadIdx.x], enc_ordered(fminf(fminf(sm[0][threadIdx.x], sm[1][threadIdx.x]), fminf(sm[2][threadIdx.x], sm[3][threadIdx.x]))));
  else if (threadIdx.x < 6)
    atomicMax(&bbox[threadIdx.x], enc_ordered(fmaxf(fmaxf(sm[0][threadIdx.x], sm[1][threadIdx.x]), fmaxf(sm[2][threadIdx.x], sm[3][threadIdx.x]))));
}
void launch_voxel_bbox(const float4* xyzi, int n, int limit_axis, float lo, float hi, uint32_t* bbox, hipStream_t s) {
  hipLaunchKernelGGL(k_bbox_init_b, dim3(1), dim3(64), 0, s, bbox, 1);  // slots 0-2 = min, 3-5 = max (6, 7 unused)
  int blocks = (n + 255) / 256;
  if (blocks > 128) blocks = 128;
  hipLaunchKernelGGL(k_voxel_bbox, dim3(blocks), dim3(256), 0, s, xyzi, n, limit_axis, lo, hi, bbox);
}

__global__ void __launch_bounds__(256) k_voxel_keys(const float4* __restrict__ xyzi, int n, VoxelGridDesc g, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = xyzi[i];
  uint32_t key = 0xFFFFFFFFu;
  if (voxel_accept(p, g.limit_axis, g.lo, g.hi)) {
    int i0 = (int)(floorf(p.x * g.inv_leaf) - (float)g.minb[0]);
    int i1 = (int)(floorf(p.y * g.inv_leaf) - (float)g.minb[1]);
    int i2 = (int)(floorf(p.z * g.inv_leaf) - (float)g.minb[2]);
    key = (uint32_t)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}
void launch_voxel_keys(const float4* xyzi, int n, VoxelGridDesc g, uint32_t* keys, uint32_t* vals, hipStream_t s) {
  hipLaunchKernelGGL(k_voxel_keys, dim3((n + 255) / 256), dim3(256), 0, s, xyzi, n, g, keys, vals);
}

__global__ void __launch_bounds__(256) k_voxel_heads(const uint32_t* __restrict__ keys, int n, uint32_t* __restrict__ heads) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k = keys[i];
  heads[i] = (k != 0xFFFFFFFFu && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}
void launch_voxel_heads(const uint32_t* keys, int n, uint32_t* heads, hipStream_t s) {
  hipLaunchKernelGGL(k_voxel_heads, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, heads);
}

// nrm / out_nrm non-null: the pcl::VoxelGrid<PointXYZINormal> flavour (PointCloudFilter.cc:119-124) -- every field is
// averaged by pcl::CentroidPoint's accumulators: xyz, intensity and curvature are float sums / n, the normal is the float
// sum of the 4-vectors (normal_x, normal_y, normal_z, 0) NORMALISED (a zero sum stays zero)
// in_inten / out_inten non-null: the input is a device cloud (xyz with w = 1 + a separate intensity array) and the output goes straight into
// the new cloud's arrays -- no packing pass before the filter, no unpacking pass after it (two launches and 70 MB of traffic per 1 M-point frame)
__global__ void __launch_bounds__(256) k_voxel_centroids(const float4* __restrict__ xyzi, const float4* __restrict__ nrm, const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, const uint32_t* __restrict__ heads,
                                                         const uint32_t* __restrict__ rank, int n, float4* __restrict__ out, float4* __restrict__ out_nrm,
                                                         uint32_t cap, const float* __restrict__ in_inten, float* __restrict__ out_inten) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !heads[i]) return;
  uint32_t k = keys[i];
  float ax = 0.f, ay = 0.f, az = 0.f, ai = 0.f;
  float nx = 0.f, ny = 0.f, nz = 0.f, cu = 0.f;
  int cnt = 0;
  for (int j = i; j < n && keys[j] == k; j++) {  // stable radix sort => ascending point index inside a voxel
    uint32_t v = vals[j];
    float4 p = xyzi[v];
    ax += p.x; ay += p.y; az += p.z; ai += in_inten ? in_inten[v] : (out_inten ? 0.0f : p.w);   // (a cloud without intensities: zeros, like the packed copy had)
    if (nrm) {
      float4 q = nrm[v];
      nx += q.x; ny += q.y; nz += q.z; cu += q.w;
    }
    cnt++;
  }
  float c = (float)cnt;
  uint32_t r = rank[i] - 1u;
  if (r >= cap) return;
  if (out_inten) {
    out[r] = make_float4(ax / c, ay / c, az / c, 1.0f);
    out_inten[r] = ai / c;
  } else
    out[r] = make_float4(ax / c, ay / c, az / c, ai / c);
  if (nrm && out_nrm) {
    float z = (nx * nx + ny * ny) + nz * nz;
    if (z > 0.0f) { float l = sqrtf(z); nx = nx / l; ny = ny / l; nz = nz / l; }
    out_nrm[r] = make_float4(nx, ny, nz, cu / c);
  }
}
void launch_voxel_centroids(const float4* xyzi, const float4* nrm, const uint32_t* keys, const uint32_t* vals, const uint32_t* heads,
                            const uint32_t* rank_incl, int n, float4* out, float4* out_nrm, uint32_t out_cap, hipStream_t s, const float* in_inten,
                            float* out_inten) {
  hipLaunchKernelGGL(k_voxel_centroids, dim3((n + 255) / 256), dim3(256), 0, s, xyzi, nrm, keys, vals, heads, rank_incl, n, out, out_nrm, out_cap, in_inten,
                     out_inten);
}

}  // namespace lh
