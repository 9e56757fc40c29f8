// lh_index_small.hip -- K2 for SMALL clouds: the whole index build of one cloud in ONE launch, one workgroup per cloud.
//
// LOCUS runs its registration on ~3 000 points per scan (the adaptive voxel filter's target, lo_settings.yaml:84-85), one scan at a time.
// The general build (lh_kernels.hip: 17 launches -- bounding box, keys, a three-pass segmented radix sort (nine), leaf flags, leaves,
// the box tables, radix tree, 4-ary nodes) is laid out for batches of 100 k-point clouds; on one 3 k-point cloud every launch is a few
// microseconds of work behind ~9 us of dependent-launch latency: 122 us, half of a whole odometry update.  Here the same steps run as phases
// of one 1024-thread workgroup with the sort, the keys, the leaf numbering and the leaf records in LDS (112 KB) and the box / node tables in
// the build's usual global scratch (the workgroup's own L1 keeps them coherent between phases: workgroup-scope barriers only).
//
// THE TREE IS THE GENERAL PATH'S, BIT FOR BIT: every per-element decision is made by the same function (spatial_key30, leafcell_flag,
// radix_node, key_common, range boxes by exact min / max, quant_box, leaf_ref), the sort orders (key, original index) pairs -- what a
// stable sort by key leaves -- and node / leaf numbering is cloud-local in both.  (Clouds this small carry no start grid: GRID_MIN_POINTS.)
// tests/test_gpu_kernels.py compares the two builds' reachable nodes, headers and sorted arrays byte for byte.
#include "lh_kernels.hpp"

namespace lh {

constexpr int SMALL_THREADS = 1024;

struct SBox { float lx, ly, lz, hx, hy, hz; };
__device__ __forceinline__ SBox sbox_empty() { return SBox{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}; }
__device__ __forceinline__ void sbox_add(SBox& b, const float4& lo, const float4& hi) {
  b.lx = fminf(b.lx, lo.x); b.ly = fminf(b.ly, lo.y); b.lz = fminf(b.lz, lo.z);
  b.hx = fmaxf(b.hx, hi.x); b.hy = fmaxf(b.hy, hi.y); b.hz = fmaxf(b.hz, hi.z);
}
__device__ __forceinline__ void sbox_store(float4* tab, int idx, const SBox& b) {
  tab[2 * (size_t)idx] = make_float4(b.lx, b.ly, b.lz, 0.f);
  tab[2 * (size_t)idx + 1] = make_float4(b.hx, b.hy, b.hz, 0.f);
}
// box of the leaves [a, e): leaf boxes from the global table (eight independent loads per round), whole groups of 8 / 64 / 512 leaves from
// three small tables in LDS.  min / max are exact and order-free, so the box does not depend on how the range is cut up -- the general
// build cuts it along the batch's global leaf numbers in chunks of 32 and 1024 and gets the same bits.
struct SBoxC { float v[6]; };   // compact LDS entry
__device__ __forceinline__ void sbox_add_c(SBox& b, const SBoxC& c) {
  b.lx = fminf(b.lx, c.v[0]); b.ly = fminf(b.ly, c.v[1]); b.lz = fminf(b.lz, c.v[2]);
  b.hx = fmaxf(b.hx, c.v[3]); b.hy = fmaxf(b.hy, c.v[4]); b.hz = fmaxf(b.hz, c.v[5]);
}
__device__ __forceinline__ void leaf_run(SBox& b, const float4* __restrict__ lbox, int first, int last) {
  for (int l = first; l < last; l += 8) {
    float4 lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int idx = l + k < last ? l + k : last - 1;   // (a short round repeats its last entry: min / max do not mind)
      lo[k] = lbox[2 * (size_t)idx]; hi[k] = lbox[2 * (size_t)idx + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sbox_add(b, lo[k], hi[k]);
  }
}
__device__ __forceinline__ SBox small_range_box(const float4* __restrict__ lbox, const SBoxC* b8, const SBoxC* b64, const SBoxC* b512, int a, int e) {
  SBox b = sbox_empty();
  if (e - a <= 16) { leaf_run(b, lbox, a, e); return b; }
  const int a8 = (a + 7) & ~7, e8 = e & ~7;
  leaf_run(b, lbox, a, a8);
  leaf_run(b, lbox, e8, e);
  int c0 = a8 >> 3, c1 = e8 >> 3;
  if (c1 - c0 <= 16) { for (int c = c0; c < c1; c++) sbox_add_c(b, b8[c]); return b; }
  int c0a = (c0 + 7) & ~7, c1a = c1 & ~7;
  for (int c = c0; c < c0a; c++) sbox_add_c(b, b8[c]);
  for (int c = c1a; c < c1; c++) sbox_add_c(b, b8[c]);
  c0 = c0a >> 3; c1 = c1a >> 3;
  if (c1 - c0 <= 16) { for (int c = c0; c < c1; c++) sbox_add_c(b, b64[c]); return b; }
  c0a = (c0 + 7) & ~7; c1a = c1 & ~7;
  for (int c = c0; c < c0a; c++) sbox_add_c(b, b64[c]);
  for (int c = c1a; c < c1; c++) sbox_add_c(b, b64[c]);
  for (int c = c0a >> 3; c < (c1a >> 3); c++) sbox_add_c(b, b512[c]);
  return b;
}

__global__ void __launch_bounds__(SMALL_THREADS) k_index_small(const IndexDesc* __restrict__ descs, TreeScratch t) {
  const IndexDesc d = descs[blockIdx.x];
  const int n = d.n, tid = threadIdx.x;
  if (n <= 0) return;
  __shared__ uint64_t skey[SMALL_INDEX_MAX_N];        // sort keys (key30 << 32 | index), then (cloud << 32 | key30) of the sorted positions; LATER the radix nodes' children
  __shared__ uint32_t vals[SMALL_INDEX_MAX_N];        // original index of the point at a sorted position; LATER the radix nodes' parents
  __shared__ alignas(16) uint32_t lid[SMALL_INDEX_MAX_N];   // inclusive scan of the leaf-start flags (the radix sort reads it as uint4: 16-byte aligned)
  __shared__ uint64_t lkey[SMALL_INDEX_MAX_N + 1];    // key of a leaf's first point (+ sentinel)
  __shared__ uint32_t lstart[SMALL_INDEX_MAX_N + 1];  // first sorted position of a leaf (+ sentinel)
  __shared__ SBoxC b8[SMALL_INDEX_MAX_N / 8], b64[SMALL_INDEX_MAX_N / 64], b512[SMALL_INDEX_MAX_N / 512];   // boxes of 8 / 64 / 512 consecutive leaves
  __shared__ float red[SMALL_THREADS / 64][6];
  __shared__ float bb[6];
  __shared__ uint32_t wsum[SMALL_THREADS / 64];
  __shared__ uint32_t n_leaves_s;
  // the cloud's slices of the build scratch (position- and leaf-indexed tables: a cloud has at most as many leaves as points)
  float4* const lbox = t.lbox + 2 * (size_t)d.offset;
  float4* const ibox = t.ibox + 2 * (size_t)d.offset;
  // the binary radix tree lives in LDS, in the space the sort keys and the permutation no longer need once the leaves exist (the 4-ary
  // nodes' depth test climbs parent links: a dozen dependent reads per node, LDS latency instead of L2 latency)
  int32_t* const ichild = reinterpret_cast<int32_t*>(skey);
  int32_t* const iparent = reinterpret_cast<int32_t*>(vals);

  // ---- A: bounding box (k_bbox_b) and the quantisation frame (k_key_b, thread 0) ----
  {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += SMALL_THREADS) {
      const float4 p = d.xyz[i];
      lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
      lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
      lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int a = 0; a < 3; a++) {
        lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
        hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
      }
    if ((tid & 63) == 0)
#pragma unroll
      for (int a = 0; a < 3; a++) { red[tid >> 6][a] = lo[a]; red[tid >> 6][3 + a] = hi[a]; }
    __syncthreads();
    if (tid < 6) {
      float v = red[0][tid];
      for (int w = 1; w < SMALL_THREADS / 64; w++) v = tid < 3 ? fminf(v, red[w][tid]) : fmaxf(v, red[w][tid]);
      bb[tid] = v;
    }
    __syncthreads();
  }
  TreeHeader fr;
  {
    const float lo[3] = {bb[0], bb[1], bb[2]}, hi[3] = {bb[3], bb[4], bb[5]};
    quant_frame(lo, hi, &fr);   // every thread holds the frame; thread 0 publishes it
    fr.grid_on = 0;             // (n < GRID_MIN_POINTS)
    fr.root = 0; fr.n_leaves = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) fr.pad[k] = 0;
  }
  // ---- B: keys (k_key_b) ----
  for (int i = tid; i < n; i += SMALL_THREADS) {
    const float4 p = d.xyz[i];
    skey[i] = ((uint64_t)spatial_key30(p.x, p.y, p.z, bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]) << 32) | (uint32_t)i;
  }
  __syncthreads();
  // ---- C: sort by the 30-bit key, stable (equal keys keep ascending original index: the order a sort of the (key, index) pairs leaves).
  // Round 5: a least-significant-digit radix sort in LDS, four passes of 8 bits, instead of the bitonic network (78 stages for 4 096 keys:
  // 31 of the build's 63 us on a 3 k-point cloud, tools/probe_small_index.py).  Wave w owns positions [256 w, 256 w + 256) in four batches of 64
  // consecutive elements; per batch eight ballots give every lane the mask of the lanes that hold its digit (rank = the set bits below it), a
  // per-(digit, wave) counter in LDS takes the batch's group sizes in batch order, one workgroup-wide exclusive scan in (digit, wave) order turns
  // the counters into first positions, and every element goes to first position + (its group's offset inside the wave) + rank.  The same scheme
  // as k_rs_scatter (lh_radix.hip) with the whole array in one workgroup.  Buffers: skey <-> lkey (free until phase G), counters in lid.
  {
    uint64_t* src = skey;
    uint64_t* dst = lkey;
    uint32_t* const cnt = lid;                 // [256 digits][16 waves]
    const int lane = tid & 63, wave = tid >> 6;
    for (int shift = 0; shift < 32; shift += 8) {
      uint4* const cnt4 = reinterpret_cast<uint4*>(cnt);
      cnt4[tid] = make_uint4(0u, 0u, 0u, 0u);  // 1 024 threads x 4 counters
      __syncthreads();
      uint64_t e[4];
      uint32_t where[4];                       // digit << 16 | (offset of the element among its wave's elements of that digit)
#pragma unroll
      for (int bt = 0; bt < 4; bt++) {
        const int i = wave * 256 + bt * 64 + lane;
        const bool live = i < n;
        e[bt] = live ? src[i] : ~0ull;
        const uint32_t dg = (uint32_t)(e[bt] >> (32 + shift)) & 255u;
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const bool bit = (dg >> k) & 1u;
          const unsigned long long m = __ballot(bit);
          same &= bit ? m : ~m;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
        uint32_t base = 0;
        if (live) {
          base = cnt[dg * 16 + wave];                                   // what the wave's earlier batches put there (a wave's LDS operations complete in order)
          if (rank == 0) cnt[dg * 16 + wave] = base + (uint32_t)__popcll(same);
        }
        where[bt] = (dg << 16) | (base + rank);
      }
      __syncthreads();
      {  // exclusive scan of the 4 096 counters in (digit, wave) order: four consecutive ones per thread
        const uint4 v = cnt4[tid];
        const uint32_t mine = v.x + v.y + v.z + v.w;
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
        cnt4[tid] = make_uint4(b0, b0 + v.x, b0 + v.x + v.y, b0 + v.x + v.y + v.z);
        __syncthreads();
      }
#pragma unroll
      for (int bt = 0; bt < 4; bt++) {
        const int i = wave * 256 + bt * 64 + lane;
        if (i < n) dst[cnt[(where[bt] >> 16) * 16 + wave] + (where[bt] & 0xffffu)] = e[bt];
      }
      __syncthreads();
      uint64_t* const sw = src; src = dst; dst = sw;
    }
    // (four passes: the sorted pairs are back in skey)
  }
  // ---- D: sorted keys in the build's form (cloud id above the 30-bit key) + the permutation ----
  for (int g = tid; g < n; g += SMALL_THREADS) {
    const uint64_t s = skey[g];
    vals[g] = (uint32_t)s;
    skey[g] = ((uint64_t)blockIdx.x << 32) | (s >> 32);
  }
  __syncthreads();
  // ---- E + F: leaf-start flags (k_leafcell_b) and their inclusive scan (k_leaves_b's lid), four consecutive positions per thread ----
  {
    uint32_t f[4], s = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int g = 4 * tid + e;
      f[e] = g < n ? leafcell_flag(skey, n, g) : 0u;
      s += f[e];
    }
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - s;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (tid == SMALL_THREADS - 1) n_leaves_s = base + s;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int g = 4 * tid + e;
      base += f[e];
      if (g < n) lid[g] = base;
    }
    __syncthreads();
  }
  const int nl = (int)n_leaves_s;
  // ---- G: the sorted points (x, y, z, original index), their padding, and the leaf records (k_leaves_b) ----
  for (int g = tid; g < n; g += SMALL_THREADS) {
    const uint32_t j = vals[g];
    const float4 p = d.xyz[j];
    d.sorted[g] = make_float4(p.x, p.y, p.z, __uint_as_float(j));
    if (g == n - 1) {
#pragma unroll
      for (int e = 1; e <= LEAF_CAP; e++) d.sorted[g + e] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0x7fffffffu));
    }
    const uint32_t L = lid[g];
    if (L != (g ? lid[g - 1] : 0u)) { lkey[L - 1u] = skey[g]; lstart[L - 1u] = (uint32_t)g; }
    if (g == n - 1) { lstart[L] = (uint32_t)n; lkey[L] = ~0ull; }
  }
  __syncthreads();
  if (nl == 1) {   // the whole cloud is one leaf (k_leafbox_b's special case): no internal node
    if (tid == 0) {
      fr.root = leaf_ref(0u, n);
      fr.n_leaves = 1;
      *d.hdr = fr;
    }
    return;
  }
  // ---- H: leaf boxes (k_leafbox_b): a leaf's <= 8 points in one round of loads (the sorted array is padded, the count masks) ----
  for (int L = tid; L < nl; L += SMALL_THREADS) {
    const uint32_t s0 = lstart[L], s1 = lstart[L + 1];
    float4 q[LEAF_CAP];
#pragma unroll
    for (int e = 0; e < LEAF_CAP; e++) q[e] = d.sorted[s0 + e];
    SBox b = sbox_empty();
#pragma unroll
    for (int e = 0; e < LEAF_CAP; e++)
      if (s0 + (uint32_t)e < s1) sbox_add(b, q[e], q[e]);
    sbox_store(lbox, L, b);
  }
  __syncthreads();
  // ---- I: group tables (the general build's k_chunkbox_b), in LDS ----
  const int n8 = (nl + 7) >> 3, n64 = (n8 + 7) >> 3, n512 = (n64 + 7) >> 3;
  for (int c = tid; c < n8; c += SMALL_THREADS) {
    SBox b = sbox_empty();
    leaf_run(b, lbox, 8 * c, min(nl, 8 * c + 8));
    b8[c] = SBoxC{{b.lx, b.ly, b.lz, b.hx, b.hy, b.hz}};
  }
  __syncthreads();
  for (int c = tid; c < n64; c += SMALL_THREADS) {
    SBox b = sbox_empty();
    for (int l = 8 * c; l < min(n8, 8 * c + 8); l++) sbox_add_c(b, b8[l]);
    b64[c] = SBoxC{{b.lx, b.ly, b.lz, b.hx, b.hy, b.hz}};
  }
  __syncthreads();
  for (int c = tid; c < n512; c += SMALL_THREADS) {
    SBox b = sbox_empty();
    for (int l = 8 * c; l < min(n64, 8 * c + 8); l++) sbox_add_c(b, b64[l]);
    b512[c] = SBoxC{{b.lx, b.ly, b.lz, b.hx, b.hy, b.hz}};
  }
  __syncthreads();
  // ---- J: Karras' radix tree over the leaf keys + one range box per binary node (k_radix_b) ----
  for (int i = tid; i < nl - 1; i += SMALL_THREADS) {
    int left, right, lo, hi, delta;
    radix_node(lkey, nl, i, left, right, lo, hi, &delta);
    ichild[2 * i] = left; ichild[2 * i + 1] = right;   // (no start grid here: the common-prefix lengths and the leaf ranges are not kept)
    if (left >= 0) iparent[left] = i;
    if (right >= 0) iparent[right] = i;
    sbox_store(ibox, i, small_range_box(lbox, b8, b64, b512, lo, hi + 1));
  }
  __syncthreads();
  // ---- K: the 4-ary nodes the walk can reach (k_nodex_b without a start grid: even depth below the root) ----
  for (int i = tid; i < nl - 1; i += SMALL_THREADS) {
    int depth = 0, j = i;
    while (j != 0) { j = iparent[j]; depth++; }   // (Karras: node 0 covers every leaf -- the root)
    if (depth & 1) continue;
    const int lc = ichild[2 * i], rc = ichild[2 * i + 1];
    int cref[4] = {0, 0, 0, 0}, cnt = 0;
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int c = side ? rc : lc;
      if (c < 0) cref[cnt++] = c;
      else { cref[cnt++] = ichild[2 * c]; cref[cnt++] = ichild[2 * c + 1]; }
    }
    NodeX nd;
#pragma unroll
    for (int k = 0; k < 4; k++) { nd.lo_xy[k] = 0xffffffffu; nd.hi_xy[k] = 0u; nd.z_lohi[k] = 0xffffffffu; nd.child[k] = NO_CHILD; }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < cnt) {
        const int ref = cref[k];
        const float4* tab = ref < 0 ? lbox : ibox;
        const int idx = ref < 0 ? ~ref : ref;
        const float4 blo = tab[2 * (size_t)idx], bhi = tab[2 * (size_t)idx + 1];
        quant_box(fr, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k]);
        nd.child[k] = ref < 0 ? leaf_ref(lstart[~ref], (int)(lstart[~ref + 1] - lstart[~ref])) : ref;
      }
    d.nodes[i] = nd;
    if (i == 0) {   // the root publishes the header
      fr.root = i;
      fr.n_leaves = nl;
      *d.hdr = fr;
    }
  }
}

void launch_index_small(const IndexDesc* descs_dev, int n_clouds, const TreeScratch& t, hipStream_t s) {
  hipLaunchKernelGGL(k_index_small, dim3(n_clouds), dim3(SMALL_THREADS), 0, s, descs_dev, t);
}

}  // namespace lh
