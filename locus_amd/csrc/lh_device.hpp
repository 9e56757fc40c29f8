// lh_device.hpp -- data layout + per-thread device functions of the GICP hot path (gfx950).
//
// The functions here are __host__ __device__ so that tests/host_emu can exercise the traversal logic on
// the CPU against brute force; the product only ever calls them from HIP kernels (lh_kernels.hip).
//
// NN index ("K2", replaces the FLANN kd-tree built by tree_->setInputCloud in pcl::Registration::initCompute):
//   * target points are sorted by a 30-bit Morton (Z-order) index (10 bits per axis of a cubic grid over the cloud's box): a key
//     PREFIX is then an axis-aligned box whose extent can be read off the prefix length -- what the start grid of the GICP
//     sweeps needs (grid_start below).  (Round 1-2 sorted along a Hilbert curve, which mattered for the first layout's
//     equal-count runs; the cell-aligned leaves below are the same sets of grid cells under either curve.)
//   * LEAVES are cells of that grid hierarchy: the largest key-prefix cell around a point that holds <= LEAF_CAP points
//     (runs of > LEAF_CAP identical keys are cut into chunks).  A key prefix is an aligned box, so leaves are DISJOINT in
//     space; equal-count runs of the curve (the previous layout) overlap their neighbours at every level and cost 2-3x
//     the node visits per query (measured: 17.8 -> 10.2 dependent steps warm, 31.8 -> 10.4 cold)
//   * a binary radix tree over the leaves' keys (Karras 2012: one thread per internal node, no scans) gives the
//     hierarchy; every internal node adopts its GRANDCHILDREN, so the walk sees 4-ary nodes (NodeX, 64 B: the 16-bit
//     fixed-point AABBs of <= 4 children + their references) and half the dependent steps of the binary tree
//   * sorted point = float4(x, y, z, bitcast(original index)) so one 16-B load yields position + id
// Exactness: point distances are float ((dx*dx+dy*dy)+dz*dz), no FMA contraction (the TU is compiled with
// -ffp-contract=off).  The box distance is a LOWER bound of that float value for every point inside (boxes are rounded
// outwards by a whole grid step, which dominates every rounding error of the bound -- see quant_lo / boxd2_q), so pruning
// on box_d2 > best is exact, and ties are resolved to the lowest original index -- the same rule as the CPU oracle, so
// indices match bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LH_HD __host__ __device__ __forceinline__

// 16-B load from GLOBAL memory: the tree and point pointers reach the walk through descriptor structs, so the compiler
// only knows them as generic pointers and would emit flat_load (which also takes a slot in the LDS queue and makes every
// wait on the loads a wait on the traversal-stack LDS traffic too)
template <class T>
__host__ __device__ __forceinline__ T gld(const T* p) {   // typed load / store through a GLOBAL pointer (see gload16)
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(p));
#else
  return *p;
#endif
}
template <class T>
__host__ __device__ __forceinline__ void gst(T* p, const T& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  *reinterpret_cast<T __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(p)) = v;
#else
  *p = v;
#endif
}
template <class V>
__host__ __device__ __forceinline__ V gload16(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const V __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(p));
#else
  return *reinterpret_cast<const V*>(p);
#endif
}

#ifndef LH_GRID_FINEST
#define LH_GRID_FINEST 5
#endif
#define LH_GRID_FINEST_OR_DEFAULT LH_GRID_FINEST

namespace lh {

constexpr int LEAF_CAP = 8;     // max points per leaf (128 B = one cache line of sorted points)
#ifndef LH_LDS_STACK
#define LH_LDS_STACK 12
#endif
constexpr int LDS_STACK = LH_LDS_STACK;   // traversal-stack entries (64-bit) a thread keeps in LDS; deeper entries spill to private memory (-DLH_LDS_STACK: A/B builds)
constexpr int MAX_POINT_BITS = 27; // a cloud holds at most 2^27 points (leaf references keep 27 bits of sorted position; build_indices checks)
constexpr int GRID_CELL_ROOTS = LH_GRID_FINEST_OR_DEFAULT - 2; // levels of the start grid (a cell root is kept as a child instead of being adopted: one 4-ary level for ONE binary level)
// The traversal stack is unchecked.  Its structural bound: the binary radix tree is at most 30 key bits + MAX_POINT_BITS tie-break bits deep
// (runs of identical keys are split by leaf index); a 4-ary node adopts its grandchildren, so a 4-ary level covers two binary levels except
// at the (at most three) cell roots on a path => ceil((57 + 3) / 2) = 30 levels; a visit stacks at most 3 siblings and descends into the
// fourth child => 3 * 30 + 1 = 91 entries for a walk from the root.  A walk that starts in the grid stacks <= 7 neighbour cells and then
// begins at a level-5 cell root, >= 7 levels down: 7 + 3 * 23 + 1 = 77.  LDS_STACK + SPILL_MAX = 92.
constexpr int STACK_BOUND = 3 * ((30 + MAX_POINT_BITS + GRID_CELL_ROOTS + 1) / 2) + 1;   // 91 with three table levels, 94 with four or five
constexpr int SPILL_MAX = (STACK_BOUND > 92 ? STACK_BOUND : 92) - LDS_STACK;   // (80 with the default LDS_STACK and three table levels)
constexpr int MAX_DEPTH = 12;   // (size of the instrumentation histogram; only slot 0 is used by the explicit tree)

// child reference: >= 0 internal node index (cloud-local); < 0 leaf: ~ref = (first sorted position << 4) | (count - 1)
// Child boxes are 16-bit fixed point on the cloud's own ISOTROPIC grid (TreeHeader: origin + one step), rounded OUTWARDS by
// one step, packed so that the box distance is integer SIMD-within-a-register work: per child two saturating packed
// subtractions for x|y, one for z (low half: lo - q, high half: q - hi on complemented values), and two
// v_dot2_u32_u16 for the sum of squares -- 8 VALU instructions per child instead of 12 float ones plus conversions, and a
// node is 64 B = four 16-B loads per visit instead of seven.  (The walking sweeps are bound by VALU issue at ~50 % lane
// utilisation, with the texture-address path 60 % busy next to it.)  An outward-rounded box only ever lowers a lower
// bound, so pruning stays exact (quant_lo / quant_hi / boxd2_q below).
constexpr int32_t NO_CHILD = 0x7fffffff;
struct alignas(16) NodeX {
  uint32_t lo_xy[4];   // lox | loy << 16                       (grid units, per child)
  uint32_t hi_xy[4];   // hix | hiy << 16
  uint32_t z_lohi[4];  // loz | (65535 - hiz) << 16
  int32_t child[4];    // absent child: NO_CHILD (its boxes are unused)
};
static_assert(sizeof(NodeX) == 64, "NodeX is half a 128-B line");
static_assert(3 * ((30 + MAX_POINT_BITS + GRID_CELL_ROOTS + 1) / 2) + 1 <= LDS_STACK + SPILL_MAX,
              "the unchecked traversal stack must hold the deepest walk of the largest cloud");
constexpr float QUANT_STEPS = 65532.0f;  // grid steps across the cloud's largest extent: 0 .. 65532, +-1 of outward rounding <= 65535
struct TreeHeader {
  int32_t root;       // child reference of the root (a leaf reference for clouds of <= LEAF_CAP points)
  int32_t n_leaves;
  float org[3];       // grid origin (the cloud's bounding-box minimum)
  float inv;          // 1 / step
  float scl;          // step = largest extent / QUANT_STEPS (>= 1e-30)
  float scl2;         // step^2
  int32_t grid_on;    // the start grid behind this header is filled (clouds of >= GRID_MIN_POINTS points)
  float key_sc;       // cells of the 10-bit KEY grid per metre (spatial_key30's scale: same float, so a query is binned like the points were)
  float key_inv;      // 1 / key_sc
  int32_t pad[5];
};
static_assert(sizeof(TreeHeader) == 64, "TreeHeader occupies one NodeX slot in front of the grid and the nodes");
// The START GRID of the GICP sweeps (grid_start): three direct tables -- the 32^3, 16^3 and 8^3 coarsenings of the key grid -- whose
// entry for a cell is the child reference of the radix-tree node (or leaf) that holds EXACTLY the cloud's points inside that cell:
// keys that share a prefix are one subtree of a binary radix tree, and a Morton prefix of 3 l bits is the level-l cell.  A warm 1-NN
// query starts its walk there instead of at the root (the five or six levels above are the same for every query of a cell) and
// owes the rest of the cloud only the neighbour cells its candidate's distance ball reaches: their subtrees go on the traversal
// stack with the cell's box distance as key, and everything else is bounded by the distance to the faces it does not cross.
// Layout: [TreeHeader][GRID_ENTRIES x int32][nodes ...] in one buffer, so the tables need no pointer of their own.
#ifndef LH_GRID_FINEST
#define LH_GRID_FINEST 5
#endif
constexpr int GRID_FINEST = LH_GRID_FINEST;          // finest table level (5: 32^3; the host model of tools/model tries 6 and 7)
constexpr int GRID_LEVELS = GRID_FINEST - 2;         // levels GRID_FINEST .. 3: cells of 32, 64, 128 key cells (2.5 / 5 / 10 m on an 80-m scene) at 5, 4, 3
LH_HD constexpr int grid_off(int level) {            // tables in order of decreasing level: entries of the finer tables come first
  int off = 0;
  for (int l = GRID_FINEST; l > level; l--) off += 1 << (3 * l);
  return off;
}
constexpr int GRID_OFF5 = grid_off(5), GRID_OFF4 = grid_off(4), GRID_OFF3 = grid_off(3);
constexpr int GRID_ENTRIES = grid_off(3) + 512;      // 37 376 entries = 146 KB with three levels
constexpr int GRID_NODEX = GRID_ENTRIES * 4 / 64;    // the same in NodeX slots (2 336)
static_assert(GRID_NODEX * 64 == GRID_ENTRIES * 4, "the grid fills whole node slots");
static_assert(GRID_FINEST != 5 || (GRID_OFF5 == 0 && GRID_OFF4 == 32768 && GRID_OFF3 == 32768 + 4096 && GRID_ENTRIES == 37376), "three-level layout");
constexpr int32_t GRID_EMPTY = (int32_t)0x80000000;  // no point of the cloud lies in the cell (never a valid reference)
constexpr int GRID_MIN_POINTS = 8192;                // smaller clouds: the tree is shallow, the tables would cost more than they save
constexpr float GRID_SLACK = 4e-3f;                  // key cells (0.3 mm on an 80-m scene): ten times the float rounding of a key coordinate
struct TreeView {
  const float4* pts;        // sorted points (+ LEAF_CAP padding entries of +inf / id INT_MAX)
  const NodeX* nodes;       // internal nodes, cloud-local indices
  const TreeHeader* hdr;    // written by the build kernels (device memory); the start grid lies right behind it
  int n_points;
  LH_HD const int32_t* grid() const { return reinterpret_cast<const int32_t*>(hdr + 1); }
};
LH_HD int32_t leaf_ref(uint32_t first_pos, int count) { return ~(int32_t)((first_pos << 4) | (uint32_t)(count - 1)); }

LH_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  union { float f; uint32_t u; } v; v.f = f; return v.u;
#endif
}
LH_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  union { float f; uint32_t u; } v; v.u = u; return v.f;
#endif
}
LH_HD float inf_f() { return u2f(0x7f800000u); }
// square root for BOUNDS that carry a relative margin of >= 1e-6 anyway (certificate tests, the start grid's ball radius): the raw
// v_sqrt_f32 (1 ulp) on the device -- sqrtf is the correctly rounded one, fifteen instructions of refinement each
LH_HD float sqrt_bound(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}

LH_HD float d2f(float qx, float qy, float qz, float px, float py, float pz) {
  float dx = qx - px, dy = qy - py, dz = qz - pz;
  return (dx * dx + dy * dy) + dz * dz;
}
LH_HD float boxd2(float qx, float qy, float qz, float lx, float ly, float lz, float hx, float hy, float hz) {
  float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.0f);
  float dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.0f);
  float dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.0f);
  return (dx * dx + dy * dy) + dz * dz;
}

// cells of the 10-bit key grid per metre: ONE isotropic cell size (cubes), the largest extent spans the 1024 cells
LH_HD float key_scale(float lx, float ly, float lz, float hx, float hy, float hz) {
  float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
  return 1023.999f / ext;
}
// The cloud's quantisation frame from its bounding box (written once per index build; shared with the host-side check).
LH_HD bool quant_frame(const float lo[3], const float hi[3], TreeHeader* h) {   // false: empty or non-finite bounding box
  float ext = 0.0f;
  bool ok = true;
  for (int a = 0; a < 3; a++) {
    ok = ok && hi[a] >= lo[a] && (hi[a] - lo[a]) < inf_f();   // false for an empty / non-finite box
    ext = fmaxf(ext, hi[a] - lo[a]);
  }
  if (!ok) ext = 0.0f;
  h->scl = fmaxf(ext / QUANT_STEPS, 1e-30f);
  h->inv = 1.0f / h->scl;
  h->scl2 = h->scl * h->scl;
  for (int a = 0; a < 3; a++) h->org[a] = ok ? lo[a] : 0.0f;
  h->key_sc = ok ? key_scale(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]) : 1.0f;
  h->key_inv = 1.0f / h->key_sc;
  h->grid_on = 0;   // the build switches it on once the tables are filled (k_key_b)
  return ok;
}
// Outward rounding: org + quant_lo(v) * scl <= v <= org + quant_hi(v) * scl for every v inside the cloud's box, with
// a whole grid step of slack on each side; the float error of (v - org) * inv (< 0.02 step at 65532 steps), the integer
// rounding of the query (towards the box, see grid_query) and the float rounding of the final scaling (relative 3e-7 of a
// distance of at most 1.2e5 steps) together can never use it up.
LH_HD uint32_t quant_lo(float v, float org, float inv) {
  float g = floorf((v - org) * inv) - 1.0f;
  g = g > 0.0f ? g : 0.0f;            // also catches NaN
  return (uint32_t)(g < 65535.0f ? g : 65535.0f);
}
LH_HD uint32_t quant_hi(float v, float org, float inv) {
  float g = ceilf((v - org) * inv) + 1.0f;
  g = g < 65535.0f ? g : 65535.0f;
  return (uint32_t)(g > 0.0f ? g : 0.0f);
}
LH_HD void quant_box(const TreeHeader& h, float lx, float ly, float lz, float hx, float hy, float hz, uint32_t& lo_xy,
                     uint32_t& hi_xy, uint32_t& z_lohi) {
  lo_xy = quant_lo(lx, h.org[0], h.inv) | (quant_lo(ly, h.org[1], h.inv) << 16);
  hi_xy = quant_hi(hx, h.org[0], h.inv) | (quant_hi(hy, h.org[1], h.inv) << 16);
  z_lohi = quant_lo(lz, h.org[2], h.inv) | ((65535u - quant_hi(hz, h.org[2], h.inv)) << 16);
}
// The query on the grid: clamped into it, rounded UP where it is subtracted from a box minimum and DOWN where a box maximum
// is subtracted from it.  A query outside the grid by e_a steps along axis a is (e_a + c_a) steps from a box that is c_a
// steps from the clamped position, and sum (e_a + c_a)^2 >= sum c_a^2 + sum e_a^2: the second sum is a per-query constant
// (e2, world units, rounded down) added to every box distance, so scan points outside the target's bounding box (the
// sensor moved) prune as well as the ones inside.
struct GridQuery { uint32_t up_xy, dn_xy, z; float e2; };
LH_HD GridQuery grid_query(const TreeHeader& h, float qx, float qy, float qz) {
  float g[3] = {(qx - h.org[0]) * h.inv, (qy - h.org[1]) * h.inv, (qz - h.org[2]) * h.inv};
  uint32_t up[3], dn[3];
  float e2 = 0.0f;
  for (int a = 0; a < 3; a++) {
    float c = fminf(fmaxf(g[a], 0.0f), 65535.0f);   // NaN -> 0
    float e = fmaxf(fmaxf(-g[a], g[a] - 65535.0f), 0.0f);
    e = fminf(e, 1e15f);                            // keeps e * e finite; NaN -> 0
    e2 = e2 + e * e;
    up[a] = (uint32_t)ceilf(c);
    dn[a] = (uint32_t)floorf(c);
  }
  return GridQuery{up[0] | (up[1] << 16), dn[0] | (dn[1] << 16), up[2] | ((65535u - dn[2]) << 16), (e2 * h.scl2) * 0.999998f};
}
LH_HD uint32_t pk_subsat_u16(uint32_t a, uint32_t b) {   // per 16-bit half: max(a - b, 0)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b)));
#else
  uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return (al > bl ? al - bl : 0u) | ((ah > bh ? ah - bh : 0u) << 16);
#endif
}
LH_HD uint32_t udot2_sat(uint32_t a, uint32_t c) {       // lo(a)^2 + hi(a)^2 + c, saturating at 2^32 - 1
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, a), c, true);
#else
  uint64_t l = a & 0xffffu, h = a >> 16, r = l * l + h * h + c;
  return r > 0xffffffffull ? 0xffffffffu : (uint32_t)r;
#endif
}
// squared distance (a lower bound, world units) from a grid query to a quantised child box
LH_HD float boxd2_q(const GridQuery& q, uint32_t lo_xy, uint32_t hi_xy, uint32_t z_lohi, float scl2) {
  uint32_t dxy = pk_subsat_u16(lo_xy, q.up_xy) | pk_subsat_u16(q.dn_xy, hi_xy);   // per axis at most one side is non-zero
  uint32_t dz = pk_subsat_u16(z_lohi, q.z);                                        // (lo - q | q - hi), at most one non-zero
  return fmaf((float)udot2_sat(dxy, udot2_sat(dz, 0u)), scl2, q.e2);
}

// y = T * (x,y,z,1) in float: ((m0*x + m1*y) + m2*z) + m3 per row; T = 12 floats row-major 3x4.
// Same operation order as the oracle's xform_pt (stands in for Eigen's Matrix4f * Vector4f at gicp.hpp:469,382).
LH_HD void xform_pt(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  oy = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  oz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}
LH_HD void xform_nrm(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = (T[0] * x + T[1] * y) + T[2] * z;
  oy = (T[4] * x + T[5] * y) + T[6] * z;
  oz = (T[8] * x + T[9] * y) + T[10] * z;
}

LH_HD uint32_t expand10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
LH_HD uint32_t morton30(uint32_t ix, uint32_t iy, uint32_t iz) { return (expand10(ix) << 2) | (expand10(iy) << 1) | expand10(iz); }

// sort key of a point given the cloud's bounding box (shared by the build kernel and the host-side traversal check)
LH_HD uint32_t spatial_key30(float px, float py, float pz, float lx, float ly, float lz, float hx, float hy, float hz) {
  float sc = key_scale(lx, ly, lz, hx, hy, hz);
  int ix = (int)((px - lx) * sc), iy = (int)((py - ly) * sc), iz = (int)((pz - lz) * sc);
  ix = ix < 0 ? 0 : (ix > 1023 ? 1023 : ix);
  iy = iy < 0 ? 0 : (iy > 1023 ? 1023 : iy);
  iz = iz < 0 ? 0 : (iz > 1023 ? 1023 : iz);
  return morton30((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}
LH_HD uint32_t compact10(uint32_t v) {  // every third bit -> 10 bits (inverse of expand10)
  v &= 0x09249249u;
  v = (v | (v >> 2)) & 0x030C30C3u;
  v = (v | (v >> 4)) & 0x0300F00Fu;
  v = (v | (v >> 8)) & 0x030000FFu;
  v = (v | (v >> 16)) & 0x000003FFu;
  return v;
}
// The cell of a key prefix: `common` = number of leading bits of the 30-bit key shared by every key of the subtree (0..30).
// Morton order interleaves x, y, z from the top, so the prefix fixes ceil(common / 3), ceil((common - 1) / 3), ceil((common - 2) / 3)
// leading bits of the three cell coordinates: an aligned box of 10-bit cells [c0[a], c1[a]] per axis.
LH_HD void morton_cell(uint32_t key30, int common, uint32_t c0[3], uint32_t c1[3]) {
  const uint32_t cell[3] = {compact10(key30 >> 2), compact10(key30 >> 1), compact10(key30)};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int nb = (common - a + 2) / 3;           // bits of this axis inside the prefix
    nb = nb < 0 ? 0 : (nb > 10 ? 10 : nb);
    const uint32_t free_bits = 10u - (uint32_t)nb;
    c0[a] = (cell[a] >> free_bits) << free_bits;
    c1[a] = c0[a] + ((1u << free_bits) - 1u);
  }
}
LH_HD void cswap(uint64_t& a, uint64_t& b) {
  uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo; b = hi;
}

// Collector concept: float bound() const; void offer(float d2, int id); void skip(float box_d2);
// (a collector with kXyz = true is offered the point itself instead of its id: offer_xyz(d2, x, y, z))
// A subtree / leaf is visited iff box_d2 <= bound() (ties must be visited for the lowest-index rule); skip() is told the
// box distance of every subtree that is pruned (so a collector can keep a lower bound on everything it never looked at).
// kGreedy collectors (1-NN: re-offering a point is harmless) that start without a bound first walk straight down to the
// nearest leaf, so that the real traversal starts with a finite bound and stacks only what can still matter.
// Traversal stack: 64-bit entries (float bits of the box distance | child reference); the first LDS_STACK entries of a
// thread live in LDS (layout [entry][thread], conflict-free; a plain array on the host), deeper ones in private memory --
// a warm-started walk (GICP sweeps) never gets that deep (measured max 5..17), a cold k-NN walk occasionally does.
LH_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return x ? __clzll((long long)x) : 64;
#else
  return x ? __builtin_clzll(x) : 64;
#endif
}
LH_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return x ? __clz((int)x) : 32;
#else
  return x ? __builtin_clz(x) : 32;
#endif
}

template <class Collector>
LH_HD void scan_leaf(const TreeView& t, int32_t ref, float qx, float qy, float qz, Collector& col) {
  col.count_leaf();
  const uint32_t u = (uint32_t)~ref;
  const int cnt = (int)(u & 15u) + 1;
  const float4* p = t.pts + (u >> 4);
  if constexpr (Collector::kGreedy && !Collector::kXyz) {
    // 1-NN collectors (re-offering a point is harmless): ALL LEAF_CAP entries, unconditionally.  The ones past the leaf's own are the
    // next leaf's points (or the +inf padding): real points at their real distances, so neither the neighbour nor a certificate
    // bound can suffer -- and without a branch per entry the eight loads go out together.  (With `if (e < cnt)` around each offer
    // the compiler sank every load into its branch: a leaf scan was a chain of up to eight dependent memory round trips.)
    float4 v[LEAF_CAP];
#pragma unroll
    for (int e = 0; e < LEAF_CAP; e++) v[e] = gload16<float4>(p + e);
#pragma unroll
    for (int e = 0; e < LEAF_CAP; e++) col.offer(d2f(qx, qy, qz, v[e].x, v[e].y, v[e].z), (int)f2u(v[e].w));
    (void)cnt;
    return;
  }
  // the other collectors (k-NN lists, radius sums) must not see a point twice: only the leaf's own entries are offered, but every
  // distance is formed unconditionally so that the loads cannot sink into the branches (the array is padded: always in bounds)
  float4 v[LEAF_CAP];
  float dd[LEAF_CAP];
#pragma unroll
  for (int e = 0; e < LEAF_CAP; e++) v[e] = gload16<float4>(p + e);
#pragma unroll
  for (int e = 0; e < LEAF_CAP; e++) dd[e] = d2f(qx, qy, qz, v[e].x, v[e].y, v[e].z);
#pragma unroll
  for (int e = 0; e < LEAF_CAP; e++)
    if (e < cnt) {
      if constexpr (Collector::kXyz) col.offer_xyz(dd[e], v[e].x, v[e].y, v[e].z);
      else col.offer(dd[e], (int)f2u(v[e].w));
    }
}

// ---- start grid (see TreeHeader) --------------------------------------------------------------------------------------------
LH_HD int grid_index(int level, int cx, int cy, int cz) {
  return grid_off(level) + (((cx << level) | cy) << level | cz);
}
constexpr int32_t GRID_USE_ROOT = NO_CHILD - 1;   // grid_start: the candidate is too far for the coarsest table, walk from the root
// Where a WARM 1-NN walk starts.  `col` holds a real candidate at squared distance bd (finite).  The table level is the finest one
// whose cells are wider than the ball's diameter, so along every axis the ball crosses AT MOST one face of the query's cell; the
// (up to 7) neighbour cells behind crossed faces are pushed with a lower bound of their box distance as key (the walk's own pop
// prunes them against the bound of the moment and tells the collector), every other point of the cloud lies behind a face that
// is not crossed or two cells away, and the collector is told those bounds too (skip), so its certificate stays a bound on
// EVERY other point.  Returns the reference of the query's own cell (GRID_EMPTY: nothing there, start with a pop).
// Exactness: a point binned below coarse cell c (key coordinate < c << sh, a truncated FLOAT product) is, in real numbers, at
// least (g - (c << sh) - 4e-4) key cells below a query whose float key coordinate is g; the faces are moved by GRID_SLACK = 4e-3
// cells, the bounds are formed with the operation order of d2f ((x^2 + y^2) + z^2, monotone rounding), so bound <= float distance
// of every point behind the face, exactly like the node boxes (boxd2_q).  A query outside the grid belongs to the nearest boundary
// cell: the face it lies beyond counts as crossed (bound 0) and has no cells behind it.
template <class Collector, class Push>
LH_HD int32_t grid_start(const float* org, float key_sc, float key_inv, const int32_t* __restrict__ grid, float qx, float qy, float qz,
                         Collector& col, Push&& push) {
  const float bd = col.bound();
  const float ru = sqrt_bound(bd) * key_sc * 1.000001f + 2.0f * GRID_SLACK;   // the ball's radius in key cells, rounded up
  if (!(ru < 63.5f)) return GRID_USE_ROOT;
  int sh = 5 + (ru >= 15.5f ? 1 : 0) + (ru >= 31.5f ? 1 : 0);            // key-cell bits inside one table cell: wider than the ball
  if (GRID_FINEST >= 6 && ru < 7.5f) sh = 4;
  if (GRID_FINEST >= 7 && ru < 3.5f) sh = 3;
  const int level = 10 - sh, G = 1 << level;
  const float cell = (float)(1 << sh);
  const float q[3] = {qx, qy, qz};
  int home = grid_off(level);
  int cm = 0, dstep[3];
  float sq[3];   // squared metres to the crossed face (a lower bound), per axis
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float g = (q[a] - org[a]) * key_sc;                     // the key coordinate spatial_key30 truncates
    const int ci = (int)fminf(fmaxf(g, 0.0f), 1023.0f) >> sh;
    const float f_lo = g - (float)(ci << sh), f_hi = cell - f_lo; // key cells to the two faces (negative: the query lies beyond)
    const float l_lo = fmaxf((f_lo - GRID_SLACK) * key_inv, 0.0f), l_hi = fmaxf((f_hi - GRID_SLACK) * key_inv, 0.0f);
    const float s_lo = l_lo * l_lo, s_hi = l_hi * l_hi;
    const bool x_lo = s_lo <= bd, x_hi = !x_lo && s_hi <= bd;     // crossed faces (never both: the cell is wider than the ball)
    col.skip(x_lo ? s_hi : (x_hi ? s_lo : fminf(s_lo, s_hi)));    // everything behind the nearest face that is NOT crossed
    int dir = x_lo ? -1 : (x_hi ? 1 : 0);
    if (ci + dir < 0 || ci + dir >= G) dir = 0;                   // beyond the grid's own boundary: no cells, no points, nothing to bound
    if (dir) {
      const float l_far = ((x_lo ? f_lo : f_hi) + (cell - GRID_SLACK)) * key_inv;   // two cells away behind the crossed face
      col.skip(l_far * l_far);
      cm |= 1 << a;
    }
    sq[a] = x_lo ? s_lo : s_hi;
    const int stride_bits = (2 - a) * level;
    dstep[a] = dir * (1 << stride_bits);
    home += ci << stride_bits;
  }
  const int32_t href = gld(grid + home);   // (issued before the neighbour loop: its latency runs beside that loop's own table reads)
  // the neighbour cells behind crossed faces: the non-empty subsets of the crossed axes (usually one, at most seven)
  for (int m = cm; m; m = (m - 1) & cm) {
    const float lb = (((m & 1) ? sq[0] : 0.0f) + ((m & 2) ? sq[1] : 0.0f)) + ((m & 4) ? sq[2] : 0.0f);
    if (lb <= bd) {
      const int32_t ref = gld(grid + home + ((m & 1) ? dstep[0] : 0) + ((m & 2) ? dstep[1] : 0) + ((m & 4) ? dstep[2] : 0));
      if (ref != GRID_EMPTY) push(f2u(lb) & ~3u, ref);
    } else
      col.skip(lb);
  }
  return href;
}

// Nearest-child descent to ONE leaf and a scan of it: a good candidate, not the neighbour (nothing is stacked, nothing pruned).  What a
// cold search starts with anyway (tree_search's kGreedy phase); on its own it is a seed: any target point is a valid warm-start candidate.
template <class Collector, bool kGrid = false>
LH_HD void tree_descend(const TreeView& t, float qx, float qy, float qz, Collector& col) {
  const float INF = inf_f();
  TreeHeader h;
  h.root = gld(&t.hdr->root);
  h.org[0] = gld(&t.hdr->org[0]); h.org[1] = gld(&t.hdr->org[1]); h.org[2] = gld(&t.hdr->org[2]);
  h.inv = gld(&t.hdr->inv); h.scl2 = gld(&t.hdr->scl2);
  const GridQuery gq = grid_query(h, qx, qy, qz);
  const float scl2 = h.scl2;
  int32_t r = h.root;
  if constexpr (kGrid) {   // a seed only has to be A point near the query: the subtree of the query's own level-5 cell, if it holds anything
    if (gld(&t.hdr->grid_on)) {
      const float ksc = gld(&t.hdr->key_sc);
      const int cx = (int)fminf(fmaxf((qx - h.org[0]) * ksc, 0.0f), 1023.0f) >> 5, cy = (int)fminf(fmaxf((qy - h.org[1]) * ksc, 0.0f), 1023.0f) >> 5,
                cz = (int)fminf(fmaxf((qz - h.org[2]) * ksc, 0.0f), 1023.0f) >> 5;
      const int32_t g = gld(t.grid() + grid_index(5, cx, cy, cz));
      if (g != GRID_EMPTY) r = g;
    }
  }
  while (r >= 0) {
    const NodeX& nd = t.nodes[r];
    const uint4 a = gload16<uint4>(nd.lo_xy);
    const uint4 b = gload16<uint4>(nd.hi_xy);
    const uint4 c = gload16<uint4>(nd.z_lohi);
    const int4 ch = gload16<int4>(nd.child);
    float d0 = boxd2_q(gq, a.x, b.x, c.x, scl2), d1 = boxd2_q(gq, a.y, b.y, c.y, scl2);
    float d2 = boxd2_q(gq, a.z, b.z, c.z, scl2), d3 = boxd2_q(gq, a.w, b.w, c.w, scl2);
    d1 = ch.y == NO_CHILD ? INF : d1;
    d2 = ch.z == NO_CHILD ? INF : d2;
    d3 = ch.w == NO_CHILD ? INF : d3;
    float dm = d0; int32_t rm = ch.x;          // child 0 always exists
    if (d1 < dm) { dm = d1; rm = ch.y; }
    if (d2 < dm) { dm = d2; rm = ch.z; }
    if (d3 < dm) { dm = d3; rm = ch.w; }
    r = rm;
  }
  scan_leaf(t, r, qx, qy, qz, col);
}

// kGrid (the GICP sweeps' warm 1-NN walks): start at the query's cell of the start grid instead of at the root (grid_start)
template <class Collector, bool kGrid = false>
LH_HD void tree_search(const TreeView& t, float qx, float qy, float qz, Collector& col, uint64_t* stack, int stride) {
  const uint32_t NONE = 0xffffffffu;
  const float INF = inf_f();
  TreeHeader h;   // only the fields the walk needs (uniform address: scalar loads once the pointer is known to be global)
  h.root = gld(&t.hdr->root);
  h.org[0] = gld(&t.hdr->org[0]); h.org[1] = gld(&t.hdr->org[1]); h.org[2] = gld(&t.hdr->org[2]);
  h.inv = gld(&t.hdr->inv); h.scl2 = gld(&t.hdr->scl2);
  const int32_t root = h.root;
  const GridQuery gq = grid_query(h, qx, qy, qz);
  const float scl2 = h.scl2;
  // box distances of a node's four children (+inf for an absent child)
  auto child_dists = [&](const NodeX& nd, int4& ch, float& d0, float& d1, float& d2, float& d3) {
    const uint4 a = gload16<uint4>(nd.lo_xy);
    const uint4 b = gload16<uint4>(nd.hi_xy);
    const uint4 c = gload16<uint4>(nd.z_lohi);
    ch = gload16<int4>(nd.child);
    d0 = boxd2_q(gq, a.x, b.x, c.x, scl2);
    d1 = boxd2_q(gq, a.y, b.y, c.y, scl2);
    d2 = boxd2_q(gq, a.z, b.z, c.z, scl2);
    d3 = boxd2_q(gq, a.w, b.w, c.w, scl2);
    d1 = ch.y == NO_CHILD ? INF : d1;   // child 0 always exists; computed unconditionally so that the four 16-B loads stay
    d2 = ch.z == NO_CHILD ? INF : d2;   // whole and no branch (with a second, dependent round of loads) appears
    d3 = ch.w == NO_CHILD ? INF : d3;
  };
  if constexpr (Collector::kGreedy) {
    if (!(col.bound() < INF)) {  // cold start: nearest-child descent to one leaf, nothing stacked, nothing pruned
      int32_t r = root;
      if constexpr (kGrid) {      // ... from the query's own level-5 cell of the start grid, if that holds anything (any point is a valid candidate)
        if (gld(&t.hdr->grid_on)) {
          const float ksc = gld(&t.hdr->key_sc);
          const int cx = (int)fminf(fmaxf((qx - h.org[0]) * ksc, 0.0f), 1023.0f) >> 5, cy = (int)fminf(fmaxf((qy - h.org[1]) * ksc, 0.0f), 1023.0f) >> 5,
                    cz = (int)fminf(fmaxf((qz - h.org[2]) * ksc, 0.0f), 1023.0f) >> 5;
          const int32_t g = gld(t.grid() + grid_index(5, cx, cy, cz));
          if (g != GRID_EMPTY) r = g;
        }
      }
      while (r >= 0) {
        int4 ch;
        float d0, d1, d2, d3;
        child_dists(t.nodes[r], ch, d0, d1, d2, d3);
        float dm = d0; int32_t rm = ch.x;          // child 0 always exists
        if (d1 < dm) { dm = d1; rm = ch.y; }
        if (d2 < dm) { dm = d2; rm = ch.z; }
        if (d3 < dm) { dm = d3; rm = ch.w; }
        r = rm;
      }
      scan_leaf(t, r, qx, qy, qz, col);
    }
  }
  int sp = 0;
  uint64_t spill[SPILL_MAX];
#if defined(__HIP_DEVICE_COMPILE__)
  // the kernels' stack base is LDS: say so, or the pop below (LDS entry or spilled entry) becomes ONE flat load through a
  // selected generic pointer, with the flat path's latency on every pop
  __attribute__((address_space(3))) uint64_t* const lstack = (__attribute__((address_space(3))) uint64_t*)stack;
#else
  uint64_t* const lstack = stack;
#endif
  auto push = [&](uint32_t key, int32_t ref) {
    uint64_t e = ((uint64_t)key << 32) | (uint32_t)ref;
    if (sp < LDS_STACK) lstack[sp * stride] = e;
    else spill[sp - LDS_STACK] = e;
    sp++;
  };
  const int32_t DONE = NO_CHILD;  // never a valid reference (internal indices are < 2^31 - 1)
  auto pop = [&]() -> int32_t {
    for (;;) {
      if (sp == 0) return DONE;
      --sp;
      uint64_t e;
      if (sp < LDS_STACK) e = lstack[sp * stride];
      else e = spill[sp - LDS_STACK];
      float dk = u2f((uint32_t)(e >> 32) & ~3u);  // key = distance bits with the child slot in the two low mantissa bits (rounded DOWN)
      if (dk <= col.bound()) return (int32_t)(uint32_t)e;
      col.skip(dk);
    }
  };
  int32_t ref = root;
  if constexpr (kGrid) {
    if (col.bound() < INF && gld(&t.hdr->grid_on)) {
      const int32_t g = grid_start(h.org, gld(&t.hdr->key_sc), gld(&t.hdr->key_inv), t.grid(), qx, qy, qz, col, push);
      if (g != GRID_USE_ROOT) ref = (g == GRID_EMPTY) ? pop() : g;
    }
  }
  auto node_step = [&]() {   // one internal node: its children's box distances, the nearest goes on, the others are stacked
    col.count_node(0);
    int4 ch;
    float d0, d1, d2, d3;
    child_dists(t.nodes[ref], ch, d0, d1, d2, d3);
    float bd = col.bound();
    bool v0 = d0 <= bd && d0 < INF, v1 = d1 <= bd && d1 < INF, v2 = d2 <= bd && d2 < INF, v3 = d3 <= bd && d3 < INF;
    // 32-bit sort keys: float bits of the (non-negative) box distance with the two low mantissa bits replaced by the
    // child slot -> a 4-key sort is ten v_min/v_max_u32
    uint32_t k0 = v0 ? ((f2u(d0) & ~3u) | 0u) : NONE;
    uint32_t k1 = v1 ? ((f2u(d1) & ~3u) | 1u) : NONE;
    uint32_t k2 = v2 ? ((f2u(d2) & ~3u) | 2u) : NONE;
    uint32_t k3 = v3 ? ((f2u(d3) & ~3u) | 3u) : NONE;
    col.skip(v0 ? INF : d0); col.skip(v1 ? INF : d1); col.skip(v2 ? INF : d2); col.skip(v3 ? INF : d3);
    uint32_t a, b;
    a = k0 < k1 ? k0 : k1; b = k0 < k1 ? k1 : k0; k0 = a; k1 = b;
    a = k2 < k3 ? k2 : k3; b = k2 < k3 ? k3 : k2; k2 = a; k3 = b;
    a = k0 < k2 ? k0 : k2; b = k0 < k2 ? k2 : k0; k0 = a; k2 = b;
    a = k1 < k3 ? k1 : k3; b = k1 < k3 ? k3 : k1; k1 = a; k3 = b;
    a = k1 < k2 ? k1 : k2; b = k1 < k2 ? k2 : k1; k1 = a; k2 = b;
    auto child_of = [&](uint32_t k) -> int32_t {
      uint32_t sl = k & 3u;
      return sl == 0 ? ch.x : (sl == 1 ? ch.y : (sl == 2 ? ch.z : ch.w));
    };
    if (k1 != NONE) {  // valid keys sort first: k1 invalid => k2, k3 invalid
      if (k2 != NONE) {
        if (k3 != NONE) push(k3, child_of(k3));
        push(k2, child_of(k2));
      }
      push(k1, child_of(k1));
    }
    ref = (k0 != NONE) ? child_of(k0) : pop();
  };
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (kGrid) {
    // The sweeps' walks: the WAVE takes a node step or a leaf step, whichever more of its lanes are waiting for (a uniform branch on
    // two ballots).  The plain loop below runs a node phase until the LAST lane has reached a leaf -- half of its iterations had
    // <= 8 lanes at work (host model, tools/model/grid_start_model.cpp: 22.2 -> 17.6 wave iterations in the cold sweep, 12.2 -> 11.2
    // in the warm ones).  Visits, order and pruning of a lane are unchanged: same neighbours, same certificates.
    for (;;) {
      const bool is_node = ref >= 0 && ref != DONE, is_leaf = ref < 0;
      const unsigned long long mn = __ballot(is_node), ml = __ballot(is_leaf);
      if ((mn | ml) == 0ull) return;
      if (__popcll(mn) >= __popcll(ml)) {
        if (is_node) node_step();
      } else if (is_leaf) {
        scan_leaf(t, ref, qx, qy, qz, col);
        ref = pop();
      }
    }
  }
#endif
  for (;;) {
    while (ref >= 0 && ref != DONE) node_step();
    if (ref == DONE) return;
    scan_leaf(t, ref, qx, qy, qz, col);
    ref = pop();
  }
}

// The same traversal cut into steps for a caller that drives the loop itself (k_walk's persistent lanes: a lane starts its next
// query while the others are in the middle of theirs).  WalkStack = tree_search's stack (first N entries in LDS at
// base[entry * stride], deeper ones in private memory); node_visit = one iteration of tree_search's inner loop: the child to
// descend into, or the next stacked subtree that can still matter, or NO_CHILD when the search has ended.  Same keys, same order,
// same pruning rule, same skip() calls as tree_search, so a collector ends with the same state.
template <int N>
struct WalkStack {
#if defined(__HIP_DEVICE_COMPILE__)
  __attribute__((address_space(3))) uint64_t* base;
#else
  uint64_t* base;
#endif
  int stride;
  int sp;
  uint64_t spill[SPILL_MAX + LDS_STACK - N];
#if defined(__HIP_DEVICE_COMPILE__)
  LH_HD WalkStack(uint64_t* b, int st) : base((__attribute__((address_space(3))) uint64_t*)b), stride(st), sp(0) {}
#else
  LH_HD WalkStack(uint64_t* b, int st) : base(b), stride(st), sp(0) {}
#endif
  LH_HD void push(uint32_t key, int32_t ref) {
    uint64_t e = ((uint64_t)key << 32) | (uint32_t)ref;
    if (sp < N) base[sp * stride] = e;
    else spill[sp - N] = e;
    sp++;
  }
  template <class Collector>
  LH_HD int32_t pop(Collector& col) {
    for (;;) {
      if (sp == 0) return NO_CHILD;
      --sp;
      uint64_t e;
      if (sp < N) e = base[sp * stride];
      else e = spill[sp - N];
      float dk = u2f((uint32_t)(e >> 32) & ~3u);
      if (dk <= col.bound()) return (int32_t)(uint32_t)e;
      col.skip(dk);
    }
  }
};
template <class Collector, class Stack>
LH_HD int32_t node_visit(const NodeX& nd, const GridQuery& gq, float scl2, Collector& col, Stack& stk) {
  const uint32_t NONE = 0xffffffffu;
  const float INF = inf_f();
  col.count_node(0);
  const uint4 a = gload16<uint4>(nd.lo_xy);
  const uint4 b = gload16<uint4>(nd.hi_xy);
  const uint4 c = gload16<uint4>(nd.z_lohi);
  const int4 ch = gload16<int4>(nd.child);
  float d0 = boxd2_q(gq, a.x, b.x, c.x, scl2);
  float d1 = boxd2_q(gq, a.y, b.y, c.y, scl2);
  float d2 = boxd2_q(gq, a.z, b.z, c.z, scl2);
  float d3 = boxd2_q(gq, a.w, b.w, c.w, scl2);
  d1 = ch.y == NO_CHILD ? INF : d1;
  d2 = ch.z == NO_CHILD ? INF : d2;
  d3 = ch.w == NO_CHILD ? INF : d3;
  const float bd = col.bound();
  const bool v0 = d0 <= bd && d0 < INF, v1 = d1 <= bd && d1 < INF, v2 = d2 <= bd && d2 < INF, v3 = d3 <= bd && d3 < INF;
  uint32_t k0 = v0 ? ((f2u(d0) & ~3u) | 0u) : NONE;
  uint32_t k1 = v1 ? ((f2u(d1) & ~3u) | 1u) : NONE;
  uint32_t k2 = v2 ? ((f2u(d2) & ~3u) | 2u) : NONE;
  uint32_t k3 = v3 ? ((f2u(d3) & ~3u) | 3u) : NONE;
  col.skip(v0 ? INF : d0); col.skip(v1 ? INF : d1); col.skip(v2 ? INF : d2); col.skip(v3 ? INF : d3);
  uint32_t x, y;
  x = k0 < k1 ? k0 : k1; y = k0 < k1 ? k1 : k0; k0 = x; k1 = y;
  x = k2 < k3 ? k2 : k3; y = k2 < k3 ? k3 : k2; k2 = x; k3 = y;
  x = k0 < k2 ? k0 : k2; y = k0 < k2 ? k2 : k0; k0 = x; k2 = y;
  x = k1 < k3 ? k1 : k3; y = k1 < k3 ? k3 : k1; k1 = x; k3 = y;
  x = k1 < k2 ? k1 : k2; y = k1 < k2 ? k2 : k1; k1 = x; k2 = y;
  auto child_of = [&](uint32_t k) -> int32_t {
    uint32_t sl = k & 3u;
    return sl == 0 ? ch.x : (sl == 1 ? ch.y : (sl == 2 ? ch.z : ch.w));
  };
  if (k1 != NONE) {
    if (k2 != NONE) {
      if (k3 != NONE) stk.push(k3, child_of(k3));
      stk.push(k2, child_of(k2));
    }
    stk.push(k1, child_of(k1));
  }
  return (k0 != NONE) ? child_of(k0) : stk.pop(col);
}

// ---- index build, per-element steps shared by the build kernels and the host-side check ---------------------------------
// sorted keys: (cloud id << 32) | 30-bit Morton key, so a batch of clouds is one sorted array and no cell spans two clouds
constexpr int KEY_PREFIX_MIN = 34;  // 64 - 30: shortest prefix that still pins the cloud id (and the two unused bits)

// does sorted position g start a leaf?  Leaf = the largest prefix cell around g with <= LEAF_CAP points.
LH_HD uint32_t leafcell_flag(const uint64_t* __restrict__ keys, int64_t total, int64_t g) {
  if (g == 0) return 1u;
  const uint64_t K = keys[g];
  int L[LEAF_CAP], R[LEAF_CAP];
#pragma unroll
  for (int k = 1; k <= LEAF_CAP; k++) {
    L[k - 1] = (g - k >= 0) ? clz64(K ^ keys[g - k]) : 0;
    R[k - 1] = (g + k < total) ? clz64(K ^ keys[g + k]) : 0;
  }
  // The cell of prefix length p around g holds 1 + #{k: L[k] >= p} + #{k: R[k] >= p} points.  L and R are non-increasing in k (sorted
  // keys: the common prefix with a farther element is the minimum over the adjacent ones), so "more than LEAF_CAP points" means: for
  // some split a + b = LEAF_CAP the a-th left AND the b-th right neighbour still share p bits, i.e. p <= max_a min(L[a-1], R[b-1]) =: P.
  // The smallest prefix that fits is P + 1 (round 2 found it by bisection over p: five passes over the sixteen values).
  const int c_prev = L[0];
  int P = R[LEAF_CAP - 1] < L[LEAF_CAP - 1] ? L[LEAF_CAP - 1] : R[LEAF_CAP - 1];   // a = 0 (all on the right) or a = LEAF_CAP (all on the left)
#pragma unroll
  for (int a = 1; a < LEAF_CAP; a++) {
    const int m = L[a - 1] < R[LEAF_CAP - 1 - a] ? L[a - 1] : R[LEAF_CAP - 1 - a];
    P = P < m ? m : P;
  }
  if (P >= 64) return (c_prev < 64 || (g % LEAF_CAP) == 0) ? 1u : 0u;  // > LEAF_CAP identical keys: fixed-size chunks
  const int lo = P + 1 < KEY_PREFIX_MIN ? KEY_PREFIX_MIN : P + 1;
  return c_prev < lo ? 1u : 0u;
}

// Karras' binary radix tree over the n_leaves leaf keys: internal node i in [0, n_leaves - 1)
LH_HD int radix_delta(const uint64_t* __restrict__ lkey, int n_leaves, int i, int j) {
  if (j < 0 || j >= n_leaves) return -1;
  uint64_t x = lkey[i] ^ lkey[j];
  return x ? clz64(x) : 64 + clz32((uint32_t)i ^ (uint32_t)j);
}
// children are returned as binary-tree references: >= 0 internal node, < 0 leaf ~index; [lo, hi] = leaf range covered
LH_HD void radix_node(const uint64_t* __restrict__ lkey, int n_leaves, int i, int& left, int& right, int& lo, int& hi, int* delta_out = nullptr) {
  int d = (radix_delta(lkey, n_leaves, i, i + 1) - radix_delta(lkey, n_leaves, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = radix_delta(lkey, n_leaves, i, i - d);
  int lmax = 2;
  while (radix_delta(lkey, n_leaves, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (radix_delta(lkey, n_leaves, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = radix_delta(lkey, n_leaves, i, j);
  if (delta_out) *delta_out = dnode;   // leading bits of the 64-bit key shared by the whole range (>= 64: identical keys)
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (radix_delta(lkey, n_leaves, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  int gamma = i + s * d + (d < 0 ? -1 : 0);
  lo = i < j ? i : j;
  hi = i < j ? j : i;
  left = (lo == gamma) ? ~gamma : gamma;
  right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
}

// leading bits of the 30-bit key shared by a node's range, from radix_node's delta (the 64-bit key = cloud id << 32 | key30):
// -1 for a node that joins two clouds, 30 for identical keys
LH_HD int key_common(int delta) { return delta < 34 ? -1 : (delta > 64 ? 30 : delta - 34); }
// how many of the start grid's levels (3, 4, 5 = 9, 12, 15 prefix bits) a node with `com` common bits lies inside a cell of
LH_HD int grid_depth(int com) {
  int d = (com >= 9 ? 1 : 0) + (com >= 12 ? 1 : 0) + (com >= 15 ? 1 : 0);
  if (GRID_FINEST >= 6) d += com >= 18 ? 1 : 0;
  if (GRID_FINEST >= 7) d += com >= 21 ? 1 : 0;
  return d;
}
// Start-grid entries contributed by ONE child of binary node i (com_i common key bits); called for both children of every node of
// a cloud.  An internal child whose keys share >= 3 l bits while i's do not IS the level-l cell of its keys: the keys of a radix-tree
// node are all keys with its prefix, so the child holds exactly the cloud's points inside that cell.  A LEAF child is the cell of
// the (com_i + 1)-bit prefix, which may span several table cells (a few points in a large empty region): every one of them gets the
// leaf.  key30_c = any key of the child, ref_c = the child reference the walk uses (cloud-local node index / leaf reference).
LH_HD bool grid_child_cells(int l, int com_i, bool child_is_leaf, int com_c, uint32_t key30_c, uint32_t lo3[3], uint32_t hi3[3]) {
  const int bits = 3 * l, sh = 10 - l;
  if (com_i >= bits) return false;                      // i lies inside a level-l cell already: an ancestor's child made that entry
  if (!child_is_leaf && com_c < bits) return false;     // the child still spans several level-l cells: its descendants make the entries
  uint32_t c0[3], c1[3];
  morton_cell(key30_c, child_is_leaf ? (com_i + 1 < bits ? com_i + 1 : bits) : bits, c0, c1);
#pragma unroll
  for (int a = 0; a < 3; a++) { lo3[a] = c0[a] >> sh; hi3[a] = c1[a] >> sh; }
  return true;
}
LH_HD void grid_fill_child(int32_t* grid, int com_i, bool child_is_leaf, int com_c, uint32_t key30_c, int32_t ref_c) {
#pragma unroll
  for (int l = GRID_FINEST; l >= 3; l--) {
    uint32_t lo3[3], hi3[3];
    if (!grid_child_cells(l, com_i, child_is_leaf, com_c, key30_c, lo3, hi3)) continue;
    for (uint32_t x = lo3[0]; x <= hi3[0]; x++)
      for (uint32_t y = lo3[1]; y <= hi3[1]; y++)
        for (uint32_t z = lo3[2]; z <= hi3[2]; z++) grid[grid_index(l, (int)x, (int)y, (int)z)] = ref_c;
  }
}
// ... and by the cloud's root, for the levels at which the WHOLE cloud lies inside one cell
LH_HD void grid_fill_root(int32_t* grid, int com_root, uint32_t key30, int32_t ref_root) {
#pragma unroll
  for (int l = GRID_FINEST; l >= 3; l--) {
    if (com_root < 3 * l) continue;
    uint32_t c0[3], c1[3];
    morton_cell(key30, 3 * l, c0, c1);
    grid[grid_index(l, (int)(c0[0] >> (10 - l)), (int)(c0[1] >> (10 - l)), (int)(c0[2] >> (10 - l)))] = ref_root;
  }
}

struct Nn1Collector {
  float bd;
  int bi;
  static constexpr bool kXyz = false;
  static constexpr bool kGreedy = true;
  LH_HD float bound() const { return bd; }
  LH_HD void offer(float d, int id) {  // branch-free: selects instead of exec-mask branches in the 8-point leaf scan
    bool better = (d < bd) | ((d == bd) & (id < bi));
    bd = better ? d : bd;
    bi = better ? id : bi;
  }
  LH_HD void skip(float) {}
  LH_HD void count_node(int) {}
  LH_HD void count_leaf() {}
};

// instrumented 1-NN (lh_debug_traversal_stats): counts node / leaf visits of the query
struct Nn1CountCollector {
  float bd;
  int bi;
  int nodes, leaves;
  static constexpr bool kXyz = false;
  static constexpr bool kGreedy = true;
  LH_HD float bound() const { return bd; }
  LH_HD void offer(float d, int id) {
    if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
  }
  LH_HD void skip(float) {}
  int per_level[MAX_DEPTH];
  LH_HD void count_node(int l) { nodes++; per_level[l]++; }
  LH_HD void count_leaf() { leaves++; }
};

// 1-NN plus a certificate: `lb` = lower bound on the squared distance of every point other than the winner
// (second best among the examined points, and the box distance of every pruned subtree).  A later query q' within
// distance e of this query keeps the same winner whenever d(q', winner) + e < sqrt(lb) (triangle inequality), which
// lets the next GICP sweep skip the traversal for that point without changing its result.
struct Nn1CertCollector {
  float bd;
  int bi;
  float lb;
  static constexpr bool kXyz = false;
  static constexpr bool kGreedy = true;
  LH_HD float bound() const { return bd; }
  LH_HD void offer(float d, int id) {  // branch-free
    bool same = id == bi;                                   // the warm-start candidate met again: no-op
    bool better = (d < bd) | ((d == bd) & (id < bi));
    float runner = better ? bd : d;                         // dethroned winner, or the rejected point, is a runner-up
    lb = same ? lb : fminf(lb, runner);                     // (bd is +inf while there is no winner yet)
    bool take = better & !same;
    bd = take ? d : bd;
    bi = take ? id : bi;
  }
  LH_HD void skip(float d) { lb = fminf(lb, d); }
  LH_HD void count_node(int) {}
  LH_HD void count_leaf() {}
};

// radius search for the normal filter's radius mode (normal_computation.cc:71-74): every point with d2 < r2 feeds the
// nine float moment accumulators of computeMeanAndCovarianceMatrix directly (no neighbour list: the count is unbounded).
struct RadiusMomentCollector {
  float r2;
  int cnt;
  float a[9];
  static constexpr bool kXyz = true;
  static constexpr bool kGreedy = false;
  LH_HD float bound() const { return r2; }
  LH_HD void offer_xyz(float d, float x, float y, float z) {
    if (d < r2) {  // FLANN RadiusResultSet: strict
      cnt++;
      a[0] += x * x; a[1] += x * y; a[2] += x * z;
      a[3] += y * y; a[4] += y * z; a[5] += z * z;
      a[6] += x; a[7] += y; a[8] += z;
    }
  }
  LH_HD void skip(float) {}
  LH_HD void count_node(int) {}
  LH_HD void count_leaf() {}
};

// k best (d2, id) ascending, lexicographic; storage strided so that a workgroup can keep the lists in LDS
// as [element][thread] (conflict-free) -- stride 1 on the host.
struct KnnCollector {
  float* kd;
  int* ki;
  int k, stride, cnt;
  static constexpr bool kXyz = false;
  static constexpr bool kGreedy = false;
  LH_HD float bound() const { return cnt < k ? inf_f() : kd[(k - 1) * stride]; }
  LH_HD void skip(float) {}
  LH_HD void count_node(int) {}
  LH_HD void count_leaf() {}
  LH_HD void offer(float d, int id) {
    if (id == 0x7fffffff) return;  // padding point
    if (cnt == k) {
      float ld = kd[(k - 1) * stride];
      int li = ki[(k - 1) * stride];
      if (!(d < ld || (d == ld && id < li))) return;
    }
    int p = (cnt < k) ? cnt++ : k - 1;
    while (p > 0) {
      float pd = kd[(p - 1) * stride];
      int pi = ki[(p - 1) * stride];
      if (!(d < pd || (d == pd && id < pi))) break;
      kd[p * stride] = pd;
      ki[p * stride] = pi;
      p--;
    }
    kd[p * stride] = d;
    ki[p * stride] = id;
  }
};

// Register-resident k-best list (device fast path of the k-NN kernels).  A wave scans leaf points in lock-step, so an
// insertion loop through LDS (KnnCollector above) is replayed for the whole wave whenever ANY lane accepts a point and
// every shift is a dependent LDS round trip -- measured 1.9 ms per 100 k queries at k = 20.  Here the sorted list lives
// in KCAP registers and an offer is a branch-free compare/select chain, executed only if some lane accepts.
template <int KCAP>
struct KnnRegCollector {
  float d[KCAP];
  int id[KCAP];
  int k;
  float kd;  // cached (k-1)-th entry = pruning bound
  int ki;
  LH_HD void init(int k_) {
    k = k_;
#pragma unroll
    for (int j = 0; j < KCAP; j++) { d[j] = inf_f(); id[j] = 0x7fffffff; }
    kd = inf_f();
    ki = 0x7fffffff;
  }
  static constexpr bool kXyz = false;
  static constexpr bool kGreedy = false;
  LH_HD float bound() const { return kd; }
  LH_HD void skip(float) {}
  LH_HD void count_node(int) {}
  LH_HD void count_leaf() {}
  LH_HD void offer(float dd, int ii) {
    bool acc = (dd < kd) | ((dd == kd) & (ii < ki));
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__any(acc)) return;
#else
    if (!acc) return;
#endif
    bool prev_lt = true;
    float pd = 0.f;
    int pi = 0;
#pragma unroll
    for (int j = 0; j < KCAP; j++) {
      bool lt = ((d[j] < dd) | ((d[j] == dd) & (id[j] < ii))) | !acc;  // entry j stays where it is
      float nd = lt ? d[j] : (prev_lt ? dd : pd);
      int ni = lt ? id[j] : (prev_lt ? ii : pi);
      pd = d[j]; pi = id[j];
      d[j] = nd; id[j] = ni;
      prev_lt = lt;
    }
#pragma unroll
    for (int j = 0; j < KCAP; j++)
      if (j == k - 1) { kd = d[j]; ki = id[j]; }
  }
  // copy the k best into strided arrays (LDS on the device), count of valid entries returned
  LH_HD int dump(float* od, int* oi, int stride) const {
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < KCAP; j++)
      if (j < k) {
        od[j * stride] = d[j];
        oi[j * stride] = id[j];
        cnt += (id[j] != 0x7fffffff) ? 1 : 0;
      }
    return cnt;
  }
};

// ---- 3x3 double helpers (row-major) ------------------------------------------------------------------
// C = I - (1-eps) n n^T  (CalculateCovarianceFromNormals restated; zero / non-finite normal => I)
LH_HD void cov_from_normal(float nx, float ny, float nz, double eps, double* C) {
  double n0 = nx, n1 = ny, n2 = nz;
  double l2 = n0 * n0 + n1 * n1 + n2 * n2;
  C[0] = 1.0; C[1] = 0.0; C[2] = 0.0; C[3] = 0.0; C[4] = 1.0; C[5] = 0.0; C[6] = 0.0; C[7] = 0.0; C[8] = 1.0;
  if (!(l2 > 0.0) || !(l2 < 1.0e300)) return;
  double inv = 1.0 / sqrt(l2);
  n0 *= inv; n1 *= inv; n2 *= inv;
  double s = 1.0 - eps;
  C[0] -= s * n0 * n0; C[1] -= s * n0 * n1; C[2] -= s * n0 * n2;
  C[3] -= s * n1 * n0; C[4] -= s * n1 * n1; C[5] -= s * n1 * n2;
  C[6] -= s * n2 * n0; C[7] -= s * n2 * n1; C[8] -= s * n2 * n2;
}
LH_HD void sym6_to_mat9(const double* s, double* C) {  // (00,01,02,11,12,22)
  C[0] = s[0]; C[1] = s[1]; C[2] = s[2]; C[3] = s[1]; C[4] = s[3]; C[5] = s[4]; C[6] = s[2]; C[7] = s[4]; C[8] = s[5];
}
LH_HD double cof3(const double* m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
// M = (R C1 R^T + C2)^-1 (gicp.hpp:488-493), Eigen's 3x3 cofactor inverse
LH_HD void mahalanobis(const double* R, const double* C1, const double* C2, double* Minv) {
  double M[9], t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) M[i * 3 + j] = R[i * 3 + 0] * C1[0 * 3 + j] + R[i * 3 + 1] * C1[1 * 3 + j] + R[i * 3 + 2] * C1[2 * 3 + j];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      t[i * 3 + j] = (M[i * 3 + 0] * R[j * 3 + 0] + M[i * 3 + 1] * R[j * 3 + 1] + M[i * 3 + 2] * R[j * 3 + 2]) + C2[i * 3 + j];
  double c00 = cof3(t, 0, 0), c10 = cof3(t, 1, 0), c20 = cof3(t, 2, 0);
  double det = c00 * t[0] + c10 * t[3] + c20 * t[6];
  double invdet = 1.0 / det;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Minv[i * 3 + j] = cof3(t, j, i) * invdet;
}

// cyclic Jacobi eigen-decomposition of a symmetric 3x3 (double); returns eigenvector of the smallest |eigenvalue|.
// Stands in for Eigen::JacobiSVD<Matrix3d> at gicp.hpp:140 (U's last column).
LH_HD void smallest_sv_vector3(const double* Ain, double* u) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; i++) A[i] = Ain[i];
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 3; k++) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  // ascending eigenvalue order first (like the oracle's sort), then smallest |ev|, first index on ties
  double ev[3] = {A[0], A[4], A[8]};
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 3; i++) {
    int m = i;
    for (int j = i + 1; j < 3; j++)
      if (ev[ord[j]] < ev[ord[m]]) m = j;
    int tmp = ord[i]; ord[i] = ord[m]; ord[m] = tmp;
  }
  int s = 0;
  for (int a = 1; a < 3; a++)
    if (fabs(ev[ord[a]]) < fabs(ev[ord[s]])) s = a;
  int col = ord[s];
  u[0] = V[0 * 3 + col]; u[1] = V[1 * 3 + col]; u[2] = V[2 * 3 + col];
}

}  // namespace lh
