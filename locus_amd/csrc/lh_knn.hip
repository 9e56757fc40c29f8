// lh_knn.hip -- K3: k nearest neighbours, the covariances of gicp.hpp:85-154 and the normals of normal_computation.cc:26-81 (gfx950, wave64).
// Compiled like lh_kernels.hip with -ffp-contract=off: the float distances round like the oracle's, so neighbour sets are bit-exact.
//   k_knn_block / k_knn_redo              every point of a cloud against its own cloud, one WAVE per 64 Morton-consecutive queries, any number
//                                         of clouds per launch (lh_knn_block.hpp): the normal filter and the recompute-covariance mode
//   k_knn / k_knn_cov / k_knn_normals     one query per lane (arbitrary queries, k > 32, and the redo list's engine)
//   k_radius_normals                      the filter's radius mode
#include <cstdlib>

#include "lh_kernels.hpp"
#include "lh_knn_block.hpp"
#include "lh_launch.hpp"

namespace lh {

// ===== K3: k-NN, covariances, normals ======================================================================
constexpr int KNN_BLOCK = 128;

template <int KCAP>
__device__ __forceinline__ int knn_search_regs(const TreeView& tv, float x, float y, float z, int k, float* kd, int* ki, uint64_t* stack) {
  KnnRegCollector<KCAP> col;
  col.init(k);
  tree_search(tv, x, y, z, col, stack, KNN_BLOCK);
  return col.dump(kd, ki, KNN_BLOCK);
}
// k best of one query into kd/ki ([k][KNN_BLOCK] LDS, already offset by threadIdx.x); returns the number found.
// KCAP = register-list capacity chosen by the host (smallest of 8 / 20 / 32 that holds k; 0 = LDS insertion list for k > 32),
// a template parameter of the kernels so that each instantiation only pays for its own registers.
template <int KCAP>
__device__ __forceinline__ int knn_search(const TreeView& tv, float x, float y, float z, int k, float* kd, int* ki, uint64_t* stack) {
  if constexpr (KCAP > 0) {
    return knn_search_regs<KCAP>(tv, x, y, z, k, kd, ki, stack);
  } else {
    KnnCollector col{kd, ki, k, KNN_BLOCK, 0};
    tree_search(tv, x, y, z, col, stack, KNN_BLOCK);
    return col.cnt;
  }
}
#define LH_KNN_DISPATCH(KERNEL, k, ...)                                              \
  do {                                                                              \
    if ((k) <= 8) hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__);                       \
    else if ((k) <= 20) hipLaunchKernelGGL(KERNEL<20>, __VA_ARGS__);                \
    else if ((k) <= 32) hipLaunchKernelGGL(KERNEL<32>, __VA_ARGS__);                \
    else hipLaunchKernelGGL(KERNEL<0>, __VA_ARGS__);                                \
  } while (0)

template <int KCAP>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn(const float4* __restrict__ q, int nq, TreeView tv, int k, int32_t* __restrict__ idx,
                                                   float* __restrict__ d2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* kd = reinterpret_cast<float*>(smem);
  int* ki = reinterpret_cast<int*>(smem + sizeof(float) * (size_t)k * KNN_BLOCK);
  int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  if (i >= nq) return;
  float4 p = q[i];
  uint64_t* lds_stack = reinterpret_cast<uint64_t*>(smem + (size_t)8 * k * KNN_BLOCK);
  struct { int cnt; } col;
  col.cnt = knn_search<KCAP>(tv, p.x, p.y, p.z, k, kd + threadIdx.x, ki + threadIdx.x, lds_stack + threadIdx.x);
  for (int e = 0; e < k; e++) {
    bool ok = e < col.cnt;
    idx[(size_t)i * k + e] = ok ? ki[e * KNN_BLOCK + threadIdx.x] : -1;
    d2[(size_t)i * k + e] = ok ? kd[e * KNN_BLOCK + threadIdx.x] : INFINITY;
  }
}
void launch_knn(const float4* q, int nq, TreeView tree, int k, int32_t* idx, float* d2, hipStream_t s) {
  size_t sh = (size_t)k * KNN_BLOCK * 8 + stack_lds_bytes(0, KNN_BLOCK);
  LH_KNN_DISPATCH(k_knn, k, dim3((nq + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), sh, s, q, nq, tree, k, idx, d2);
}

// computeCovariances k-NN branch (gicp.hpp:85-154)
template <int KCAP>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_cov(const float4* __restrict__ xyz, int n, int n_pad, TreeView tv, int k, double eps,
                                                       double* __restrict__ cov6) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* kd = reinterpret_cast<float*>(smem);
  int* ki = reinterpret_cast<int*>(smem + sizeof(float) * (size_t)k * KNN_BLOCK);
  int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  if (i >= n) return;
  float4 p = xyz[i];
  uint64_t* lds_stack = reinterpret_cast<uint64_t*>(smem + (size_t)8 * k * KNN_BLOCK);
  struct { int cnt; } col;
  col.cnt = knn_search<KCAP>(tv, p.x, p.y, p.z, k, kd + threadIdx.x, ki + threadIdx.x, lds_stack + threadIdx.x);
  double mean[3] = {0, 0, 0}, c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
  for (int e = 0; e < k; e++) {  // neighbours in ascending (d2, id) order, like the search returns them
    float4 t = xyz[ki[e * KNN_BLOCK + threadIdx.x]];
    double x = t.x, y = t.y, z = t.z;
    mean[0] += x; mean[1] += y; mean[2] += z;
    c00 += x * x;
    c10 += y * x; c11 += y * y;
    c20 += z * x; c21 += z * y; c22 += z * z;
  }
  double kk = (double)k;
  mean[0] /= kk; mean[1] /= kk; mean[2] /= kk;
  double cov[9];
  cov[0] = c00 / kk - mean[0] * mean[0];
  cov[3] = c10 / kk - mean[1] * mean[0];
  cov[4] = c11 / kk - mean[1] * mean[1];
  cov[6] = c20 / kk - mean[2] * mean[0];
  cov[7] = c21 / kk - mean[2] * mean[1];
  cov[8] = c22 / kk - mean[2] * mean[2];
  cov[1] = cov[3]; cov[2] = cov[6]; cov[5] = cov[7];
  double u[3];
  smallest_sv_vector3(cov, u);
  double s = 1.0 - eps;
  cov6[(size_t)0 * n_pad + i] = 1.0 - s * u[0] * u[0];
  cov6[(size_t)1 * n_pad + i] = 0.0 - s * u[0] * u[1];
  cov6[(size_t)2 * n_pad + i] = 0.0 - s * u[0] * u[2];
  cov6[(size_t)3 * n_pad + i] = 1.0 - s * u[1] * u[1];
  cov6[(size_t)4 * n_pad + i] = 0.0 - s * u[1] * u[2];
  cov6[(size_t)5 * n_pad + i] = 1.0 - s * u[2] * u[2];
}
void launch_knn_cov(const float4* xyz, int n, int n_pad, TreeView tree, int k, double eps, double* cov6, hipStream_t s) {
  size_t sh = (size_t)k * KNN_BLOCK * 8 + stack_lds_bytes(0, KNN_BLOCK);
  LH_KNN_DISPATCH(k_knn_cov, k, dim3((n + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), sh, s, xyz, n, n_pad, tree, k, eps, cov6);
}

// pcl::eigen33 smallest eigenpair, float closed form (PCL 1.10 common/eigen.hpp restated)
__device__ void roots2f(float b, float c, float* r) {
  r[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  float sd = sqrtf(d);
  r[2] = 0.5f * (b + sd);
  r[1] = 0.5f * (b - sd);
}
__device__ void roots3f(float m00, float m01, float m02, float m11, float m12, float m22, float* r) {
  float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
  float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
  float c2 = m00 + m11 + m22;
  if (fabsf(c0) < 1.1920929e-07f) {
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

// solvePlaneParameters + flipNormalTowardsViewpoint on the nine raw moment sums of `cnt` neighbours (PCL 1.10)
__device__ float4 normal_from_moments(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8, int cnt,
                                      float4 p) {
  float c = (float)cnt;
  a0 /= c; a1 /= c; a2 /= c; a3 /= c; a4 /= c; a5 /= c; a6 /= c; a7 /= c; a8 /= c;
  float m00 = a0 - a6 * a6, m01 = a1 - a6 * a7, m02 = a2 - a6 * a8, m11 = a3 - a7 * a7, m12 = a4 - a7 * a8, m22 = a5 - a8 * a8;
  // pcl::eigen33(mat, eigenvalue, eigenvector)
  float scale = fmaxf(fmaxf(fmaxf(fabsf(m00), fabsf(m01)), fmaxf(fabsf(m02), fabsf(m11))), fmaxf(fabsf(m12), fabsf(m22)));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float s00 = m00 / scale, s01 = m01 / scale, s02 = m02 / scale, s11 = m11 / scale, s12 = m12 / scale, s22 = m22 / scale;
  float r[3];
  roots3f(s00, s01, s02, s11, s12, s22, r);
  float ev = r[0] * scale;
  s00 -= r[0]; s11 -= r[0]; s22 -= r[0];
  // rows: r0 = (s00,s01,s02) r1 = (s01,s11,s12) r2 = (s02,s12,s22)
  float v1x = s01 * s12 - s02 * s11, v1y = s02 * s01 - s00 * s12, v1z = s00 * s11 - s01 * s01;  // r0 x r1
  float v2x = s01 * s22 - s02 * s12, v2y = s02 * s02 - s00 * s22, v2z = s00 * s12 - s01 * s02;  // r0 x r2
  float v3x = s11 * s22 - s12 * s12, v3y = s12 * s02 - s01 * s22, v3z = s01 * s12 - s11 * s02;  // r1 x r2
  float l1 = (v1x * v1x + v1y * v1y) + v1z * v1z;
  float l2 = (v2x * v2x + v2y * v2y) + v2z * v2z;
  float l3 = (v3x * v3x + v3y * v3y) + v3z * v3z;
  float nx, ny, nz, l;
  if (l1 >= l2 && l1 >= l3) { nx = v1x; ny = v1y; nz = v1z; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { nx = v2x; ny = v2y; nz = v2z; l = l2; }
  else { nx = v3x; ny = v3y; nz = v3z; l = l3; }
  float sl = sqrtf(l);
  nx /= sl; ny /= sl; nz /= sl;
  float eig_sum = m00 + m11 + m22;
  float curv = (eig_sum != 0.0f) ? fabsf(ev / eig_sum) : 0.0f;
  float vx = 0.0f - p.x, vy = 0.0f - p.y, vz = 0.0f - p.z;  // flipNormalTowardsViewpoint, vp = 0
  float cos_theta = (vx * nx + vy * ny) + vz * nz;
  if (cos_theta < 0) { nx = -nx; ny = -ny; nz = -nz; }
  return make_float4(nx, ny, nz, curv);
}

template <int KCAP>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_normals(const float4* __restrict__ xyz, int n, TreeView tv, int k,
                                                           float4* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* kd = reinterpret_cast<float*>(smem);
  int* ki = reinterpret_cast<int*>(smem + sizeof(float) * (size_t)k * KNN_BLOCK);
  int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  if (i >= n) return;
  float4 p = xyz[i];
  uint64_t* lds_stack = reinterpret_cast<uint64_t*>(smem + (size_t)8 * k * KNN_BLOCK);
  struct { int cnt; } col;
  col.cnt = knn_search<KCAP>(tv, p.x, p.y, p.z, k, kd + threadIdx.x, ki + threadIdx.x, lds_stack + threadIdx.x);
  const float qnan = __uint_as_float(0x7fc00000u);
  if (col.cnt < 3) {
    out[i] = make_float4(qnan, qnan, qnan, qnan);
    return;
  }
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
  for (int e = 0; e < col.cnt; e++) {  // computeMeanAndCovarianceMatrix, float accumulators (PCL 1.10)
    float4 t = xyz[ki[e * KNN_BLOCK + threadIdx.x]];
    a0 += t.x * t.x; a1 += t.x * t.y; a2 += t.x * t.z;
    a3 += t.y * t.y; a4 += t.y * t.z; a5 += t.z * t.z;
    a6 += t.x; a7 += t.y; a8 += t.z;
  }
  out[i] = normal_from_moments(a0, a1, a2, a3, a4, a5, a6, a7, a8, col.cnt, p);
}
void launch_knn_normals(const float4* xyz, int n, TreeView tree, int k, float4* out_nrm, hipStream_t s) {
  size_t sh = (size_t)k * KNN_BLOCK * 8 + stack_lds_bytes(0, KNN_BLOCK);
  LH_KNN_DISPATCH(k_knn_normals, k, dim3((n + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), sh, s, xyz, n, tree, k, out_nrm);
}

// radius mode of the normal filter (normal_computation.cc:71-74): moments of all points with d2 < r2, < 3 neighbours -> NaN
__global__ void __launch_bounds__(KNN_BLOCK) k_radius_normals(const float4* __restrict__ xyz, int n, TreeView tv, float r2,
                                                              float4* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  if (i >= n) return;
  float4 p = xyz[i];
  uint64_t* lds_stack = reinterpret_cast<uint64_t*>(smem);
  RadiusMomentCollector col;
  col.r2 = r2; col.cnt = 0;
#pragma unroll
  for (int e = 0; e < 9; e++) col.a[e] = 0.0f;
  tree_search(tv, p.x, p.y, p.z, col, lds_stack + threadIdx.x, KNN_BLOCK);
  const float qnan = __uint_as_float(0x7fc00000u);
  if (col.cnt < 3) {
    out[i] = make_float4(qnan, qnan, qnan, qnan);
    return;
  }
  out[i] = normal_from_moments(col.a[0], col.a[1], col.a[2], col.a[3], col.a[4], col.a[5], col.a[6], col.a[7], col.a[8], col.cnt, p);
}
void launch_radius_normals(const float4* xyz, int n, TreeView tree, float radius, float4* out_nrm, hipStream_t s) {
  size_t sh = stack_lds_bytes(0, KNN_BLOCK);
  hipLaunchKernelGGL(k_radius_normals, dim3((n + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), sh, s, xyz, n, tree, radius * radius, out_nrm);
}

// ===== the block search (lh_knn_block.hpp) ===================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8), aligned(4)));   // eight consecutive coordinates (4-byte aligned: a chunk starts anywhere)
// The 8 candidates of a chunk for the lane's query: three scalar loads (x, y, z of sorted positions first .. first + 7, from the launch's
// own coordinate arrays), then the squared distances TWO CANDIDATES PER INSTRUCTION: the register pair (x_e, x_e+1) is one operand of a
// packed-f32 subtract / multiply / add -- 4 vector instructions per candidate instead of 8, each component rounded exactly like d2f's
// (dx*dx + dy*dy) + dz*dz (packed f32 arithmetic is IEEE, and the unit is compiled without contraction).
struct ChunkSoa { f32x8 x, y, z; };
__device__ __forceinline__ ChunkSoa load_chunk(const float* sx, const float* sy, const float* sz, int first) {
  ChunkSoa c;
  c.x = *reinterpret_cast<const f32x8 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(sx + first));
  c.y = *reinterpret_cast<const f32x8 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(sy + first));
  c.z = *reinterpret_cast<const f32x8 __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(sz + first));
  return c;
}
__device__ __forceinline__ void chunk_keys(const ChunkSoa& c, int cnt, f32x2 qx2, f32x2 qy2, f32x2 qz2, uint32_t* B) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 x = {c.x[e], c.x[e + 1]}, y = {c.y[e], c.y[e + 1]}, z = {c.z[e], c.z[e + 1]};
    const f32x2 dx = qx2 - x, dy = qy2 - y, dz = qz2 - z;
    const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
    B[e] = __float_as_uint(dd.x);
    B[e + 1] = __float_as_uint(dd.y);
  }
  if (cnt < 8) {   // a short chunk: the entries past its count are the NEXT leaf's points -- their keys become +INF (one v_max with a scalar mask each)
#pragma unroll
    for (int e = 1; e < 8; e++) B[e] = max(B[e], e < cnt ? 0u : KNN_KEY_INF);
  }
}
// the launch's first step: the sorted points' coordinates as three arrays (pads included: +INF)
__global__ void __launch_bounds__(256) k_knn_soa(const KnnCloudDesc* __restrict__ descs, int n_clouds, int bpc) {
  int cl, blk;
  if (!xcd_job_map(n_clouds, bpc, cl, blk)) return;
  const KnnCloudDesc d = descs[cl];
  const int i = blk * 256 + threadIdx.x;
  if (i >= d.n + LEAF_CAP) return;
  const float4 p = d.pts[i];
  d.sx[i] = p.x; d.sy[i] = p.y; d.sz[i] = p.z;
}
// a load through the CONSTANT address space: with a wave-uniform address it is an s_load (the data arrive in SGPRs and every lane's
// vector instruction takes them as a scalar operand: no LDS staging, no broadcast).  The tree and the sorted points were written by
// earlier launches and are read-only here.
template <class V>
__device__ __forceinline__ V sload(const void* p) {
  return *reinterpret_cast<const V __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// wave-wide votes straight from a comparison (HIP's __any / __ballot take an int: the compiler then materialises 0 / 1 and compares again)
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ int count_lanes(bool p) { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }

// what a finished query does with its neighbours: entries 0 .. cnt-1 in ascending (d2, index) order, id_at(j) / d2_at(j) with a
// compile-time j (the block search keeps them in registers, the redo kernel in LDS)
template <int MODE, int KMAX, class IdAt, class D2At>
__device__ __forceinline__ void knn_consume(const KnnCloudDesc& d, int qid, float qx, float qy, float qz, int cnt, int k, double eps, IdAt id_at,
                                            D2At d2_at) {
  if constexpr (MODE == KNN_MODE_RAW) {
#pragma unroll
    for (int j = 0; j < KMAX; j++)
      if (j < k) {
        const bool ok = j < cnt;
        gst(d.idx + (size_t)qid * k + j, ok ? (int32_t)id_at(j) : -1);
        gst(d.d2 + (size_t)qid * k + j, ok ? d2_at(j) : INFINITY);
      }
  } else if constexpr (MODE == KNN_MODE_NORMALS) {
    const float qnan = __uint_as_float(0x7fc00000u);
    if (cnt < 3) {
      gst(d.nrm + qid, make_float4(qnan, qnan, qnan, qnan));
      return;
    }
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
#pragma unroll
    for (int j = 0; j < KMAX; j++)  // computeMeanAndCovarianceMatrix, float accumulators, neighbours in the order the search returns them (PCL 1.10)
      if (j < cnt) {
        const float4 t = gload16<float4>(d.xyz + id_at(j));
        a0 += t.x * t.x; a1 += t.x * t.y; a2 += t.x * t.z;
        a3 += t.y * t.y; a4 += t.y * t.z; a5 += t.z * t.z;
        a6 += t.x; a7 += t.y; a8 += t.z;
      }
    gst(d.nrm + qid, normal_from_moments(a0, a1, a2, a3, a4, a5, a6, a7, a8, cnt, make_float4(qx, qy, qz, 1.0f)));
  } else {  // computeCovariances, k-NN branch (gicp.hpp:85-154): double sums over the k neighbours (k <= n is the caller's precondition)
    double mean[3] = {0, 0, 0}, c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
#pragma unroll
    for (int j = 0; j < KMAX; j++)
      if (j < k) {
        const float4 t = gload16<float4>(d.xyz + id_at(j));
        const double x = t.x, y = t.y, z = t.z;
        mean[0] += x; mean[1] += y; mean[2] += z;
        c00 += x * x;
        c10 += y * x; c11 += y * y;
        c20 += z * x; c21 += z * y; c22 += z * z;
      }
    const double kk = (double)k;
    mean[0] /= kk; mean[1] /= kk; mean[2] /= kk;
    double cov[9];
    cov[0] = c00 / kk - mean[0] * mean[0];
    cov[3] = c10 / kk - mean[1] * mean[0];
    cov[4] = c11 / kk - mean[1] * mean[1];
    cov[6] = c20 / kk - mean[2] * mean[0];
    cov[7] = c21 / kk - mean[2] * mean[1];
    cov[8] = c22 / kk - mean[2] * mean[2];
    cov[1] = cov[3]; cov[2] = cov[6]; cov[5] = cov[7];
    double u[3];
    smallest_sv_vector3(cov, u);
    const double s = 1.0 - eps;
    gst(d.cov6 + (size_t)0 * d.n_pad + qid, 1.0 - s * u[0] * u[0]);
    gst(d.cov6 + (size_t)1 * d.n_pad + qid, 0.0 - s * u[0] * u[1]);
    gst(d.cov6 + (size_t)2 * d.n_pad + qid, 0.0 - s * u[0] * u[2]);
    gst(d.cov6 + (size_t)3 * d.n_pad + qid, 1.0 - s * u[1] * u[1]);
    gst(d.cov6 + (size_t)4 * d.n_pad + qid, 0.0 - s * u[1] * u[2]);
    gst(d.cov6 + (size_t)5 * d.n_pad + qid, 1.0 - s * u[2] * u[2]);
  }
}

// A 64-entry table of wave-uniform words kept in ONE vector register: entry i lives in lane i (v_writelane / v_readlane).  The walk's
// stack (four such registers: child reference + its packed box) and the list of remembered chunks live here -- no LDS, no exec-mask
// juggling for a one-lane store, and a popped entry arrives in SCALAR registers, where the per-lane box test wants it.
// (v_writelane through inline assembly: this compiler has no builtin for it.  gfx9 allows a vector instruction ONE scalar-register
// operand, so the lane select goes through M0, as the compiler's own lowering of the intrinsic does.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (M0 on the clobber list: nothing else in this kernel uses it)
__device__ __forceinline__ uint32_t lane_put(uint32_t tab, uint32_t v, int i) {
  const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);   // (wave-uniform by construction; this pins them to scalar registers)
  const int si = __builtin_amdgcn_readfirstlane(i);
  asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(tab) : "s"(sv), "s"(si) : "m0");
  return tab;
}
#pragma clang diagnostic pop
__device__ __forceinline__ uint32_t lane_get(uint32_t tab, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)tab, i); }

template <int K, int MODE>
__device__ __forceinline__ void knn_block_body(const KnnCloudDesc* __restrict__ descs, int n_clouds, int bpc, int k, double eps,
                                               uint32_t* __restrict__ redo_cnt, uint2* __restrict__ redo) {
  int cl, blk;
  if (!xcd_job_map(n_clouds, bpc, cl, blk)) return;
  const KnnCloudDesc d = descs[cl];
  const int b0 = blk * KNN_BLOCK_Q;
  if (b0 >= d.n) return;
  __shared__ uint32_t table[(K + 1) * KNN_BLOCK_Q];   // rows 0 .. K-1: a lane's candidates within its k-th distance; row K: where the others go
  const int lane = threadIdx.x;
  const int n = d.n;
  const int my_pos = min(b0 + lane, n - 1);   // (the lanes past the cloud's end repeat its last point: they want nothing the others do not)
  const float4 qp = gload16<float4>(d.pts + my_pos);
  const float qx = qp.x, qy = qp.y, qz = qp.z;
  const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
  TreeHeader h;
  h.root = gld(&d.hdr->root);
  h.org[0] = gld(&d.hdr->org[0]); h.org[1] = gld(&d.hdr->org[1]); h.org[2] = gld(&d.hdr->org[2]);
  h.inv = gld(&d.hdr->inv); h.scl2 = gld(&d.hdr->scl2);
  const GridQuery gq = grid_query(h, qx, qy, qz);
  const float scl2 = h.scl2;
  uint32_t L[K];
  knn_list_init<K>(L, k);
  uint32_t tau = L[K - 1];   // the lane's bound: its k-th smallest key so far (lh_knn_block.hpp: phantom keys make that L[K - 1] for every k <= K)
  const int w0 = max(0, b0 - KNN_WIN_SIDE), w1 = min(n, b0 + KNN_BLOCK_Q + KNN_WIN_SIDE);
  uint32_t st_ref = 0, st_lo = 0, st_hi = 0, st_z = 0;   // the walk's stack (lane tables)
  uint32_t acc0 = 0, acc1 = 0, acc2 = 0;                  // the remembered chunks (3 x 64)
  int n_acc = 0, sp = 0, fail = 0, wi = 0;               // wave-uniform
  if (h.root >= 0) {   // the root's entry: a box at distance zero from everything
    st_ref = lane_put(st_ref, (uint32_t)h.root, 0); st_lo = lane_put(st_lo, 0u, 0); st_hi = lane_put(st_hi, 0xffffffffu, 0); st_z = lane_put(st_z, 0u, 0);
    sp = 1;
  }
  // ---- pass 1: the window's chunks (the block's own first, then outwards), then the wave's walk; ONE place where a chunk is merged ----
  for (;;) {
    int first = 0, cnt = 0;
    if (wi < 8 + 2 * (KNN_WIN_SIDE / 8)) {
      if (wi < 8) first = b0 + 8 * wi;
      else {
        const int s = (wi - 8) >> 1;
        first = ((wi - 8) & 1) ? b0 + KNN_BLOCK_Q + 8 * s : b0 - 8 * (s + 1);
      }
      cnt = first < 0 ? 0 : min(8, w1 - first);
      wi++;
      if (cnt <= 0) continue;
    } else {
      bool have = false;
      while (sp > 0 && !fail) {
        sp = uni(sp - 1);
        const int32_t ref = (int32_t)lane_get(st_ref, sp);
        const uint32_t e_lo = lane_get(st_lo, sp), e_hi = lane_get(st_hi, sp), e_z = lane_get(st_z, sp);
        const float bd = boxd2_q(gq, e_lo, e_hi, e_z, scl2);
        if (!any_lane(__float_as_uint(bd) <= tau)) continue;   // the bounds have tightened since the entry was pushed
        if (ref < 0) {
          const uint32_t u = (uint32_t)~ref;
          if (knn_clip_chunk((int)(u >> 4), (int)(u & 15u) + 1, w0, w1, first, cnt)) { have = true; break; }
          continue;
        }
        // an internal node: a child is stacked if ANY lane's ball reaches its box; the child most lanes want goes on top
        const u32x16 nd = sload<u32x16>(d.nodes + ref);
        uint32_t want[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float bc = boxd2_q(gq, nd[c], nd[4 + c], nd[8 + c], scl2);
          const bool w = (int32_t)nd[12 + c] != NO_CHILD && __float_as_uint(bc) <= tau;
          want[c] = (uint32_t)count_lanes(w);
        }
        int top = 0;
        uint32_t best = want[0];
#pragma unroll
        for (int c = 1; c < 4; c++)
          if (want[c] > best) { best = want[c]; top = c; }
        if (sp + 4 > KNN_BLOCK_Q) { fail |= KNN_FAIL_STACK; break; }
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (want[c] != 0u && c != top) {
            st_ref = lane_put(st_ref, nd[12 + c], sp); st_lo = lane_put(st_lo, nd[c], sp); st_hi = lane_put(st_hi, nd[4 + c], sp); st_z = lane_put(st_z, nd[8 + c], sp);
            sp++;
          }
        if (best != 0u) {
          const uint32_t t_ref = top == 0 ? nd[12] : (top == 1 ? nd[13] : (top == 2 ? nd[14] : nd[15]));
          const uint32_t t_lo = top == 0 ? nd[0] : (top == 1 ? nd[1] : (top == 2 ? nd[2] : nd[3]));
          const uint32_t t_hi = top == 0 ? nd[4] : (top == 1 ? nd[5] : (top == 2 ? nd[6] : nd[7]));
          const uint32_t t_z = top == 0 ? nd[8] : (top == 1 ? nd[9] : (top == 2 ? nd[10] : nd[11]));
          st_ref = lane_put(st_ref, t_ref, sp); st_lo = lane_put(st_lo, t_lo, sp); st_hi = lane_put(st_hi, t_hi, sp); st_z = lane_put(st_z, t_z, sp);
          sp++;
        }
      }
      if (!have) break;
    }
    // merge the chunk [first, first + cnt): 8 keys per lane from a uniform (scalar) read of the sorted points
    first = uni(first);
    cnt = uni(cnt);
    uint32_t B[8];
    chunk_keys(load_chunk(d.sx, d.sy, d.sz, first), cnt, qx2, qy2, qz2, B);
    const uint32_t m = knn_min8(B);
    if (any_lane(m <= tau)) {   // pass 2 must see this chunk again
      const uint32_t r = knn_chunk_ref((uint32_t)first, cnt);
      if (n_acc < 64) acc0 = lane_put(acc0, r, n_acc);
      else if (n_acc < 128) acc1 = lane_put(acc1, r, n_acc - 64);
      else if (n_acc < KNN_ACC_CAP) acc2 = lane_put(acc2, r, n_acc - 128);
      else fail |= KNN_FAIL_CHUNKS;
      n_acc = uni(min(n_acc + 1, KNN_ACC_CAP));
    }
    if (any_lane(m < tau)) {
      knn_sort8(B);
      KnnNet<K>::merge(L, B);
      tau = L[K - 1];
    }
  }
  // ---- pass 2: every candidate with d <= tau leaves its sorted position in the lane's column of the table ----
  int lane_fail = fail;
  if (tau == KNN_KEY_INF && n >= k) lane_fail |= KNN_FAIL_INF;
  const uint32_t tau_fin = min(tau, 0x7f7fffffu);   // (never a masked entry; with fewer than k points in the cloud: every real one)
  int cnt = 0;
  for (int a = 0; a < n_acc; a++) {
    const uint32_t r = a < 64 ? lane_get(acc0, a) : (a < 128 ? lane_get(acc1, a - 64) : lane_get(acc2, a - 128));
    const int first = (int)(r >> 4), c8 = (int)(r & 15u) + 1;
    uint32_t B[8];
    chunk_keys(load_chunk(d.sx, d.sy, d.sz, first), c8, qx2, qy2, qz2, B);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const bool in = B[e] <= tau_fin;
      table[(in ? min(cnt, K) : K) * KNN_BLOCK_Q + lane] = (uint32_t)(first + e);   // (row K: not a neighbour, or one too many)
      cnt += in ? 1 : 0;
    }
  }
  const bool ties = !lane_fail && (cnt > min(k, K) || (n >= k && cnt != k));   // more than k candidates within the k-th distance
  const bool live = b0 + lane < n;
  const int qid = (int)__float_as_uint(qp.w);
  uint64_t keys[K];
  if (any_lane(ties)) {
    // Ties at the k-th distance (two points at bit-identical float distances: about one query per 100 k-point lidar scan; everywhere in
    // lattices and clouds with repeated points): the wave settles them itself -- the lanes concerned run a (d2, index) insertion list
    // over the remembered chunks, which hold every candidate within their k-th distance.  The other lanes idle through it.
    KnnRegCollector<K> col;
    col.init(k);
    for (int a = 0; a < n_acc; a++) {
      const uint32_t r = a < 64 ? lane_get(acc0, a) : (a < 128 ? lane_get(acc1, a - 64) : lane_get(acc2, a - 128));
      const int first = (int)(r >> 4), c8 = (int)(r & 15u) + 1;
#pragma unroll 1
      for (int e = 0; e < c8; e++) {
        const f32x4 c = sload<f32x4>(d.pts + first + e);
        col.offer(ties ? d2f(qx, qy, qz, c.x, c.y, c.z) : INFINITY, ties ? (int)__float_as_uint(c.w) : 0x7fffffff);
      }
    }
    if (ties) {
      cnt = 0;
#pragma unroll
      for (int j = 0; j < K; j++) {
        keys[j] = col.id[j] != 0x7fffffff ? ((uint64_t)__float_as_uint(col.d[j]) << 32) | (uint64_t)(uint32_t)col.id[j] : ~0ull;
        cnt += col.id[j] != 0x7fffffff ? 1 : 0;
      }
    }
  }
  if (lane_fail) {
    if (live) {
      const uint32_t slot = atomicAdd(redo_cnt, 1u);
      redo[slot] = make_uint2((uint32_t)cl, (uint32_t)qid);
    }
    return;
  }
  if (!live) return;
  // ---- the k (d2, index) pairs in ascending order: what nearestKSearch returns ----
  if (!ties) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      const bool ok = j < cnt;
      const uint32_t pos = ok ? table[j * KNN_BLOCK_Q + lane] : (uint32_t)my_pos;
      const float4 p = gload16<float4>(d.pts + pos);
      keys[j] = ok ? ((uint64_t)__float_as_uint(d2f(qx, qy, qz, p.x, p.y, p.z)) << 32) | (uint64_t)__float_as_uint(p.w) : ~0ull;
    }
    KnnNet<K>::sort_pairs(keys);
  }
  knn_consume<MODE, K>(d, qid, qx, qy, qz, cnt, k, eps, [&](int j) { return (uint32_t)keys[j]; },
                       [&](int j) { return __uint_as_float((uint32_t)(keys[j] >> 32)); });
}

// The kernels: one per list size, because the waves per SIMD the register allocation aims at differ (the rarely taken tie path's insertion
// list is what gets squeezed).  Measured, 32 x 100 k points: 20 keys -- 4 waves 53.8 us per cloud, 5 waves 50.3, 6 waves 48.4; 32 keys --
// 112 / 81 / 112 (six waves spill there).
#define LH_KNN_BLOCK_KERNEL(NAME, KK, WAVES)                                                                                                  \
  template <int MODE>                                                                                                                         \
  __global__ void __launch_bounds__(KNN_BLOCK_Q) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))                                               \
  NAME(const KnnCloudDesc* __restrict__ descs, int n_clouds, int bpc, int k, double eps, uint32_t* __restrict__ redo_cnt, uint2* __restrict__ redo) { \
    knn_block_body<KK, MODE>(descs, n_clouds, bpc, k, eps, redo_cnt, redo);                                                                  \
  }
LH_KNN_BLOCK_KERNEL(k_knn_block8, 8, 6)
LH_KNN_BLOCK_KERNEL(k_knn_block20, 20, 6)
LH_KNN_BLOCK_KERNEL(k_knn_block32, 32, 5)
#undef LH_KNN_BLOCK_KERNEL

// the redo list's engine: one query per lane, the register-list search of k_knn_*, the same consumer
template <int K, int MODE>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_redo(const KnnCloudDesc* __restrict__ descs, const uint32_t* __restrict__ redo_cnt,
                                                        const uint2* __restrict__ redo, int k, double eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* kd = reinterpret_cast<float*>(smem) + threadIdx.x;
  int* ki = reinterpret_cast<int*>(smem + sizeof(float) * (size_t)K * KNN_BLOCK) + threadIdx.x;
  uint64_t* lds_stack = reinterpret_cast<uint64_t*>(smem + (size_t)8 * K * KNN_BLOCK) + threadIdx.x;
  const uint32_t total = *redo_cnt;
  for (uint32_t i = blockIdx.x * KNN_BLOCK + threadIdx.x; i < total; i += gridDim.x * KNN_BLOCK) {
    const uint2 r = redo[i];
    const KnnCloudDesc d = descs[r.x];
    const float4 p = d.xyz[r.y];
    const TreeView tv{d.pts, d.nodes, d.hdr, d.n};
    const int cnt = knn_search_regs<K>(tv, p.x, p.y, p.z, k, kd, ki, lds_stack);
    knn_consume<MODE, K>(d, (int)r.y, p.x, p.y, p.z, cnt, k, eps, [&](int j) { return (uint32_t)ki[j * KNN_BLOCK]; },
                         [&](int j) { return kd[j * KNN_BLOCK]; });
  }
}

template <int K, int MODE>
static void launch_knn_block_t(const KnnCloudDesc* descs, int n_clouds, int bpc, int k, double eps, uint32_t* redo_cnt, uint2* redo, hipStream_t s) {
  const dim3 grid(xcd_grid(n_clouds, bpc)), wg(KNN_BLOCK_Q);
  if constexpr (K == 8) hipLaunchKernelGGL(k_knn_block8<MODE>, grid, wg, 0, s, descs, n_clouds, bpc, k, eps, redo_cnt, redo);
  else if constexpr (K == 20) hipLaunchKernelGGL(k_knn_block20<MODE>, grid, wg, 0, s, descs, n_clouds, bpc, k, eps, redo_cnt, redo);
  else hipLaunchKernelGGL(k_knn_block32<MODE>, grid, wg, 0, s, descs, n_clouds, bpc, k, eps, redo_cnt, redo);
  const size_t sh = (size_t)K * KNN_BLOCK * 8 + stack_lds_bytes(0, KNN_BLOCK);
  hipLaunchKernelGGL((k_knn_redo<K, MODE>), dim3(128), dim3(KNN_BLOCK), sh, s, descs, redo_cnt, redo, k, eps);
}
template <int MODE>
static void launch_knn_block_m(const KnnCloudDesc* descs, int n_clouds, int bpc, int k, double eps, uint32_t* redo_cnt, uint2* redo, hipStream_t s) {
  if (k <= 8) launch_knn_block_t<8, MODE>(descs, n_clouds, bpc, k, eps, redo_cnt, redo, s);
  else if (k <= 20) launch_knn_block_t<20, MODE>(descs, n_clouds, bpc, k, eps, redo_cnt, redo, s);
  else launch_knn_block_t<32, MODE>(descs, n_clouds, bpc, k, eps, redo_cnt, redo, s);
}
void launch_knn_block(const KnnCloudDesc* descs_dev, int n_clouds, int max_n, int k, int mode, double eps, uint32_t* redo_cnt, uint2* redo,
                      hipStream_t s) {
  const int bpc = (max_n + KNN_BLOCK_Q - 1) / KNN_BLOCK_Q;
  (void)hipMemsetAsync(redo_cnt, 0, sizeof(uint32_t), s);
  {
    const int bps = (max_n + LEAF_CAP + 255) / 256;
    hipLaunchKernelGGL(k_knn_soa, dim3(xcd_grid(n_clouds, bps)), dim3(256), 0, s, descs_dev, n_clouds, bps);
  }
  if (mode == KNN_MODE_NORMALS) launch_knn_block_m<KNN_MODE_NORMALS>(descs_dev, n_clouds, bpc, k, eps, redo_cnt, redo, s);
  else if (mode == KNN_MODE_COV) launch_knn_block_m<KNN_MODE_COV>(descs_dev, n_clouds, bpc, k, eps, redo_cnt, redo, s);
  else launch_knn_block_m<KNN_MODE_RAW>(descs_dev, n_clouds, bpc, k, eps, redo_cnt, redo, s);
}

}  // namespace lh
