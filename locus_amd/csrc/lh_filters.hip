// lh_filters.hip -- C ABI (include/locus_hip.h), second half: K8 point-to-plane information + ICP covariance, K3 normal filters, K1 voxel
// grids, NDT (registration_method ndt), body filter, local map, profiling.  Runtime types: lh_runtime.hpp.
#include "lh_runtime.hpp"

// ---- K8: normalizePCloud's reductions (utils.cc:106-128) -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_centroid_partials(const float4* __restrict__ xyz, int n, double* __restrict__ part) {
  // per-block sums of x, y, z over finite points + count (pcl::compute3DCentroid), fixed reduction shape
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int base = blockIdx.x * 1024;
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n) {
      float4 p = xyz[i];
      if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) { a0 += p.x; a1 += p.y; a2 += p.z; a3 += 1.0; }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64); a1 += __shfl_down(a1, off, 64); a2 += __shfl_down(a2, off, 64); a3 += __shfl_down(a3, off, 64);
  }
  __shared__ double sm[4][4];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = a0; sm[threadIdx.x >> 6][1] = a1; sm[threadIdx.x >> 6][2] = a2; sm[threadIdx.x >> 6][3] = a3; }
  __syncthreads();
  if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}
__global__ void __launch_bounds__(256) k_dist_partials(const float4* __restrict__ xyz, int n, float cx, float cy, float cz,
                                                      double* __restrict__ part) {
  double a = 0;
  int base = blockIdx.x * 1024;
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n) {
      float4 p = xyz[i];
      float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
      a += (double)sqrtf((dx * dx + dy * dy) + dz * dz);  // utils.cc:118
    }
  }
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// ---- K8 without the host in the middle (lh_gicp_measurement_update): normalizePCloud's two reductions end ON the device, the information
// matrix is summed from the un-normalised query with the normalisation applied in flight, the 21 sums leave in one copy.  Every value is
// formed exactly as lh_p2plane_information forms it (same partial sums, same order of the final sums, same float expressions), so the
// two entry points return the same bits.
// the per-block bodies, shared by the multi-block kernels and by the one-launch form for small clouds (k_p2plane_small): same thread -> point map,
// same shuffle tree, same order of the four waves -- a block's partial sums are the same bits whichever kernel forms them
__device__ __forceinline__ void centroid_block(const float4* __restrict__ xyz, int n, int blk, double (*sm)[4], double* __restrict__ out4) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int base = blk * 1024;
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n) {
      float4 p = xyz[i];
      if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) { a0 += p.x; a1 += p.y; a2 += p.z; a3 += 1.0; }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64); a1 += __shfl_down(a1, off, 64); a2 += __shfl_down(a2, off, 64); a3 += __shfl_down(a3, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = a0; sm[threadIdx.x >> 6][1] = a1; sm[threadIdx.x >> 6][2] = a2; sm[threadIdx.x >> 6][3] = a3; }
  __syncthreads();
  if (threadIdx.x < 4) out4[threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
  __syncthreads();
}
__device__ __forceinline__ void dist_block(const float4* __restrict__ xyz, int n, int blk, float cx, float cy, float cz, double* sm4, double* __restrict__ out1) {
  double a = 0;
  int base = blk * 1024;
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n) {
      float4 p = xyz[i];
      float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
      a += (double)sqrtf((dx * dx + dy * dy) + dz * dz);  // utils.cc:118
    }
  }
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out1[0] = ((sm4[0] + sm4[1]) + sm4[2]) + sm4[3];
  __syncthreads();
}
// Ap = sum H^T H, H = [a x n, n] (PointCloudLocalization.cc:723-750) with a = factor * (p - centroid) formed in flight (the expression of
// launch_transform with T12 = {f, 0, 0, -f cx, ...}); corr < 0 (a query without a neighbour: a non-finite point) is skipped like a NaN
__device__ __forceinline__ void ap_block(const float4* __restrict__ xyz, int n, int blk, const float* cf, const float4* __restrict__ ref_nrm,
                                         const int32_t* __restrict__ corr, double (*sm)[21], double* __restrict__ out21) {
  const float f = cf[3];
  const float T12[12] = {f, 0, 0, -f * cf[0], 0, f, 0, -f * cf[1], 0, 0, f, -f * cf[2]};
  double acc[21];
#pragma unroll
  for (int k = 0; k < 21; k++) acc[k] = 0.0;
  int base = blk * 1024;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int i = base + r * 256 + threadIdx.x;
    if (i < n) {
      const int32_t j = corr[i];
      if (j >= 0) {
        float4 p = xyz[i];
        float ax, ay, az;
        lh::xform_pt(T12, p.x, p.y, p.z, ax, ay, az);
        float4 n4 = ref_nrm[j];
        double a0 = ax, a1 = ay, a2 = az, n0 = n4.x, n1 = n4.y, n2 = n4.z;
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
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 21; k++) acc[k] += __shfl_down(acc[k], off, 64);
  }
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 21; k++) sm[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 21) out21[threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
  __syncthreads();
}
__global__ void __launch_bounds__(256) k_centroid_partials_dev(const float4* __restrict__ xyz, int n, double* __restrict__ part) {
  __shared__ double sm[4][4];
  centroid_block(xyz, n, blockIdx.x, sm, part + blockIdx.x * 4);
}
__global__ void __launch_bounds__(64) k_centroid_final(const double* __restrict__ part, int nb, float* __restrict__ cf) {
  if (threadIdx.x != 0) return;
  double sx = 0, sy = 0, sz = 0, cnt = 0;
  for (int b = 0; b < nb; b++) { sx += part[b * 4]; sy += part[b * 4 + 1]; sz += part[b * 4 + 2]; cnt += part[b * 4 + 3]; }
  cf[0] = (float)(sx / cnt); cf[1] = (float)(sy / cnt); cf[2] = (float)(sz / cnt);
}
__global__ void __launch_bounds__(256) k_dist_partials_dev(const float4* __restrict__ xyz, int n, const float* __restrict__ cf, double* __restrict__ part) {
  __shared__ double sm[4];
  dist_block(xyz, n, blockIdx.x, cf[0], cf[1], cf[2], sm, part + blockIdx.x);
}
__global__ void __launch_bounds__(64) k_factor_final(const double* __restrict__ part, int nb, int n, float* __restrict__ cf) {
  if (threadIdx.x != 0) return;
  double dist = 0;
  for (int b = 0; b < nb; b++) dist += part[b];
  cf[3] = (float)n / (float)dist;  // utils.cc:120
}
__global__ void __launch_bounds__(256) k_ap_norm(const float4* __restrict__ xyz, int n, const float* __restrict__ cf, const float4* __restrict__ ref_nrm,
                                                 const int32_t* __restrict__ corr, double* __restrict__ partials) {
  __shared__ double sm[4][21];
  ap_block(xyz, n, blockIdx.x, cf, ref_nrm, corr, sm, partials + blockIdx.x * 21);
}
__global__ void __launch_bounds__(64) k_ap_final(const double* __restrict__ partials, int nb, double* __restrict__ out21) {
  if (threadIdx.x >= 21) return;
  double s = 0;
  for (int b = 0; b < nb; b++) s += partials[(size_t)b * 21 + threadIdx.x];
  out21[threadIdx.x] = s;
}
// The whole chain in ONE launch of one workgroup for the clouds LOCUS registers (a few thousand points: six dependent launches of 4-5 us each
// for microseconds of work): the blocks one after the other through the same per-block bodies, the final sums by one thread in block order.
constexpr int P2P_SMALL_BLOCKS = 16;   // <= 16 384 points
__global__ void __launch_bounds__(256) k_p2plane_small(const float4* __restrict__ xyz, int n, const float4* __restrict__ ref_nrm, const int32_t* __restrict__ corr,
                                                       double* __restrict__ out21) {
  __shared__ double sm4[4][4], smd[4], sma[4][21];
  __shared__ double part[P2P_SMALL_BLOCKS * 21];
  __shared__ float cf[4];
  const int nb = (n + 1023) / 1024;
  for (int b = 0; b < nb; b++) centroid_block(xyz, n, b, sm4, part + b * 4);
  if (threadIdx.x == 0) {
    double sx = 0, sy = 0, sz = 0, cnt = 0;
    for (int b = 0; b < nb; b++) { sx += part[b * 4]; sy += part[b * 4 + 1]; sz += part[b * 4 + 2]; cnt += part[b * 4 + 3]; }
    cf[0] = (float)(sx / cnt); cf[1] = (float)(sy / cnt); cf[2] = (float)(sz / cnt);
  }
  __syncthreads();
  for (int b = 0; b < nb; b++) dist_block(xyz, n, b, cf[0], cf[1], cf[2], smd, part + b);
  if (threadIdx.x == 0) {
    double dist = 0;
    for (int b = 0; b < nb; b++) dist += part[b];
    cf[3] = (float)n / (float)dist;  // utils.cc:120
  }
  __syncthreads();
  for (int b = 0; b < nb; b++) ap_block(xyz, n, b, cf, ref_nrm, corr, sma, part + b * 21);
  if (threadIdx.x < 21) {
    double s = 0;
    for (int b = 0; b < nb; b++) s += part[b * 21 + threadIdx.x];
    out21[threadIdx.x] = s;
  }
}
// The whole chain, enqueued on s: out21 (device) receives the 21 unique entries of Ap.  scratch: (21 nb + 4) doubles of device memory.
lh_status p2plane_information_device(const float4* qxyz, int n, const float4* ref_nrm, const int32_t* corr, double* scratch, double* out21, hipStream_t s) {
  const int nb = lh::sum_blocks(n);
  static const bool small_on = []() { const char* e = getenv("LH_P2P_SMALL"); return e ? atoi(e) != 0 : true; }();   // (0: the six-launch chain for every size: A/B, tests)
  if (small_on && nb <= P2P_SMALL_BLOCKS) {
    hipLaunchKernelGGL(k_p2plane_small, dim3(1), dim3(256), 0, s, qxyz, n, ref_nrm, corr, out21);
    return hipGetLastError() == hipSuccess ? LH_OK : LH_EDEVICE;
  }
  float* cf = reinterpret_cast<float*>(scratch + (size_t)nb * 21);
  hipLaunchKernelGGL(k_centroid_partials_dev, dim3(nb), dim3(256), 0, s, qxyz, n, scratch);
  hipLaunchKernelGGL(k_centroid_final, dim3(1), dim3(64), 0, s, scratch, nb, cf);
  hipLaunchKernelGGL(k_dist_partials_dev, dim3(nb), dim3(256), 0, s, qxyz, n, cf, scratch);
  hipLaunchKernelGGL(k_factor_final, dim3(1), dim3(64), 0, s, scratch, nb, n, cf);
  hipLaunchKernelGGL(k_ap_norm, dim3(nb), dim3(256), 0, s, qxyz, n, cf, ref_nrm, corr, scratch);
  hipLaunchKernelGGL(k_ap_final, dim3(1), dim3(64), 0, s, scratch, nb, out21);
  return hipGetLastError() == hipSuccess ? LH_OK : LH_EDEVICE;
}

#pragma GCC visibility push(default)   // the C ABI is the library's ONLY exported surface (the TUs are compiled -fvisibility=hidden)
extern "C" {

// ---- K8 / H2 -----------------------------------------------------------------------------------------------
lh_status lh_p2plane_information(lh_ctx* c, const lh_cloud* query, const lh_cloud* reference, const int64_t* corr, double Ap[36]) {
  if (!c || !query || !reference || !corr || !Ap || !reference->nrm) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->device));
  int n = query->n;
  for (int i = 0; i < n; i++)
    if (corr[i] < 0 || corr[i] >= reference->n) return LH_EINVAL;
  int nb = sum_blocks(n);
  lh_status st = ctx_ensure_small(c, (size_t)nb * 21);
  if (st) return st;
  double* d_part = nullptr;
  float4* d_qn = nullptr;
  int64_t* d_corr = nullptr;
  DevGuard guard;
  HIPCHK(guard.alloc(&d_part, sizeof(double) * (size_t)nb * 21));
  HIPCHK(guard.alloc(&d_qn, sizeof(float4) * (size_t)n));
  HIPCHK(guard.alloc(&d_corr, sizeof(int64_t) * (size_t)n));
  HIPCHK(hipMemcpyAsync(d_corr, corr, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  // normalizePCloud (utils.cc:106-128): centroid, factor = N / sum |p - c|, q' = factor*(p - c).
  // The reference accumulates both sums sequentially in float; here the sums are double with a fixed tree
  // (more accurate; differences vs the float-sequential reference are O(1e-6) relative -- see DESIGN.md).
  hipLaunchKernelGGL(k_centroid_partials, dim3(nb), dim3(256), 0, c->stream, query->xyz, n, d_part);
  HIPCHK(hipMemcpyAsync(c->small_host, d_part, sizeof(double) * (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double sx = 0, sy = 0, sz = 0, cnt = 0;
  for (int b = 0; b < nb; b++) { sx += c->small_host[b * 4]; sy += c->small_host[b * 4 + 1]; sz += c->small_host[b * 4 + 2]; cnt += c->small_host[b * 4 + 3]; }
  float cx = (float)(sx / cnt), cy = (float)(sy / cnt), cz = (float)(sz / cnt);
  hipLaunchKernelGGL(k_dist_partials, dim3(nb), dim3(256), 0, c->stream, query->xyz, n, cx, cy, cz, d_part);
  HIPCHK(hipMemcpyAsync(c->small_host, d_part, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double dist = 0;
  for (int b = 0; b < nb; b++) dist += c->small_host[b];
  float factor = (float)n / (float)dist;  // utils.cc:120
  float T12[12] = {factor, 0, 0, -factor * cx, 0, factor, 0, -factor * cy, 0, 0, factor, -factor * cz};
  launch_transform(query->xyz, nullptr, n, T12, d_qn, nullptr, c->stream);
  { ProfScope p(c, "p2plane_Ap", 40.0 * n); launch_ap(d_qn, n, reference->nrm, d_corr, d_part, c->stream); }
  HIPCHK(hipMemcpyAsync(c->small_host, d_part, sizeof(double) * (size_t)nb * 21, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double U[21];
  for (int k = 0; k < 21; k++) U[k] = 0;
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 21; k++) U[k] += c->small_host[(size_t)b * 21 + k];
  int t = 0;
  for (int r = 0; r < 6; r++)
    for (int cc = r; cc < 6; cc++) { Ap[r * 6 + cc] = U[t]; Ap[cc * 6 + r] = U[t]; t++; }
  return LH_OK;
}

// ComputePoint2PlaneICPCovariance conditioning (PointCloudLocalization.cc:487-538): 6x6, host-side by nature
static void sym_eig6(const double* Ain, double* ev) {  // cyclic Jacobi, eigenvalues only
  double A[36];
  memcpy(A, Ain, sizeof(A));
  const int n = 6;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
        for (int k = 0; k < n; k++) { double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = cs * akp - sn * akq; A[k * n + q] = sn * akp + cs * akq; }
        for (int k = 0; k < n; k++) { double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = cs * apk - sn * aqk; A[q * n + k] = sn * apk + cs * aqk; }
      }
  }
  for (int i = 0; i < n; i++) ev[i] = A[i * n + i];
}

lh_status lh_icp_covariance(const double Ap[36], double upper_bound, double cov[36], double* condition_number) {
  if (!Ap || !cov) return LH_EINVAL;
  const int n = 6;
  // cov = 0.05^2 * Ap^-1 (Gauss-Jordan with partial pivoting; Eigen uses PartialPivLU)
  double a[6][12];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) { a[i][j] = Ap[i * n + j]; a[i][n + j] = (i == j); }
  for (int col = 0; col < n; col++) {
    int piv = col;
    for (int r = col + 1; r < n; r++)
      if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
    if (piv != col)
      for (int j = 0; j < 2 * n; j++) std::swap(a[col][j], a[piv][j]);
    double d = a[col][col];
    for (int j = 0; j < 2 * n; j++) a[col][j] /= d;
    for (int r = 0; r < n; r++) {
      if (r == col) continue;
      double f = a[r][col];
      if (f != 0.0 || std::isnan(f))
        for (int j = 0; j < 2 * n; j++) a[r][j] -= f * a[col][j];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) cov[i * n + j] = 0.05 * 0.05 * a[i][n + j];
  // Eigen LDLT (lower, diagonal pivoting); the reference recomposes L*D*L^T without the permutation (:518)
  double M[36];
  memcpy(M, cov, sizeof(M));
  for (int k = 0; k < n; k++) {
    int big = k;
    double bv = fabs(M[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (fabs(M[i * n + i]) > bv) { bv = fabs(M[i * n + i]); big = i; }
    if (big != k) {
      int s = n - big - 1;
      for (int j = 0; j < k; j++) std::swap(M[k * n + j], M[big * n + j]);
      for (int i = 0; i < s; i++) std::swap(M[(big + 1 + i) * n + k], M[(big + 1 + i) * n + big]);
      std::swap(M[k * n + k], M[big * n + big]);
      for (int i = k + 1; i < big; i++) std::swap(M[i * n + k], M[big * n + i]);
    }
    int rs = n - k - 1;
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; j++) temp[j] = M[j * n + j] * M[k * n + j];
      double s = 0;
      for (int j = 0; j < k; j++) s += M[k * n + j] * temp[j];
      M[k * n + k] -= s;
      for (int i = 0; i < rs; i++) {
        double tt = 0;
        for (int j = 0; j < k; j++) tt += M[(k + 1 + i) * n + j] * temp[j];
        M[(k + 1 + i) * n + k] -= tt;
      }
    }
    double akk = M[k * n + k];
    bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) break;
    if (rs > 0 && valid)
      for (int i = 0; i < rs; i++) M[(k + 1 + i) * n + k] /= akk;
  }
  double L[36], D[6];
  for (int i = 0; i < n; i++) {
    D[i] = M[i * n + i];
    for (int j = 0; j < n; j++) L[i * n + j] = (i == j) ? 1.0 : (i > j ? M[i * n + j] : 0.0);
  }
  for (int i = 0; i < n; i++)
    if (std::isnan(D[i])) {  // :499-503
      for (int q = 0; q < 36; q++) cov[q] = (q % 7 == 0) ? upper_bound : 0.0;
      if (condition_number) *condition_number = 1.0;
      return LH_ESOLVER;
    }
  bool recompute = false;
  for (int i = 0; i < n; i++) {
    if (D[i] <= 0) { D[i] = 1e-12; recompute = true; }
    if (D[i] > upper_bound) { D[i] = upper_bound; recompute = true; }
  }
  if (recompute)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += L[i * n + k] * D[k] * L[j * n + k];
        cov[i * n + j] = s;
      }
  bool has_nan = false;
  for (int q = 0; q < 36; q++)
    if (std::isnan(cov[q])) has_nan = true;
  if (has_nan)
    for (int q = 0; q < 36; q++) cov[q] = (q % 7 == 0) ? upper_bound : 0.0;
  if (condition_number) {
    double sym[36], ev[6];
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) sym[i * n + j] = 0.5 * (cov[i * n + j] + cov[j * n + i]);
    sym_eig6(sym, ev);
    double smax = 0, smin = 1e300;
    for (int i = 0; i < n; i++) { smax = std::max(smax, fabs(ev[i])); smin = std::min(smin, fabs(ev[i])); }
    *condition_number = smax / smin;
  }
  return LH_OK;
}

// ---- K3 filter flavour ---------------------------------------------------------------------------------------
lh_status lh_normals_knn_cloud(lh_cloud* c, int k) {
  if (!c) return LH_EINVAL;
  return lh_normals_knn_batch(&c, 1, k);
}
// the NormalComputation nodelet (normal_computation.cc:26-59) for a queue of scans: one index build and one k-NN launch for all of them
lh_status lh_normals_knn_batch(lh_cloud* const* clouds, int n_clouds, int k) {
  if (!clouds || n_clouds <= 0 || !clouds[0] || k < 3 || k > 64) return LH_EINVAL;
  lh_ctx* x = clouds[0]->ctx;
  HIPCHK(hipSetDevice(x->device));
  return knn_block_batch(x, clouds, n_clouds, k, KNN_MODE_NORMALS, 0.0);
}
// computeCovariances' k-NN branch (gicp.hpp:85-154) for a queue of clouds: fills the covariances lh_gicp_* uses when
// recompute_*_cov is set (they stay with the cloud, like lh_cov_knn's)
lh_status lh_cov_knn_batch(lh_cloud* const* clouds, int n_clouds, int k, double gicp_epsilon) {
  if (!clouds || n_clouds <= 0 || !clouds[0] || k < 1 || k > 64) return LH_EINVAL;
  lh_ctx* x = clouds[0]->ctx;
  HIPCHK(hipSetDevice(x->device));
  return knn_block_batch(x, clouds, n_clouds, k, KNN_MODE_COV, gicp_epsilon);
}
// radius mode (normal_computation.cc:71-74): NaN normals where fewer than 3 neighbours lie within `radius`
lh_status lh_normals_radius_cloud(lh_cloud* c, float radius) {
  if (!c || !(radius > 0.0f)) return LH_EINVAL;
  lh_ctx* x = c->ctx;
  HIPCHK(hipSetDevice(x->device));
  if (!c->has_index) { lh_status st = cloud_build_index(c); if (st) return st; }
  if (!c->nrm) HIPCHK(lhMalloc(&c->nrm, sizeof(float4) * (size_t)c->n_pad));
  { ProfScope p(x, "radius_normals", 32.0 * c->n); launch_radius_normals(c->xyz, c->n, c->view(), radius, c->nrm, x->stream); }
  HIPCHK(hipGetLastError());
  return LH_OK;
}
lh_status lh_normals_radius(lh_ctx* ctx, const lh_cloud_view* in, float radius, float* out_normals4) {
  if (!ctx || !in || !out_normals4) return LH_EINVAL;
  lh_cloud* c = nullptr;
  lh_status st = upload_view(ctx, in, &c);
  if (st) return st;
  st = lh_normals_radius_cloud(c, radius);
  if (!st) {
    hipError_t e = hipMemcpyAsync(out_normals4, c->nrm, sizeof(float4) * (size_t)c->n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = LH_EDEVICE;
  }
  (void)hipStreamSynchronize(ctx->stream);
  cloud_free(c);
  return st;
}
// pcl::removeNaNNormalsFromPointCloud (normal_computation.cc:52-56) on the device: order-preserving compaction into a new cloud
lh_status lh_cloud_remove_nan_normals(const lh_cloud* in, lh_cloud** out) {
  if (!in || !out || !in->nrm || in->n <= 0) return LH_EINVAL;
  lh_ctx* c = in->ctx;
  HIPCHK(hipSetDevice(c->device));
  int n = in->n;
  uint32_t *d_flags = nullptr, *d_incl = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = scan_temp_bytes(n);
  HIPCHK(lhMalloc(&d_flags, sizeof(uint32_t) * (size_t)n));
  HIPCHK(lhMalloc(&d_incl, sizeof(uint32_t) * (size_t)n));
  HIPCHK(lhMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
  launch_finite_normal_flags(in->nrm, n, d_flags, c->stream);
  inclusive_scan_u32(d_tmp, tmp_bytes, d_flags, d_incl, n, c->stream);
  uint32_t total = 0;
  hipError_t e = hipMemcpyAsync(&total, d_incl + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  lh_status st = e == hipSuccess ? LH_OK : LH_EDEVICE;
  lh_cloud* o = nullptr;
  if (!st && total == 0) st = LH_EINVAL;  // nothing survives: no cloud to return
  if (!st) {
    o = new lh_cloud();
    o->ctx = c; o->n = (int)total; o->n_pad = round_up(o->n, 256);
    if (lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad) != hipSuccess || lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad) != hipSuccess ||
        (in->intensity && lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad) != hipSuccess))
      st = LH_ENOMEM;
  }
  if (!st) {
    launch_compact(d_incl, n, in->xyz, in->nrm, in->intensity, o->xyz, o->nrm, o->intensity, c->stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) st = LH_EDEVICE;
  }
  (void)lhFree(d_flags); (void)lhFree(d_incl); (void)lhFree(d_tmp);
  if (st) { cloud_free(o); return st; }
  *out = o;
  return LH_OK;
}
lh_status lh_normals_knn(lh_ctx* ctx, const lh_cloud_view* in, int k, float* out_normals4) {
  if (!ctx || !in || !out_normals4) return LH_EINVAL;
  lh_cloud* c = nullptr;
  lh_status st = upload_view(ctx, in, &c);
  if (st) return st;
  st = lh_normals_knn_cloud(c, k);
  if (!st) {
    hipError_t e = hipMemcpyAsync(out_normals4, c->nrm, sizeof(float4) * (size_t)c->n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = LH_EDEVICE;
  }
  (void)hipStreamSynchronize(ctx->stream);
  cloud_free(c);
  return st;
}

// ---- K1: CustomVoxelGrid::filter (custom_voxel_grid.cc:76-87 -> pcl::VoxelGrid::applyFilter) ------------------
static float dec_ordered_host(uint32_t e) {
  uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// device core: d_in = n x (x, y, z, intensity); on success *d_out (hipMalloc'ed, caller frees) holds *total centroids
// voxel segmentation shared by the voxel-grid filter and the NDT target grid: sorted (voxel key, point) pairs in the context's
// scratch (c->keys1 / c->vals1), segment heads and their inclusive scan; total = number of occupied voxels
struct VoxelSegments {
  uint32_t *heads = nullptr, *rank = nullptr;
  void* scan_tmp = nullptr;
  uint32_t total = 0;
  void release() { (void)lhFree(heads); (void)lhFree(rank); (void)lhFree(scan_tmp); heads = rank = nullptr; scan_tmp = nullptr; }
};
static lh_status voxel_segments(lh_ctx* c, const float4* d_in, int n, float leaf, int limit_axis, double lo, double hi, VoxelSegments* vs) {
  vs->total = 0;
  lh_status st = ctx_ensure_scratch(c, n);
  if (st) return st;
  size_t scan_bytes = scan_temp_bytes(n);
  hipError_t e = lhMalloc(&vs->heads, sizeof(uint32_t) * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&vs->rank, sizeof(uint32_t) * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&vs->scan_tmp, scan_bytes ? scan_bytes : 16);
  if (e != hipSuccess) { vs->release(); return LH_ENOMEM; }
  float flo = (float)std::max(lo, -3.0e38), fhi = (float)std::min(hi, 3.0e38);
  { ProfScope p(c, "voxel_bbox", 16.0 * n); launch_voxel_bbox(d_in, n, limit_axis, flo, fhi, c->bbox, c->stream); }
  uint32_t enc[6];
  e = hipMemcpyAsync(enc, c->bbox, sizeof(enc), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { vs->release(); return LH_EDEVICE; }
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = dec_ordered_host(enc[a]); mx[a] = dec_ordered_host(enc[3 + a]); }
  if (!(mn[0] <= mx[0])) return LH_OK;  // no point passed the filter (total = 0)
  float inv = 1.0f / leaf;
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) { vs->release(); return LH_EINVAL; }  // PCL: "Leaf size is too small ... Integer indices would overflow"
  VoxelGridDesc g;
  g.inv_leaf = inv; g.limit_axis = limit_axis; g.lo = flo; g.hi = fhi;
  int divb[3];
  for (int a = 0; a < 3; a++) {
    g.minb[a] = (int)floorf(mn[a] * inv);
    divb[a] = (int)floorf(mx[a] * inv) - g.minb[a] + 1;
  }
  g.mul[0] = 1; g.mul[1] = divb[0]; g.mul[2] = divb[0] * divb[1];
  { ProfScope p(c, "voxel_keys", 24.0 * n); launch_voxel_keys(d_in, n, g, c->keys0, c->vals0, c->stream); }
  // only as many key bits as the grid has cells: a rejected point's key is all ones, so with 2^bits > cells it still sorts behind every voxel
  int key_bits = 1;
  while (key_bits < 32 && ((int64_t)1 << key_bits) <= (int64_t)divb[0] * divb[1] * divb[2]) key_bits++;
  { ProfScope p(c, "voxel_radix_sort", 16.0 * n * ((key_bits + 9) / 10)); sort_pairs_u32(c->sort_temp, c->sort_temp_bytes, c->keys0, c->keys1, c->vals0, c->vals1, n, key_bits, c->stream); }
  { ProfScope p(c, "voxel_segments", 16.0 * n);
    launch_voxel_heads(c->keys1, n, vs->heads, c->stream);
    inclusive_scan_u32(vs->scan_tmp, scan_bytes, vs->heads, vs->rank, n, c->stream); }
  e = hipMemcpyAsync(&vs->total, vs->rank + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { vs->release(); return LH_EDEVICE; }
  return LH_OK;
}
// d_inten / d_out_inten (device clouds): d_in is a cloud's xyz array (w = 1) with the intensities beside it, and the centroids leave as a cloud's
// two arrays (n_pad entries each) -- the filter reads and writes the cloud layout itself instead of a packed copy
static lh_status voxel_grid_device(lh_ctx* c, const float4* d_in, int n, float leaf, int limit_axis, double lo, double hi,
                                   float4** d_out, uint32_t* total_out, const float4* d_nrm = nullptr, float4** d_out_nrm = nullptr,
                                   const float* d_inten = nullptr, float** d_out_inten = nullptr) {
  *d_out = nullptr;
  *total_out = 0;
  if (d_out_nrm) *d_out_nrm = nullptr;
  if (d_out_inten) *d_out_inten = nullptr;
  VoxelSegments vs;
  lh_status st = voxel_segments(c, d_in, n, leaf, limit_axis, lo, hi, &vs);
  if (st) return st;
  if (vs.total > 0) {
    const size_t n_pad = (size_t)round_up((int)vs.total, 256);
    if (lhMalloc(d_out, sizeof(float4) * (d_out_inten ? n_pad : (size_t)vs.total)) != hipSuccess) { vs.release(); return LH_ENOMEM; }
    if (d_out_inten && lhMalloc(d_out_inten, sizeof(float) * n_pad) != hipSuccess) { (void)lhFree(*d_out); *d_out = nullptr; vs.release(); return LH_ENOMEM; }
    if (d_nrm && d_out_nrm && lhMalloc(d_out_nrm, sizeof(float4) * n_pad) != hipSuccess) {  // n_pad entries, like every cloud's normals
      (void)lhFree(*d_out); *d_out = nullptr;
      if (d_out_inten) { (void)lhFree(*d_out_inten); *d_out_inten = nullptr; }
      vs.release(); return LH_ENOMEM;
    }
    ProfScope p(c, "voxel_centroids", (d_nrm ? 64.0 : 32.0) * n);
    launch_voxel_centroids(d_in, d_nrm, c->keys1, c->vals1, vs.heads, vs.rank, n, *d_out, d_out_nrm ? *d_out_nrm : nullptr, vs.total, c->stream, d_inten,
                           d_out_inten ? *d_out_inten : nullptr);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  *total_out = vs.total;
  vs.release();
  return e == hipSuccess ? LH_OK : LH_EDEVICE;
}

lh_status lh_voxel_grid(lh_ctx* c, const lh_cloud_view* in, float leaf, int limit_axis, double lo, double hi, float* out_xyzi,
                        uint32_t out_capacity, uint32_t* out_count) {
  if (!c || !in || !in->base || !out_count || !(leaf > 0.0f) || limit_axis > 2) return LH_EINVAL;
  if (out_capacity > 0 && !out_xyzi) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->device));
  *out_count = 0;
  int n = (int)in->count;
  if (n == 0) return LH_OK;
  std::vector<float> host((size_t)n * 4);  // pack x,y,z,intensity
  const char* base = (const char*)in->base;
  for (int i = 0; i < n; i++) {
    const char* p = base + (size_t)i * in->stride;
    memcpy(&host[4 * (size_t)i], p + in->off_xyz, 12);
    host[4 * (size_t)i + 3] = (in->off_intensity != UINT32_MAX) ? *(const float*)(p + in->off_intensity) : 0.0f;
  }
  float4 *d_in = nullptr, *d_out = nullptr;
  HIPCHK(lhMalloc(&d_in, sizeof(float4) * (size_t)n));
  HIPCHK(hipMemcpyAsync(d_in, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice, c->stream));
  uint32_t total = 0;
  lh_status st = voxel_grid_device(c, d_in, n, leaf, limit_axis, lo, hi, &d_out, &total);
  if (!st) {
    *out_count = total;
    uint32_t ncopy = std::min(total, out_capacity);
    if (ncopy && hipMemcpy(out_xyzi, d_out, sizeof(float4) * (size_t)ncopy, hipMemcpyDeviceToHost) != hipSuccess) st = LH_EDEVICE;
  }
  (void)lhFree(d_in);
  (void)lhFree(d_out);
  return st;
}

// device-resident variant: cloud in -> new cloud out (x, y, z, intensity centroids; no normals), nothing crosses PCIe
static lh_status cloud_voxel_grid(const lh_cloud* in, float leaf, int limit_axis, double lo, double hi, bool all_fields, lh_cloud** out) {
  if (!in || !out || !(leaf > 0.0f) || limit_axis > 2 || in->n <= 0) return LH_EINVAL;
  if (all_fields && !in->nrm) return LH_EINVAL;  // the PointXYZINormal flavour needs the normal / curvature fields
  lh_ctx* c = in->ctx;
  HIPCHK(hipSetDevice(c->device));
  float4 *d_out = nullptr, *d_out_nrm = nullptr;
  float* d_out_inten = nullptr;
  uint32_t total = 0;
  // the cloud's own arrays in, a cloud's arrays out (round 5: the pack / unpack passes around the filter are gone)
  lh_status st = voxel_grid_device(c, in->xyz, in->n, leaf, limit_axis, lo, hi, &d_out, &total, all_fields ? in->nrm : nullptr,
                                   all_fields ? &d_out_nrm : nullptr, in->intensity, &d_out_inten);
  if (st || total == 0) {
    (void)lhFree(d_out); (void)lhFree(d_out_nrm); (void)lhFree(d_out_inten);
    return st ? st : LH_EINVAL;  // every point was filtered out: no cloud to return
  }
  lh_cloud* o = new lh_cloud();
  o->ctx = c;
  o->n = (int)total;
  o->n_pad = round_up(o->n, 256);
  o->xyz = d_out;
  o->intensity = d_out_inten;
  o->nrm = d_out_nrm;
  *out = o;
  return LH_OK;
}
lh_status lh_cloud_voxel_grid(const lh_cloud* in, float leaf, int limit_axis, double lo, double hi, lh_cloud** out) {
  return cloud_voxel_grid(in, leaf, limit_axis, lo, hi, false, out);
}
// pcl::VoxelGrid<PointF> of PointCloudFilter::Filter (PointCloudFilter.cc:119-124): same voxels, same order, every field averaged
lh_status lh_cloud_voxel_grid_pointf(const lh_cloud* in, float leaf, lh_cloud** out) {
  return cloud_voxel_grid(in, leaf, -1, -3.0e38, 3.0e38, true, out);
}

// ---- NDT (registration_method: ndt) -------------------------------------------------------------------------------
// pclomp::NormalDistributionsTransform on the device: the target's voxel statistics and every (score, gradient, hessian)
// evaluation are kernels (k_ndt_voxel_stats, k_ndt_derivs); the per-cell 3x3 algebra and the Newton / More-Thuente control flow
// (a handful of evaluations per iteration) run on the host (lh_ndt_host.hpp).
struct lh_ndt {
  lh_ctx* ctx = nullptr;
  lh_ndt_params P;
  lh_cloud *src = nullptr, *tgt = nullptr;
  bool own_src = false, own_tgt = false;
  // target cells (ascending voxel index = the order of VoxelGridCovariance's centroid cloud)
  bool grid_valid = false;
  int n_cells = 0;
  lh_cloud* cells = nullptr;          // centroids as a cloud + its radix-tree index (the kd-tree of the reference)
  double *d_mean = nullptr, *d_icov = nullptr;
  // evaluation buffers
  double* rows = nullptr;             // per-wave partial rows (device)
  int rows_cap = 0;
  double* chunks = nullptr;           // [FINAL_CHUNKS][NDT_ROW], pinned, written by k_rows_final
  float last_T[16];
  bool have_result = false;
};

static void ndt_drop_grid(lh_ndt* g) {
  cloud_free(g->cells);
  g->cells = nullptr;
  (void)lhFree(g->d_mean); (void)lhFree(g->d_icov);
  g->d_mean = g->d_icov = nullptr;
  g->n_cells = 0;
  g->grid_valid = false;
}

// VoxelGridCovariance::filter(true) (ndt_omp.h:257-262): voxel statistics of the target, entirely on the device: raw sums per
// voxel -> per-voxel algebra (covariance, eigenvalue inflation, inverse) -> compaction of the voxels with enough points (ascending
// voxel index) -> the centroids become a cloud with the usual radix-tree index.  The host only learns the cell count.
static lh_status ndt_build_grid(lh_ndt* g) {
  lh_ctx* c = g->ctx;
  lh_cloud* t = g->tgt;
  if (!t || t->n <= 0) return LH_EINVAL;
  ndt_drop_grid(g);
  VoxelSegments vs;
  lh_status st = voxel_segments(c, t->xyz, t->n, g->P.resolution, -1, -3.0e38, 3.0e38, &vs);
  if (st) return st;
  const int nv = (int)vs.total;
  if (nv == 0) { vs.release(); g->grid_valid = true; return LH_OK; }
  NdtVoxelRaw* d_raw = nullptr;
  double *v_mean = nullptr, *v_icov = nullptr;
  float4* v_cen = nullptr;
  uint32_t *d_flags = nullptr, *d_incl = nullptr;
  void* d_scan = nullptr;
  size_t scan_bytes = scan_temp_bytes(nv);
  auto cleanup = [&]() { (void)lhFree(d_raw); (void)lhFree(v_mean); (void)lhFree(v_icov); (void)lhFree(v_cen); (void)lhFree(d_flags); (void)lhFree(d_incl); (void)lhFree(d_scan); vs.release(); };
  hipError_t e = lhMalloc(&d_raw, sizeof(NdtVoxelRaw) * (size_t)nv);
  if (e == hipSuccess) e = lhMalloc(&v_mean, sizeof(double) * 3 * (size_t)nv);
  if (e == hipSuccess) e = lhMalloc(&v_icov, sizeof(double) * 9 * (size_t)nv);
  if (e == hipSuccess) e = lhMalloc(&v_cen, sizeof(float4) * (size_t)nv);
  if (e == hipSuccess) e = lhMalloc(&d_flags, sizeof(uint32_t) * (size_t)nv);
  if (e == hipSuccess) e = lhMalloc(&d_incl, sizeof(uint32_t) * (size_t)nv);
  if (e == hipSuccess) e = lhMalloc(&d_scan, scan_bytes ? scan_bytes : 16);
  if (e != hipSuccess) { cleanup(); return LH_ENOMEM; }
  { ProfScope p(c, "ndt_voxel_stats", 16.0 * t->n);
    launch_ndt_voxel_stats(t->xyz, c->keys1, c->vals1, vs.heads, vs.rank, t->n, d_raw, c->stream);
    launch_ndt_finish_cells(d_raw, nv, g->P.min_points_per_voxel, g->P.min_covar_eigvalue_mult, v_mean, v_icov, v_cen, d_flags, c->stream);
    inclusive_scan_u32(d_scan, scan_bytes, d_flags, d_incl, nv, c->stream); }
  uint32_t n_cells = 0;
  e = hipMemcpyAsync(&n_cells, d_incl + (nv - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { cleanup(); return LH_EDEVICE; }
  g->n_cells = (int)n_cells;
  if (n_cells > 0) {
    lh_cloud* cl = new lh_cloud();
    cl->ctx = c; cl->n = (int)n_cells; cl->n_pad = round_up(cl->n, 256);
    e = lhMalloc(&cl->xyz, sizeof(float4) * (size_t)cl->n_pad);
    if (e == hipSuccess) e = lhMalloc(&g->d_mean, sizeof(double) * 3 * (size_t)n_cells);
    if (e == hipSuccess) e = lhMalloc(&g->d_icov, sizeof(double) * 9 * (size_t)n_cells);
    if (e != hipSuccess) { cloud_free(cl); cleanup(); return LH_ENOMEM; }
    g->cells = cl;
    launch_ndt_compact_cells(d_incl, nv, v_mean, v_icov, v_cen, g->d_mean, g->d_icov, cl->xyz, c->stream);
    st = cloud_build_index(cl);
    if (!st && (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) st = LH_EDEVICE;
  }
  cleanup();
  if (st) return st;
  g->grid_valid = true;
  return LH_OK;
}

// one evaluation at pose p: (score, gradient, hessian) = sums over the source points of k_ndt_derivs
static lh_status ndt_evaluate(lh_ndt* g, const double* p6, const float* T16, int want_h, int hessian_only, double* score, double* grad6, double* hess36) {
  lh_ctx* c = g->ctx;
  const int n = g->src->n;
  *score = 0;
  for (int k = 0; k < 6; k++) grad6[k] = 0;
  for (int k = 0; k < 36; k++) hess36[k] = 0;
  if (g->n_cells == 0) return LH_OK;   // no usable voxel: every neighbourhood is empty
  int n_rows = ((n + 255) / 256) * 4;
  if (n_rows > g->rows_cap) {
    (void)lhFree(g->rows);
    g->rows = nullptr;
    HIPCHK(lhMalloc(&g->rows, sizeof(double) * NDT_ROW * (size_t)n_rows));
    g->rows_cap = n_rows;
  }
  if (!g->chunks) HIPCHK(hipHostMalloc(&g->chunks, sizeof(double) * FINAL_CHUNKS * NDT_ROW, hipHostMallocDefault));
  NdtFrame f;
  ndt_fill_frame(f, p6, T16, g->P.resolution, g->P.outlier_ratio, want_h);
  { ProfScope p(c, hessian_only ? "ndt_hessian" : "ndt_derivatives", 16.0 * n);
    launch_ndt_derivs(g->src->xyz, n, g->cells->view(), g->d_mean, g->d_icov, f, hessian_only, g->rows, g->chunks, c->stream); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  double S[NDT_NSUM];
  for (int k = 0; k < NDT_NSUM; k++) S[k] = 0.0;
  for (int ch = 0; ch < FINAL_CHUNKS; ch++)  // fixed order => bitwise reproducible
    for (int k = 0; k < NDT_NSUM; k++) S[k] += g->chunks[ch * NDT_ROW + k];
  if (!hessian_only) { *score = S[0]; for (int k = 0; k < 6; k++) grad6[k] = S[1 + k]; }
  if (want_h) for (int k = 0; k < 36; k++) hess36[k] = S[7 + k];
  return LH_OK;
}

void lh_default_ndt_params(lh_ndt_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->resolution = 1.0f;               // ndt_omp_impl.hpp:50
  p->step_size = 0.1;                 // :51
  p->outlier_ratio = 0.55;            // :52
  p->transformation_epsilon = 0.1;    // :93
  p->max_iterations = 35;             // :94
  p->min_points_per_voxel = 6;        // voxel_grid_covariance_omp.h:186
  p->min_covar_eigvalue_mult = 0.01;  // :187
}
lh_status lh_ndt_create(lh_ctx* ctx, const lh_ndt_params* p, lh_ndt** out) {
  if (!ctx || !out) return LH_EINVAL;
  lh_ndt* g = new lh_ndt();
  g->ctx = ctx;
  if (p) g->P = *p; else lh_default_ndt_params(&g->P);
  memcpy(g->last_T, I16, sizeof(I16));
  *out = g;
  return LH_OK;
}
void lh_ndt_destroy(lh_ndt* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  ndt_drop_grid(g);
  if (g->own_src) cloud_free(g->src);
  if (g->own_tgt) cloud_free(g->tgt);
  (void)lhFree(g->rows);
  if (g->chunks) (void)hipHostFree(g->chunks);
  delete g;
}
lh_status lh_ndt_set_params(lh_ndt* g, const lh_ndt_params* p) {
  if (!g || !p || !(p->resolution > 0.0f)) return LH_EINVAL;
  bool regrid = p->resolution != g->P.resolution || p->min_points_per_voxel != g->P.min_points_per_voxel ||
                p->min_covar_eigvalue_mult != g->P.min_covar_eigvalue_mult;
  g->P = *p;
  if (regrid) g->grid_valid = false;   // setResolution re-initialises the voxel structure (ndt_omp.h:124-131)
  return LH_OK;
}
lh_status lh_ndt_set_source_cloud(lh_ndt* g, lh_cloud* c) {
  if (!g || !c || c->ctx != g->ctx) return LH_EINVAL;
  if (g->own_src) cloud_free(g->src);
  g->src = c; g->own_src = false;
  return LH_OK;
}
lh_status lh_ndt_set_target_cloud(lh_ndt* g, lh_cloud* c) {
  if (!g || !c || c->ctx != g->ctx) return LH_EINVAL;
  if (g->own_tgt) cloud_free(g->tgt);
  g->tgt = c; g->own_tgt = false;
  g->grid_valid = false;               // setInputTarget -> init() (ndt_omp.h:116-119)
  return LH_OK;
}
lh_status lh_ndt_set_source(lh_ndt* g, const lh_cloud_view* v) {
  if (!g || !v) return LH_EINVAL;
  HIPCHK(hipSetDevice(g->ctx->device));
  lh_cloud* c = nullptr;
  lh_status st = upload_view(g->ctx, v, &c);
  if (st) return st;
  if (g->own_src) cloud_free(g->src);
  g->src = c; g->own_src = true;
  return LH_OK;
}
lh_status lh_ndt_set_target(lh_ndt* g, const lh_cloud_view* v) {
  if (!g || !v) return LH_EINVAL;
  HIPCHK(hipSetDevice(g->ctx->device));
  lh_cloud* c = nullptr;
  lh_status st = upload_view(g->ctx, v, &c);
  if (st) return st;
  if (g->own_tgt) cloud_free(g->tgt);
  g->tgt = c; g->own_tgt = true;
  g->grid_valid = false;
  return LH_OK;
}
// test hook: the target cells (count returned through *n_cells; arrays nullable, at most cap cells written)
lh_status lh_ndt_debug_cells(lh_ndt* g, int* n_cells, double* mean3, double* icov9, float* centroid4, int cap) {
  if (!g || !n_cells || !g->tgt) return LH_EINVAL;
  HIPCHK(hipSetDevice(g->ctx->device));
  if (!g->grid_valid) { lh_status st = ndt_build_grid(g); if (st) return st; }
  *n_cells = g->n_cells;
  int k = std::min(cap, g->n_cells);
  if (k > 0) {
    lh_ctx* c = g->ctx;
    if (mean3) HIPCHK(hipMemcpyAsync(mean3, g->d_mean, sizeof(double) * 3 * (size_t)k, hipMemcpyDeviceToHost, c->stream));
    if (icov9) HIPCHK(hipMemcpyAsync(icov9, g->d_icov, sizeof(double) * 9 * (size_t)k, hipMemcpyDeviceToHost, c->stream));
    if (centroid4) HIPCHK(hipMemcpyAsync(centroid4, g->cells->xyz, sizeof(float) * 4 * (size_t)k, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return LH_OK;
}
// test hook: computeDerivatives (hessian_only = 0) / computeHessian (hessian_only = 1) at pose p6
lh_status lh_ndt_debug_derivatives(lh_ndt* g, const double p6[6], int want_h, int hessian_only, double* score, double grad6[6], double hess36[36]) {
  if (!g || !p6 || !score || !grad6 || !hess36 || !g->src || !g->tgt) return LH_EINVAL;
  HIPCHK(hipSetDevice(g->ctx->device));
  if (!g->grid_valid) { lh_status st = ndt_build_grid(g); if (st) return st; }
  float T16[16];
  ndt_pose_to_matrix(p6, T16);
  return ndt_evaluate(g, p6, T16, want_h, hessian_only, score, grad6, hess36);
}
// pcl::Registration::align + computeTransformation (ndt_omp_impl.hpp:101-212).  out->fitness = trans_probability_ (score / n),
// out->cost_passes = device evaluations; aligned_out (nullable) receives final_T * input
lh_status lh_ndt_align(lh_ndt* g, const float guess[16], lh_gicp_result* out, void* aligned_out, uint32_t stride, uint32_t off_xyz) {
  if (!g || !out || !g->src || !g->tgt || g->src->n <= 0) return LH_EINVAL;
  lh_ctx* c = g->ctx;
  HIPCHK(hipSetDevice(c->device));
  if (!g->grid_valid) { lh_status st = ndt_build_grid(g); if (st) return st; }
  bool ident = true;
  if (guess)
    for (int k = 0; k < 16; k++)
      if (guess[k] != I16[k]) ident = false;
  lh_status dev_status = LH_OK;
  NdtEval eval = [&](const double* p6, const float* T16, int want_h, int hessian_only, double* score, double* grad6, double* hess36) {
    dev_status = ndt_evaluate(g, p6, T16, want_h, hessian_only, score, grad6, hess36);
    return dev_status == LH_OK;
  };
  NdtOutcome o;
  memset(out, 0, sizeof(*out));
  memcpy(out->T, I16, sizeof(I16));
  out->fitness = NAN;
  if (!ndt_compute_transformation(eval, guess, ident, g->P.step_size, g->P.transformation_epsilon, g->P.max_iterations, &o)) {
    out->status = dev_status ? dev_status : LH_EDEVICE;
    return out->status;
  }
  memcpy(out->T, o.T, sizeof(o.T));
  out->converged = o.converged;
  out->iterations = o.iterations;
  out->cost_passes = o.evaluations;
  out->n_correspondences_last = g->n_cells;
  out->fitness = o.score / (double)g->src->n;   // trans_probability_ (ndt_omp_impl.hpp:211)
  out->status = LH_OK;
  memcpy(g->last_T, o.T, sizeof(o.T));
  g->have_result = true;
  if (aligned_out) {
    float T12[12];
    fill_T12(o.T, T12);
    float4* d_out = nullptr;
    HIPCHK(lhMalloc(&d_out, sizeof(float4) * (size_t)g->src->n));
    launch_transform(g->src->xyz, nullptr, g->src->n, T12, d_out, nullptr, c->stream);
    std::vector<float> host((size_t)g->src->n * 4);
    hipError_t e = hipMemcpyAsync(host.data(), d_out, sizeof(float) * host.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)lhFree(d_out);
    if (e != hipSuccess) return LH_EDEVICE;
    for (int i = 0; i < g->src->n; i++) memcpy((char*)aligned_out + (size_t)i * stride + off_xyz, &host[4 * (size_t)i], 12);
  }
  return LH_OK;
}

// ---- BodyFilter (body_filter.cc:27-52): CropBox, order-preserving, on the device ---------------------------------------------
static lh_status compact_cloud(const lh_cloud* in, uint32_t* d_flags, lh_cloud** out) {  // flags -> scan -> new cloud
  lh_ctx* c = in->ctx;
  const int n = in->n;
  uint32_t* d_incl = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = scan_temp_bytes(n);
  hipError_t e = lhMalloc(&d_incl, sizeof(uint32_t) * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16);
  if (e != hipSuccess) { (void)lhFree(d_incl); (void)lhFree(d_tmp); return LH_ENOMEM; }
  inclusive_scan_u32(d_tmp, tmp_bytes, d_flags, d_incl, n, c->stream);
  uint32_t total = 0;
  e = hipMemcpyAsync(&total, d_incl + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  lh_status st = e == hipSuccess ? LH_OK : LH_EDEVICE;
  lh_cloud* o = nullptr;
  if (!st && total == 0) st = LH_EINVAL;  // nothing survives: no cloud to return
  if (!st) {
    o = new lh_cloud();
    o->ctx = c; o->n = (int)total; o->n_pad = round_up(o->n, 256);
    if (lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad) != hipSuccess ||
        (in->nrm && lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad) != hipSuccess) ||
        (in->intensity && lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad) != hipSuccess))
      st = LH_ENOMEM;
  }
  if (!st) {
    launch_map_compact(d_incl, n, in->xyz, in->nrm, in->intensity, 1.0, 0, o->xyz, o->nrm, o->intensity, nullptr, c->stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) st = LH_EDEVICE;
  }
  (void)lhFree(d_incl); (void)lhFree(d_tmp);
  if (st) { cloud_free(o); return st; }
  *out = o;
  return LH_OK;
}
lh_status lh_cloud_crop_box(const lh_cloud* in, const float min_pt[3], const float max_pt[3], float yaw, int negative, lh_cloud** out) {
  if (!in || !min_pt || !max_pt || !out || in->n <= 0) return LH_EINVAL;
  lh_ctx* c = in->ctx;
  HIPCHK(hipSetDevice(c->device));
  uint32_t* d_flags = nullptr;
  HIPCHK(lhMalloc(&d_flags, sizeof(uint32_t) * (size_t)in->n));
  { ProfScope p(c, "crop_box", 20.0 * in->n); launch_crop_flags(in->xyz, in->n, min_pt, max_pt, cosf(yaw), sinf(yaw), negative, d_flags, c->stream); }
  lh_status st = compact_cloud(in, d_flags, out);
  (void)lhFree(d_flags);
  return st;
}

// ---- local map: the state behind mapper_->InsertPoints / ApproxNearestNeighbors / Refresh (Locus.cc:464-465,
// 479-483, 531-538), device resident.  point_cloud_mapper is un-vendored ("parity unpinned"); restated from its BLAM lineage:
// a point enters the map iff the octree voxel it falls into is still empty, so the map holds one point per voxel of edge
// `resolution` (the first one offered, in input order).  Voxel = floor(double(p) / resolution) here (PCL's octree anchors its
// lattice at a bounding box that grows with the data; the lattice phase is the unpinned part).
struct lh_map {
  lh_ctx* ctx = nullptr;
  double res = 0.0;
  lh_cloud* cloud = nullptr;   // n = points in the map; buffers hold `cap` points
  int cap = 0;
  uint64_t* keys = nullptr;    // sorted occupancy keys, one per map point
};

static lh_status map_reserve(lh_map* m, int need, bool with_nrm, bool with_inten) {
  lh_cloud* c = m->cloud;
  if (need <= m->cap && (!with_nrm || c->nrm) && (!with_inten || c->intensity)) return LH_OK;
  lh_ctx* x = m->ctx;
  int cap = std::max(need, m->cap);
  if (need > m->cap) cap = round_up(std::max(need + need / 2, 4096), 256);
  float4 *xyz = nullptr, *nrm = nullptr;
  float* inten = nullptr;
  uint64_t* keys = nullptr;
  bool want_n = with_nrm || c->nrm, want_i = with_inten || c->intensity;
  hipError_t e = lhMalloc(&xyz, sizeof(float4) * (size_t)cap);
  if (e == hipSuccess && want_n) e = lhMalloc(&nrm, sizeof(float4) * (size_t)cap);
  if (e == hipSuccess && want_i) e = lhMalloc(&inten, sizeof(float) * (size_t)cap);
  if (e == hipSuccess) e = lhMalloc(&keys, sizeof(uint64_t) * (size_t)cap);
  if (e == hipSuccess && want_n) e = hipMemsetAsync(nrm, 0, sizeof(float4) * (size_t)cap, x->stream);
  if (e == hipSuccess && want_i) e = hipMemsetAsync(inten, 0, sizeof(float) * (size_t)cap, x->stream);
  if (e == hipSuccess && c->n > 0) {
    e = hipMemcpyAsync(xyz, c->xyz, sizeof(float4) * (size_t)c->n, hipMemcpyDeviceToDevice, x->stream);
    if (e == hipSuccess && c->nrm) e = hipMemcpyAsync(nrm, c->nrm, sizeof(float4) * (size_t)c->n, hipMemcpyDeviceToDevice, x->stream);
    if (e == hipSuccess && c->intensity) e = hipMemcpyAsync(inten, c->intensity, sizeof(float) * (size_t)c->n, hipMemcpyDeviceToDevice, x->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(keys, m->keys, sizeof(uint64_t) * (size_t)c->n, hipMemcpyDeviceToDevice, x->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
  if (e != hipSuccess) { (void)lhFree(xyz); (void)lhFree(nrm); (void)lhFree(inten); (void)lhFree(keys); return e == hipErrorOutOfMemory ? LH_ENOMEM : LH_EDEVICE; }
  (void)lhFree(c->xyz); (void)lhFree(c->nrm); (void)lhFree(c->intensity); (void)lhFree(m->keys); (void)lhFree(c->cov6);
  c->xyz = xyz; c->nrm = nrm; c->intensity = inten; m->keys = keys; c->cov6 = nullptr; c->cov_k = 0;
  c->n_pad = cap;
  m->cap = cap;
  return LH_OK;
}

lh_status lh_map_create(lh_ctx* ctx, double octree_resolution, lh_map** out) {
  if (!ctx || !out || !(octree_resolution > 0.0)) return LH_EINVAL;
  lh_map* m = new lh_map();
  m->ctx = ctx;
  m->res = octree_resolution;
  m->cloud = new lh_cloud();
  m->cloud->ctx = ctx;
  *out = m;
  return LH_OK;
}
void lh_map_destroy(lh_map* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  (void)lhFree(m->keys);
  cloud_free(m->cloud);
  delete m;
}
uint32_t lh_map_size(const lh_map* m) { return m ? (uint32_t)m->cloud->n : 0; }
lh_cloud* lh_map_cloud(lh_map* m) { return (m && m->cloud->n > 0) ? m->cloud : nullptr; }

// sort `n` keys of the map in place (through a temporary)
static lh_status map_sort_keys(lh_map* m, int n) {
  if (n <= 1) return LH_OK;
  lh_ctx* x = m->ctx;
  uint64_t* tmp = nullptr;
  void* st = nullptr;
  size_t sb = sort_keys64_temp_bytes(n);
  hipError_t e = lhMalloc(&tmp, sizeof(uint64_t) * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&st, sb ? sb : 16);
  if (e == hipSuccess) {
    sort_keys_u64(st, sb, m->keys, tmp, n, x->stream);
    e = hipMemcpyAsync(m->keys, tmp, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToDevice, x->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
  (void)lhFree(tmp); (void)lhFree(st);
  return e == hipSuccess ? LH_OK : LH_EDEVICE;
}

lh_status lh_map_insert(lh_map* m, const lh_cloud* pts, uint32_t* n_inserted) {
  if (!m || !pts || pts->ctx != m->ctx || pts->n <= 0) return LH_EINVAL;
  lh_ctx* x = m->ctx;
  HIPCHK(hipSetDevice(x->device));
  const int n = pts->n, m0 = m->cloud->n;
  const double inv_res = 1.0 / m->res;
  uint64_t *k0 = nullptr, *k1 = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr, *acc = nullptr, *incl = nullptr;
  void *st = nullptr, *sc = nullptr;
  size_t sb = sort64_temp_bytes(n), cb = scan_temp_bytes(n);
  auto cleanup = [&]() { (void)lhFree(k0); (void)lhFree(k1); (void)lhFree(v0); (void)lhFree(v1); (void)lhFree(acc); (void)lhFree(incl); (void)lhFree(st); (void)lhFree(sc); };
  hipError_t e = lhMalloc(&k0, 8 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&k1, 8 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&v0, 4 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&v1, 4 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&acc, 4 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&incl, 4 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&st, sb ? sb : 16);
  if (e == hipSuccess) e = lhMalloc(&sc, cb ? cb : 16);
  if (e != hipSuccess) { cleanup(); return LH_ENOMEM; }
  uint32_t total = 0;
  {
    ProfScope p(x, "map_insert", 48.0 * n);
    launch_map_keys(pts->xyz, n, inv_res, k0, v0, x->stream);
    sort_pairs_u64(st, sb, k0, k1, v0, v1, n, 64, x->stream);   // stable: equal voxels keep input order
    launch_map_accept(k1, v1, n, m->keys, m0, acc, x->stream);
    inclusive_scan_u32(sc, cb, acc, incl, n, x->stream);
  }
  e = hipMemcpyAsync(&total, incl + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
  if (e != hipSuccess) { cleanup(); return LH_EDEVICE; }
  lh_status stt = LH_OK;
  if (total > 0) {
    stt = map_reserve(m, m0 + (int)total, pts->nrm != nullptr, pts->intensity != nullptr);
    if (!stt) {
      lh_cloud* c = m->cloud;
      launch_map_compact(incl, n, pts->xyz, pts->nrm, pts->intensity, inv_res, m0, c->xyz, c->nrm, c->intensity, m->keys, x->stream);
      if (hipGetLastError() != hipSuccess) stt = LH_EDEVICE;
      if (!stt) stt = map_sort_keys(m, m0 + (int)total);
      if (!stt) { c->n = m0 + (int)total; c->has_index = false; c->cov_k = 0; }
    }
  }
  (void)hipStreamSynchronize(x->stream);
  cleanup();
  if (!stt && n_inserted) *n_inserted = total;
  return stt;
}

// mapper_->Refresh(current_pose) with box_filter_size (lo_settings.yaml:58): the sliding-window crop of the local map
lh_status lh_map_refresh(lh_map* m, const float center[3], float half_extent) {
  if (!m || !center || !(half_extent > 0.0f)) return LH_EINVAL;
  lh_ctx* x = m->ctx;
  lh_cloud* c = m->cloud;
  const int n = c->n;
  if (n == 0) return LH_OK;
  HIPCHK(hipSetDevice(x->device));
  uint32_t *flags = nullptr, *incl = nullptr;
  void* sc = nullptr;
  size_t cb = scan_temp_bytes(n);
  float4 *xyz = nullptr, *nrm = nullptr;
  float* inten = nullptr;
  uint64_t* keys = nullptr;
  auto cleanup = [&]() { (void)lhFree(flags); (void)lhFree(incl); (void)lhFree(sc); (void)lhFree(xyz); (void)lhFree(nrm); (void)lhFree(inten); (void)lhFree(keys); };
  hipError_t e = lhMalloc(&flags, 4 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&incl, 4 * (size_t)n);
  if (e == hipSuccess) e = lhMalloc(&sc, cb ? cb : 16);
  if (e == hipSuccess) e = lhMalloc(&xyz, sizeof(float4) * (size_t)m->cap);
  if (e == hipSuccess && c->nrm) e = lhMalloc(&nrm, sizeof(float4) * (size_t)m->cap);
  if (e == hipSuccess && c->intensity) e = lhMalloc(&inten, sizeof(float) * (size_t)m->cap);
  if (e == hipSuccess) e = lhMalloc(&keys, sizeof(uint64_t) * (size_t)m->cap);
  if (e != hipSuccess) { cleanup(); return LH_ENOMEM; }
  uint32_t total = 0;
  {
    ProfScope p(x, "map_refresh", 64.0 * n);
    launch_box_flags(c->xyz, n, center[0], center[1], center[2], half_extent, flags, x->stream);
    inclusive_scan_u32(sc, cb, flags, incl, n, x->stream);
    launch_map_compact(incl, n, c->xyz, c->nrm, c->intensity, 1.0 / m->res, 0, xyz, nrm, inten, keys, x->stream);
  }
  e = hipMemcpyAsync(&total, incl + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
  if (e != hipSuccess) { cleanup(); return LH_EDEVICE; }
  std::swap(c->xyz, xyz); std::swap(c->nrm, nrm); std::swap(c->intensity, inten); std::swap(m->keys, keys);
  c->n = (int)total;
  c->has_index = false;
  c->cov_k = 0;
  cleanup();  // frees the old buffers (now in the temporaries)
  return map_sort_keys(m, (int)total);
}

// ---- next-row helper: mapper_->ApproxNearestNeighbors (Locus.cc:479-483) ------------------------------
// for every query point the nearest map point is copied (xyz, normal, intensity) into a new cloud; the reference uses an
// approximate octree search, this is the exact search (never farther than the reference's answer)
static __global__ void __launch_bounds__(256) k_gather_cloud(const float4* __restrict__ xyz, const float4* __restrict__ nrm, const float* __restrict__ inten,
                                                     const int32_t* __restrict__ idx, int n, float4* __restrict__ oxyz, float4* __restrict__ onrm,
                                                     float* __restrict__ ointen) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int j = idx[i];
  if (j < 0) {  // no neighbour (a non-finite query point: every comparison of the search fails): a NaN point, never an out-of-bounds read
    const float qn = __int_as_float(0x7fc00000);
    oxyz[i] = make_float4(qn, qn, qn, 1.0f);
    if (nrm && onrm) onrm[i] = make_float4(qn, qn, qn, qn);
    if (inten && ointen) ointen[i] = qn;
    return;
  }
  oxyz[i] = xyz[j];
  if (nrm && onrm) onrm[i] = nrm[j];
  if (inten && ointen) ointen[i] = inten[j];
}
lh_status lh_cloud_nearest_neighbors(lh_cloud* map, const lh_cloud* query, lh_cloud** out) {
  if (!map || !query || !out || map->ctx != query->ctx || map->n <= 0 || query->n <= 0) return LH_EINVAL;
  lh_ctx* c = map->ctx;
  HIPCHK(hipSetDevice(c->device));
  if (!map->has_index) { lh_status st = cloud_build_index(map); if (st) return st; }
  int n = query->n;
  int32_t* d_idx = nullptr;
  float* d_d2 = nullptr;
  DevGuard guard;
  HIPCHK(guard.alloc(&d_idx, sizeof(int32_t) * (size_t)n));
  HIPCHK(guard.alloc(&d_d2, sizeof(float) * (size_t)n));
  { ProfScope p(c, "nn1", 24.0 * n); launch_nn1(query->xyz, n, nullptr, map->view(), d_idx, d_d2, c->stream); }
  lh_cloud* o = new lh_cloud();
  guard.cloud = o;
  o->ctx = c; o->n = n; o->n_pad = round_up(n, 256);
  HIPCHK(lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad));
  if (map->nrm) HIPCHK(lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad));
  if (map->intensity) HIPCHK(lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad));
  hipLaunchKernelGGL(k_gather_cloud, dim3((n + 255) / 256), dim3(256), 0, c->stream, map->xyz, map->nrm, map->intensity, d_idx, n, o->xyz, o->nrm,
                     o->intensity);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = guard.keep_cloud();
  return LH_OK;
}

// ---- instrumentation -------------------------------------------------------------------------------------------
lh_status lh_profile_enable(lh_ctx* c, int on) {
  if (!c) return LH_EINVAL;
  c->prof_flush();
  c->prof = on != 0;
  return LH_OK;
}
lh_status lh_profile_reset(lh_ctx* c) {
  if (!c) return LH_EINVAL;
  c->prof_flush();
  c->prof_entries.clear();
  return LH_OK;
}
int lh_profile_get(lh_ctx* c, lh_kernel_stat* out, int cap) {
  if (!c) return 0;
  c->prof_flush();
  int n = (int)c->prof_entries.size();
  for (int i = 0; i < n && i < cap && out; i++) {
    memset(&out[i], 0, sizeof(out[i]));
    strncpy(out[i].name, c->prof_entries[i].name.c_str(), sizeof(out[i].name) - 1);
    out[i].launches = c->prof_entries[i].launches;
    out[i].total_ms = c->prof_entries[i].ms;
    out[i].bytes = c->prof_entries[i].bytes;
  }
  return n;
}


}  // extern "C"
#pragma GCC visibility pop
