// lh_rccl.hip -- include/locus_hip_rccl.h: the exchange steps of the multi-GPU path on RCCL (one rank per GPU).
// Every payload is tiny (592 B of moment sums, 96 B per result), so each call is latency-bound: one small pinned staging
// buffer, one device buffer, the rank's own stream, and a stream sync before the host reads the answer.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/locus_hip_rccl.h"

static_assert(LH_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

struct lh_rccl {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  char* dev = nullptr;      // device staging
  char* host = nullptr;     // pinned staging
  size_t cap = 0;           // bytes of each
};

#define RCHK(expr)                                                                                             \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "[locus_hip_rccl] %s: %s\n", #expr, hipGetErrorString(e_)); return LH_EDEVICE; } \
  } while (0)
#define NCHK(expr)                                                                                             \
  do {                                                                                                         \
    ncclResult_t r_ = (expr);                                                                                  \
    if (r_ != ncclSuccess) { fprintf(stderr, "[locus_hip_rccl] %s: %s\n", #expr, ncclGetErrorString(r_)); return LH_EDEVICE; } \
  } while (0)

static lh_status ensure(lh_rccl* r, size_t bytes) {
  if (bytes <= r->cap) return LH_OK;
  RCHK(hipStreamSynchronize(r->stream));
  if (r->dev) (void)hipFree(r->dev);
  if (r->host) (void)hipHostFree(r->host);
  r->dev = nullptr; r->host = nullptr; r->cap = 0;
  size_t cap = std::max<size_t>(bytes, 4096);
  RCHK(hipMalloc(&r->dev, cap));
  RCHK(hipHostMalloc(&r->host, cap, hipHostMallocDefault));
  r->cap = cap;
  return LH_OK;
}

extern "C" {

lh_status lh_rccl_get_unique_id(char id[LH_RCCL_ID_BYTES]) {
  if (!id) return LH_EINVAL;
  ncclUniqueId u;
  NCHK(ncclGetUniqueId(&u));
  memcpy(id, u.internal, LH_RCCL_ID_BYTES);
  return LH_OK;
}

lh_status lh_rccl_create(int device_id, const char id[LH_RCCL_ID_BYTES], int rank, int world, lh_rccl** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return LH_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return LH_EDEVICE; }
  if (device_id < 0 || device_id >= ndev) return LH_EINVAL;
  RCHK(hipSetDevice(device_id));
  lh_rccl* r = new lh_rccl();
  r->device = device_id; r->rank = rank; r->world = world;
  ncclUniqueId u;
  memcpy(u.internal, id, LH_RCCL_ID_BYTES);
  ncclResult_t rc = ncclCommInitRank(&r->comm, world, u, rank);
  if (rc != ncclSuccess) {
    fprintf(stderr, "[locus_hip_rccl] ncclCommInitRank: %s\n", ncclGetErrorString(rc));
    delete r;
    return LH_EDEVICE;
  }
  if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) { (void)ncclCommDestroy(r->comm); delete r; return LH_EDEVICE; }
  *out = r;
  return LH_OK;
}

void lh_rccl_destroy(lh_rccl* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->stream) (void)hipStreamSynchronize(r->stream);
  if (r->comm) (void)ncclCommDestroy(r->comm);
  if (r->dev) (void)hipFree(r->dev);
  if (r->host) (void)hipHostFree(r->host);
  if (r->stream) (void)hipStreamDestroy(r->stream);
  delete r;
}

int lh_rccl_rank(const lh_rccl* r) { return r ? r->rank : -1; }
int lh_rccl_world(const lh_rccl* r) { return r ? r->world : 0; }

int lh_rccl_sum_hook(double* sums, int n, void* user) {
  lh_rccl* r = static_cast<lh_rccl*>(user);
  if (!r || !sums || n <= 0) return 1;
  if (hipSetDevice(r->device) != hipSuccess) return 1;
  const size_t bytes = sizeof(double) * (size_t)n;
  if (ensure(r, bytes) != LH_OK) return 1;
  memcpy(r->host, sums, bytes);
  if (hipMemcpyAsync(r->dev, r->host, bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 1;
  if (ncclAllReduce(r->dev, r->dev, (size_t)n, ncclDouble, ncclSum, r->comm, r->stream) != ncclSuccess) return 1;
  if (hipMemcpyAsync(r->host, r->dev, bytes, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 1;
  if (hipStreamSynchronize(r->stream) != hipSuccess) return 1;
  memcpy(sums, r->host, bytes);
  return 0;
}

// ... and on the device: the device-driven loop hands over the chunk sums where they lie and the stream the iteration is queued on; the
// all-reduce is enqueued there, between k_moments_final and k_solve -- no staging buffer, no copy, no synchronisation
int lh_rccl_device_sum_hook(double* dev_sums, int n, void* stream, void* user) {
  lh_rccl* r = static_cast<lh_rccl*>(user);
  if (!r || !dev_sums || n <= 0) return 1;
  return ncclAllReduce(dev_sums, dev_sums, (size_t)n, ncclDouble, ncclSum, r->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

lh_status lh_rccl_install_sum_hook(lh_ctx* ctx, lh_rccl* r) {
  if (!ctx) return LH_EINVAL;
  lh_status st = r ? lh_set_allreduce(ctx, lh_rccl_sum_hook, r) : lh_set_allreduce(ctx, nullptr, nullptr);
  if (st) return st;
  return r ? lh_set_device_allreduce(ctx, lh_rccl_device_sum_hook, r) : lh_set_device_allreduce(ctx, nullptr, nullptr);
}

lh_status lh_rccl_allgather_results(lh_rccl* r, const lh_gicp_result* local, int n_local, lh_gicp_result* all, int cap, int* counts) {
  if (!r || n_local < 0 || (n_local > 0 && !local) || !all) return LH_EINVAL;
  RCHK(hipSetDevice(r->device));
  const int W = r->world;
  // 1) the per-rank counts (they may differ by one when the pairs do not divide evenly) AND every rank's capacity: the decision to
  //    go on is taken from the gathered table, i.e. identically on every rank -- a rank that left between the two collectives on
  //    its own (a smaller `cap`) would leave the others waiting in the second all-gather
  lh_status st = ensure(r, sizeof(int) * 2 * (size_t)(W + 1));
  if (st) return st;
  int* hc = reinterpret_cast<int*>(r->host);
  hc[0] = n_local;
  hc[1] = cap;
  RCHK(hipMemcpyAsync(r->dev, hc, sizeof(int) * 2, hipMemcpyHostToDevice, r->stream));
  NCHK(ncclAllGather(r->dev, r->dev + sizeof(int) * 2, 2, ncclInt32, r->comm, r->stream));
  RCHK(hipMemcpyAsync(hc, r->dev + sizeof(int) * 2, sizeof(int) * 2 * (size_t)W, hipMemcpyDeviceToHost, r->stream));
  RCHK(hipStreamSynchronize(r->stream));
  std::vector<int> cnt(W);
  int kmax = 0, min_cap = cap;
  long total = 0;
  for (int k = 0; k < W; k++) {
    cnt[k] = hc[2 * k];
    kmax = std::max(kmax, cnt[k]);
    total += cnt[k];
    min_cap = std::min(min_cap, hc[2 * k + 1]);
  }
  if (counts) memcpy(counts, cnt.data(), sizeof(int) * (size_t)W);
  if (total > min_cap) return LH_EINVAL;   // on EVERY rank: nobody enters the second collective
  if (kmax == 0) return LH_OK;
  // 2) ONE all-gather of the records, padded to the largest block
  const size_t blk = sizeof(lh_gicp_result) * (size_t)kmax;
  st = ensure(r, blk * (size_t)(W + 1));
  if (st) return st;
  memset(r->host, 0, blk);
  if (n_local) memcpy(r->host, local, sizeof(lh_gicp_result) * (size_t)n_local);
  RCHK(hipMemcpyAsync(r->dev, r->host, blk, hipMemcpyHostToDevice, r->stream));
  NCHK(ncclAllGather(r->dev, r->dev + blk, blk, ncclChar, r->comm, r->stream));
  RCHK(hipMemcpyAsync(r->host + blk, r->dev + blk, blk * (size_t)W, hipMemcpyDeviceToHost, r->stream));
  RCHK(hipStreamSynchronize(r->stream));
  size_t at = 0;
  for (int k = 0; k < W; k++) {
    memcpy(all + at, r->host + blk * (size_t)(k + 1), sizeof(lh_gicp_result) * (size_t)cnt[k]);
    at += (size_t)cnt[k];
  }
  return LH_OK;
}

lh_status lh_rccl_max_double(lh_rccl* r, double* v) {
  if (!r || !v) return LH_EINVAL;
  RCHK(hipSetDevice(r->device));
  lh_status st = ensure(r, sizeof(double));
  if (st) return st;
  memcpy(r->host, v, sizeof(double));
  RCHK(hipMemcpyAsync(r->dev, r->host, sizeof(double), hipMemcpyHostToDevice, r->stream));
  NCHK(ncclAllReduce(r->dev, r->dev, 1, ncclDouble, ncclMax, r->comm, r->stream));
  RCHK(hipMemcpyAsync(r->host, r->dev, sizeof(double), hipMemcpyDeviceToHost, r->stream));
  RCHK(hipStreamSynchronize(r->stream));
  memcpy(v, r->host, sizeof(double));
  return LH_OK;
}

lh_status lh_rccl_barrier(lh_rccl* r) {
  double one = 1.0;
  return lh_rccl_max_double(r, &one);
}

}  // extern "C"
