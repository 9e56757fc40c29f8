// lh_pool.hip -- device-memory pool, context scratch buffers, cloud release (see lh_runtime.hpp).
#include "lh_runtime.hpp"

// ---- device-memory pool ---------------------------------------------------------------------------------------------------
// Every entry point that returns a new cloud, and every filter stage, needs a few device buffers for the duration of one call.
// hipMalloc costs tens of microseconds and hipFree synchronises the whole device, which made the pre-processing chain of a
// 1 M-point frame (merge -> crop -> voxel grid -> normals: ~1 ms of kernels) take 2.9 ms.  Blocks are therefore recycled:
// lhFree parks a block in a per-device free list (no hipFree, no sync), lhMalloc takes the smallest parked block that fits with
// <= 25 % slack.  Safe because every user allocates, launches and frees on the context's primary stream (a recycled block is
// only reused by work queued behind the work that used it last); the second scheduler stream only ever touches per-slot
// workspaces and context scratch, which are allocated once and not pooled.  The cache is trimmed when it exceeds a quarter of the device's memory.
namespace {
struct DevPool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;   // pooled blocks handed out
  std::multimap<size_t, void*> parked;
  size_t parked_bytes = 0;
};
DevPool g_pools[64];
size_t pool_round(size_t b) {
  if (b < 256) return 256;
  if (b <= (1u << 20)) { size_t r = 256; while (r < b) r <<= 1; return r; }
  return (b + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
}
void pool_trim(DevPool& P) {  // caller holds the lock
  (void)hipDeviceSynchronize();
  for (auto& kv : P.parked) (void)hipFree(kv.second);
  P.parked.clear();
  P.parked_bytes = 0;
}
}  // namespace
hipError_t lhMallocRaw(void** p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevPool& P = g_pools[dev & 63];
  const size_t want = pool_round(bytes);
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.parked.lower_bound(want);
  if (it != P.parked.end() && it->first <= want + want / 4) {
    *p = it->second;
    P.live[*p] = it->first;
    P.parked_bytes -= it->first;
    P.parked.erase(it);
    return hipSuccess;
  }
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess && !P.parked.empty()) {  // out of memory with blocks parked: give them back and retry
    (void)hipGetLastError();
    pool_trim(P);
    e = hipMalloc(p, want);
  }
  if (e == hipSuccess) P.live[*p] = want;
  return e;
}
hipError_t lhFree(void* p) {
  if (!p) return hipSuccess;
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevPool& P = g_pools[dev & 63];
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.live.find(p);
  if (it == P.live.end()) return hipFree(p);  // not from the pool (context scratch, workspaces)
  P.parked.emplace(it->second, p);
  P.parked_bytes += it->second;
  P.live.erase(it);
  // The cache is trimmed (a device synchronisation + one hipFree per parked block: milliseconds) when it exceeds a QUARTER OF THE DEVICE'S MEMORY
  // (72 GB on an MI355X; 8 GB until round 6: a bench run that had just closed a 513-scan trajectory crossed it in the middle of a timed
  // configs[4] frame -- 7.4 ms for a 0.46-ms filter stage, one frame in a dozen runs).  LH_POOL_CAP_GB overrides.
  static const size_t cap = []() {
    const char* e = getenv("LH_POOL_CAP_GB");
    if (e && atof(e) > 0) return (size_t)(atof(e) * (double)((size_t)1 << 30));
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) return (size_t)8 << 30;
    return std::max<size_t>((size_t)8 << 30, tot / 4);
  }();
  if (P.parked_bytes > cap) pool_trim(P);
  return hipSuccess;
}


void cloud_free(lh_cloud* c) {
  if (!c) return;
  (void)lhFree(c->xyz); (void)lhFree(c->nrm); (void)lhFree(c->intensity);
  (void)lhFree(c->sorted); (void)lhFree(c->node_buf); (void)lhFree(c->cov6);
  delete c;
}

lh_status ctx_ensure_scratch(lh_ctx* c, int n) {
  if (n <= c->scratch_n) return LH_OK;
  (void)hipStreamSynchronize(c->stream);
  (void)lhFree(c->keys0); (void)lhFree(c->keys1); (void)lhFree(c->vals0); (void)lhFree(c->vals1); (void)lhFree(c->sort_temp);
  int cap = round_up(n + n / 4, 1024);
  HIPCHK(hipMalloc(&c->keys0, sizeof(uint32_t) * cap));
  HIPCHK(hipMalloc(&c->keys1, sizeof(uint32_t) * cap));
  HIPCHK(hipMalloc(&c->vals0, sizeof(uint32_t) * cap));
  HIPCHK(hipMalloc(&c->vals1, sizeof(uint32_t) * cap));
  c->sort_temp_bytes = sort_temp_bytes(cap);
  HIPCHK(hipMalloc(&c->sort_temp, c->sort_temp_bytes ? c->sort_temp_bytes : 16));
  c->scratch_n = cap;
  return LH_OK;
}

lh_status ctx_ensure_small(lh_ctx* c, size_t doubles) {
  if (doubles <= c->small_host_doubles) return LH_OK;
  (void)hipStreamSynchronize(c->stream);
  if (c->small_host) (void)hipHostFree(c->small_host);
  size_t cap = std::max<size_t>(doubles, 4096);
  HIPCHK(hipHostMalloc(&c->small_host, sizeof(double) * cap, hipHostMallocDefault));
  c->small_host_doubles = cap;
  return LH_OK;
}

