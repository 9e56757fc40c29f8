// lh_api.hip -- C ABI (include/locus_hip.h) + host runtime of the MI355X GICP hot path.
//
// Runtime model: one lh_ctx per GPU (one HIP stream).  An alignment is a stackful coroutine (ucontext) that runs
// the reference's computeTransformation control flow (gicp.hpp:406-617) and yields two kinds of device requests:
//   SWEEP(T)  -> k_sweep   (NN + Mahalanobis, gicp.hpp:464-498)
//   COST(x)   -> k_cost    (fused f/df pass, gicp.hpp:362-402)
// The scheduler resumes every in-flight pair, batches their requests into ONE launch per kind (grid.y = pair),
// synchronises once per round and feeds the results back.  A single lh_gicp_align is the same machinery with one
// task.  No CPU fallback exists: without a HIP device every entry point returns LH_EDEVICE.
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/locus_hip.h"
#include "lh_bfgs.hpp"
#include "lh_ndt_host.hpp"
#include "lh_kernels.hpp"

using namespace lh;

#define HIPCHK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      fprintf(stderr, "[locus_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LH_EDEVICE;                                                                                  \
    }                                                                                                     \
  } while (0)

static const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

// ---------------------------------------------------------------------------------------------------------
// ---- device-memory pool ---------------------------------------------------------------------------------------------------
// Every entry point that returns a new cloud, and every filter stage, needs a few device buffers for the duration of one call.
// hipMalloc costs tens of microseconds and hipFree synchronises the whole device, which made the pre-processing chain of a
// 1 M-point frame (merge -> crop -> voxel grid -> normals: ~1 ms of kernels) take 2.9 ms.  Blocks are therefore recycled:
// lhFree parks a block in a per-device free list (no hipFree, no sync), lhMalloc takes the smallest parked block that fits with
// <= 25 % slack.  Safe because every user allocates, launches and frees on the context's primary stream (a recycled block is
// only reused by work queued behind the work that used it last); the second scheduler stream only ever touches per-slot
// workspaces and context scratch, which are allocated once and not pooled.  The cache is trimmed when it exceeds 8 GB.
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
static hipError_t lhMallocRaw(void** p, size_t bytes) {
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
template <class T>
static hipError_t lhMalloc(T** p, size_t bytes) { return lhMallocRaw(reinterpret_cast<void**>(p), bytes); }
static hipError_t lhFree(void* p) {
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
  if (P.parked_bytes > ((size_t)8 << 30)) pool_trim(P);
  return hipSuccess;
}

struct ProfEntry { std::string name; uint64_t launches = 0; double ms = 0, bytes = 0; };
struct ProfPending { int entry; hipEvent_t a, b; };

// Small persistent host thread pool: in cost_mode 1 every pair runs its whole BFGS solve (~30 evaluations of the 12x12
// moment model per outer iteration) on the host between two sweeps; with 32 pairs per scheduler group that is ~0.2 ms of
// serial host work per round -- as long as the GPU time of the round.  The solves are independent, so they are spread
// over a few workers (LH_HOST_THREADS, default 8; the calling thread takes part).
struct HostPool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::function<void(int)> fn;
  int n_items = 0, pending = 0;
  std::atomic<int> next{0};
  uint64_t generation = 0;
  bool stop = false;
  explicit HostPool(int n_workers) {
    for (int i = 0; i < n_workers; i++) workers.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> l(m); stop = true; }
    cv_work.notify_all();
    for (auto& t : workers) t.join();
  }
  void drain() {
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n_items) break;
      fn(i);
      std::lock_guard<std::mutex> l(m);
      if (--pending == 0) cv_done.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m);
        cv_work.wait(l, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
      }
      drain();
    }
  }
  void parallel_for(int n, std::function<void(int)> f) {
    if (n <= 0) return;
    if (workers.empty() || n == 1) { for (int i = 0; i < n; i++) f(i); return; }
    {
      std::lock_guard<std::mutex> l(m);
      fn = std::move(f);
      n_items = n;
      pending = n;
      next.store(0);
      generation++;
    }
    cv_work.notify_all();
    drain();
    std::unique_lock<std::mutex> l(m);
    cv_done.wait(l, [&] { return pending == 0; });
  }
};

// per-pair device workspace (owned by a registration object, or -- batch mode -- by a scheduler slot of the context)
struct lh_ctx;
struct Workspace {
  int cap = 0;
  float4* corr = nullptr;
  double* maha6 = nullptr;
  int32_t* prev_nn = nullptr;
  float4* cert = nullptr;     // NN certificates (see Nn1CertCollector)
  float4* rec = nullptr;      // the neighbour prev_nn points at, gathered (position, normal): 2 float4 per source point
  unsigned long long* stats = nullptr;  // 2 counters
  float4* out_xyz = nullptr;  // guess * input when guess != I
  int n_pad = 0;
  lh_status ensure(lh_ctx* c, int n);
  void release();
};

struct lh_ctx {
  int device = 0;
  HostPool* pool = nullptr;
  hipStream_t stream = nullptr, stream2 = nullptr;  // stream2: second half-batch of the pipelined scheduler
  hipStream_t stream3 = nullptr, stream4 = nullptr; // further scheduler groups of the device-driven loop
  static constexpr int MAX_GROUPS = 32;
  hipStream_t stream_more[MAX_GROUPS - 4] = {};   // groups 5..32
  void sync_side_streams() {  // everything the scheduler may have queued besides the primary stream
    if (stream2) (void)hipStreamSynchronize(stream2);
    if (stream3) (void)hipStreamSynchronize(stream3);
    if (stream4) (void)hipStreamSynchronize(stream4);
    for (hipStream_t s : stream_more)
      if (s) (void)hipStreamSynchronize(s);
  }
  // index-build scratch (shared by all clouds of the context; builds are serial on the stream)
  uint32_t *keys0 = nullptr, *keys1 = nullptr, *vals0 = nullptr, *vals1 = nullptr, *bbox = nullptr;
  void* sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  int scratch_n = 0;
  // batched index build scratch
  uint64_t *k64a = nullptr, *k64b = nullptr;
  uint32_t *v32a = nullptr, *v32b = nullptr, *idx_bbox = nullptr;
  uint64_t *k32a = nullptr, *k32b = nullptr;   // the build's radix sort: (key, index) pairs in flight between its passes
  uint32_t* rs_hist = nullptr;                 // ... and its per-tile digit tables
  void* sort64_temp = nullptr;
  size_t sort64_temp_bytes = 0;
  char* tree_tmp = nullptr;        // TreeScratch arrays (TREE_SCRATCH_BYTES_PER_POINT per point)
  void* scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  int idx_cap = 0;
  static constexpr int IDX_STAGE = 40;   // staging ring of the batched index build's descriptors: more than the scheduler's groups, so a build never waits for an older upload
  IndexDesc *idx_descs_dev = nullptr, *idx_descs_host = nullptr;   // host: IDX_STAGE x MAX_INDEX_BATCH entries (pinned)
  hipEvent_t idx_copy_done[IDX_STAGE] = {};
  hipEvent_t idx_build_done = nullptr;
  int idx_stage = 0;
  // pair slots
  PairDesc* descs_dev = nullptr;   // [n_slots]
  PairDesc* descs_host = nullptr;  // pinned staging
  int n_slots = 0;
  double* partials_host = nullptr; // pinned, device-visible: [n_slots][max_cost_blocks][COST_NSUM]
  size_t partials_per_slot = 0;    // doubles
  double* mom_partials_dev = nullptr;  // [n_slots][mom_stride] per-block moment partials (device)
  unsigned long long* wmask_dev = nullptr;  // [n_slots][mask_stride] walker masks of the two-launch sweep (one 64-bit word per wave of source points)
  int mask_stride = 0;
  int mom_stride = 0;
  // device-driven loop (cost_mode 1, k_solve): per-slot loop state, the chunk sums k_moments_final leaves for k_solve
  OuterState* states_dev = nullptr;    // [n_slots]
  OuterState* states_host = nullptr;   // pinned: upload staging at admission / download target when the host looks
  OuterState* states_init = nullptr;   // pinned: initial states (separate from the download target: uploads and downloads overlap)
  double* chunks_dev = nullptr;        // [n_slots][FINAL_CHUNKS * MOM_ROW]
  hipEvent_t group_ev[MAX_GROUPS] = {};
  // batch mode: one workspace per scheduler slot.  They live here (not in a thread-local) so that they are tied to this
  // context's device, reused by every thread that drives the context, and released by lh_destroy.
  std::vector<Workspace> slot_ws;
  // misc pinned scratch for small downloads
  double* small_host = nullptr;
  size_t small_host_doubles = 0;
  // source-sharded single pair: in-place sum of the cost/moment sums over the ranks that hold the other shards
  lh_allreduce_fn reduce_fn = nullptr;
  void* reduce_user = nullptr;
  uint64_t epoch = 0;          // one per scheduler run (run_tasks_*): see lh_cloud::built_epoch
  // profiling
  bool prof = false;
  std::vector<ProfEntry> prof_entries;
  std::vector<ProfPending> prof_pending;
  std::vector<hipEvent_t> ev_pool;

  int prof_entry(const char* name) {
    for (size_t i = 0; i < prof_entries.size(); i++)
      if (prof_entries[i].name == name) return (int)i;
    ProfEntry e;
    e.name = name;
    prof_entries.push_back(e);
    return (int)prof_entries.size() - 1;
  }
  hipEvent_t get_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
  void prof_flush() {
    if (prof_pending.empty()) return;
    for (auto& p : prof_pending) {
      float ms = 0;
      (void)hipEventSynchronize(p.b);
      (void)hipEventElapsedTime(&ms, p.a, p.b);
      prof_entries[p.entry].ms += ms;
      static FILE* plog = []() { const char* e = getenv("LH_PROF_LOG"); return e ? fopen(e, "a") : (FILE*)nullptr; }();  // per-launch trace (debug)
      if (plog) { fprintf(plog, "%s %.4f\n", prof_entries[p.entry].name.c_str(), ms); fflush(plog); }
      ev_pool.push_back(p.a);
      ev_pool.push_back(p.b);
    }
    prof_pending.clear();
  }
};

// RAII-ish profiling scope around one launch (HIP events on the context's own stream)
struct ProfScope {
  lh_ctx* c; int entry = -1; hipEvent_t a, b;
  hipStream_t st;
  ProfScope(lh_ctx* ctx, const char* name, double bytes, hipStream_t stream = nullptr) : c(ctx) {
    if (!c->prof) return;
    st = stream ? stream : c->stream;
    entry = c->prof_entry(name);
    c->prof_entries[entry].launches++;
    c->prof_entries[entry].bytes += bytes;
    a = c->get_event(); b = c->get_event();
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (entry < 0) return;
    (void)hipEventRecord(b, st);
    c->prof_pending.push_back({entry, a, b});
    if (c->prof_pending.size() > 8192) c->prof_flush();
  }
};

struct lh_cloud {
  lh_ctx* ctx = nullptr;
  int n = 0, n_pad = 0;
  float4* xyz = nullptr;
  float4* nrm = nullptr;       // null if the cloud has no normals
  float* intensity = nullptr;  // null if none
  // NN index
  bool has_index = false;
  float4* sorted = nullptr;    // [n + LEAF_CAP]
  NodeX* node_buf = nullptr;   // element 0 holds the TreeHeader, the nodes start at element 1
  int index_cap = 0;           // points the index buffers were allocated for
  NodeX* nodes() const { return node_buf ? node_buf + 1 : nullptr; }
  TreeHeader* hdr() const { return reinterpret_cast<TreeHeader*>(node_buf); }
  // k-NN covariances (6 planes of n_pad doubles), valid for (cov_k, cov_eps)
  double* cov6 = nullptr;
  int cov_k = 0;
  double cov_eps = 0;
  uint64_t built_epoch = 0;    // the batch call (lh_ctx::epoch) that last built this cloud's index: a target shared by pairs of several
                               // scheduler groups is rebuilt ONCE per call, not once per group admission under the other groups' sweeps
  TreeView view() const { return TreeView{sorted, nodes(), hdr(), n}; }
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static void cloud_free(lh_cloud* c) {
  if (!c) return;
  (void)lhFree(c->xyz); (void)lhFree(c->nrm); (void)lhFree(c->intensity);
  (void)lhFree(c->sorted); (void)lhFree(c->node_buf); (void)lhFree(c->cov6);
  delete c;
}

// Scope guard of one entry point: temporary device buffers and a cloud under construction are handed back on EVERY exit
// path (the HIPCHK early returns included).  lhFree only parks a block, and the pool hands it out again in stream order, so
// freeing while the call's own kernels are still queued is safe.
struct DevGuard {
  std::vector<void*> bufs;
  lh_cloud* cloud = nullptr;
  template <class T>
  hipError_t alloc(T** p, size_t bytes) {
    hipError_t e = lhMalloc(p, bytes);
    if (e == hipSuccess) bufs.push_back(*p);
    return e;
  }
  lh_cloud* keep_cloud() { lh_cloud* c = cloud; cloud = nullptr; return c; }
  ~DevGuard() {
    for (void* b : bufs) (void)lhFree(b);
    if (cloud) cloud_free(cloud);
  }
};

static lh_status ctx_ensure_scratch(lh_ctx* c, int n) {
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

static lh_status ctx_ensure_small(lh_ctx* c, size_t doubles) {
  if (doubles <= c->small_host_doubles) return LH_OK;
  (void)hipStreamSynchronize(c->stream);
  if (c->small_host) (void)hipHostFree(c->small_host);
  size_t cap = std::max<size_t>(doubles, 4096);
  HIPCHK(hipHostMalloc(&c->small_host, sizeof(double) * cap, hipHostMallocDefault));
  c->small_host_doubles = cap;
  return LH_OK;
}

// K2: Hilbert sort + cell-aligned radix tree with 4-ary nodes (replaces tree_->setInputCloud of pcl::Registration::initCompute).
// All clouds of a batch are built by the same launches, one radix sort and one scan (see lh_kernels.hpp "K2 batched").
static lh_status build_indices(lh_ctx* x, lh_cloud* const* clouds, int n_clouds, hipStream_t s_in = nullptr) {
  if (n_clouds <= 0) return LH_OK;
  hipStream_t s = s_in ? s_in : x->stream;
  for (int o = 0; o < n_clouds; o += MAX_INDEX_BATCH) {
    int nb = std::min(MAX_INDEX_BATCH, n_clouds - o);
    long total = 0;
    int max_n = 0, tile0 = 0;
    if (!x->idx_descs_dev) {
      HIPCHK(hipMalloc(&x->idx_descs_dev, sizeof(IndexDesc) * MAX_INDEX_BATCH));
      HIPCHK(hipHostMalloc(&x->idx_descs_host, sizeof(IndexDesc) * MAX_INDEX_BATCH * lh_ctx::IDX_STAGE, hipHostMallocDefault));
      HIPCHK(hipMalloc(&x->idx_bbox, sizeof(uint32_t) * 8 * MAX_INDEX_BATCH));
      launch_index_bbox_init(x->idx_bbox, s);   // (every build then leaves the slots reset for the next one)
      for (int k = 0; k < lh_ctx::IDX_STAGE; k++) HIPCHK(hipEventCreateWithFlags(&x->idx_copy_done[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&x->idx_build_done, hipEventDisableTiming));
    }
    // The descriptors are staged in a ring: the upload of a build is queued behind the previous build on the GPU (shared scratch), so
    // waiting for the PREVIOUS upload before refilling one staging buffer tied the scheduling thread to the GPU's index builds
    // (1.6 ms per group of 32 with sixteen groups in flight: 25 of a 60-ms step).  Only the upload IDX_STAGE builds ago is waited for.
    const int stage = x->idx_stage;
    x->idx_stage = (x->idx_stage + 1) % lh_ctx::IDX_STAGE;
    IndexDesc* const stage_host = x->idx_descs_host + (size_t)stage * MAX_INDEX_BATCH;
    HIPCHK(hipEventSynchronize(x->idx_copy_done[stage]));   // (an event that was never recorded is complete)
    for (int k = 0; k < nb; k++) {
      lh_cloud* c = clouds[o + k];
      if (!c || c->n <= 0 || c->ctx != x) return LH_EINVAL;
      if (c->n > (1 << 27)) return LH_EINVAL;  // leaf references keep 27 bits of sorted position
      if (c->n > c->index_cap) {
        (void)hipStreamSynchronize(x->stream);
        x->sync_side_streams();
        (void)lhFree(c->sorted); (void)lhFree(c->node_buf);
        c->sorted = nullptr; c->node_buf = nullptr; c->index_cap = 0;
        HIPCHK(lhMalloc(&c->sorted, sizeof(float4) * ((size_t)c->n + LEAF_CAP)));
        HIPCHK(lhMalloc(&c->node_buf, sizeof(NodeX) * ((size_t)c->n + 1)));  // worst case: every point its own leaf
        c->index_cap = c->n;
      }
      IndexDesc& d = stage_host[k];
      d.xyz = c->xyz; d.sorted = c->sorted; d.nodes = c->nodes(); d.hdr = c->hdr(); d.pos = nullptr;
      d.n = c->n; d.offset = (int)total;
      d.tile0 = tile0; d.pad = 0;
      tile0 += segsort_tiles(c->n);
      total += c->n;
      max_n = std::max(max_n, c->n);
    }
    if (total > 0x7ffffff0L) return LH_EINVAL;
    if ((int)total > x->idx_cap) {
      (void)hipStreamSynchronize(x->stream);
      x->sync_side_streams();
      (void)lhFree(x->k64a); (void)lhFree(x->k64b); (void)lhFree(x->v32a); (void)lhFree(x->v32b); (void)lhFree(x->sort64_temp);
      (void)lhFree(x->tree_tmp); (void)lhFree(x->scan_tmp); (void)lhFree(x->k32a); (void)lhFree(x->k32b); (void)lhFree(x->rs_hist);
      int cap = round_up((int)std::min<long>(total + total / 4, 0x7fffff00L), 1024);
      HIPCHK(hipMalloc(&x->k64a, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&x->k64b, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&x->v32a, sizeof(uint32_t) * (size_t)cap));
      HIPCHK(hipMalloc(&x->v32b, sizeof(uint32_t) * (size_t)cap));
      HIPCHK(hipMalloc(&x->k32a, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&x->k32b, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&x->rs_hist, sizeof(uint32_t) * segsort_hist_elems(cap, MAX_INDEX_BATCH)));
      x->sort64_temp_bytes = sort64_temp_bytes(cap);
      HIPCHK(hipMalloc(&x->sort64_temp, x->sort64_temp_bytes ? x->sort64_temp_bytes : 16));
      HIPCHK(hipMalloc(&x->tree_tmp, TREE_SCRATCH_BYTES_PER_POINT * ((size_t)cap + 16) + 4096));
      HIPCHK(hipMemsetAsync(x->tree_tmp, 0, TREE_SCRATCH_BYTES_PER_POINT * ((size_t)cap + 16) + 4096, s));   // (the per-tile leaf counts must start at zero; every build leaves them so)
      x->idx_cap = cap;
    }
    TreeScratch ts;
    {
      size_t cap = (size_t)x->idx_cap + 16;
      char* p = x->tree_tmp;
      ts.lkey = reinterpret_cast<uint64_t*>(p); p += 8 * cap;       // 16-byte aligned arrays first (cap is a multiple of 16)
      ts.lbox = reinterpret_cast<float4*>(p); p += 32 * cap;
      ts.a1box = reinterpret_cast<float4*>(p); p += 32 * (cap / 32 + 16);
      ts.a2box = reinterpret_cast<float4*>(p); p += 32 * (cap / 1024 + 16);
      ts.ibox = reinterpret_cast<float4*>(p); p += 32 * cap;
      ts.ichild = reinterpret_cast<int32_t*>(p); p += 8 * cap;
      ts.irange = reinterpret_cast<int32_t*>(p); p += 8 * cap;
      ts.iparent = reinterpret_cast<int32_t*>(p); p += 4 * cap;
      ts.flag = reinterpret_cast<uint32_t*>(p); p += 4 * cap;
      ts.lid = reinterpret_cast<uint32_t*>(p); p += 4 * cap;
      ts.lstart = reinterpret_cast<uint32_t*>(p); p += 4 * cap;
      ts.tsum = reinterpret_cast<uint32_t*>(p); p += 4 * (cap / 4096 + 16);
      ts.toff = reinterpret_cast<uint32_t*>(p); p += 4 * (cap / 4096 + 16);
      ts.keys = x->k64b;
      ts.total = (int)total;
    }
    HIPCHK(hipStreamWaitEvent(s, x->idx_build_done, 0));  // the shared build scratch may still be in use on the other stream
    HIPCHK(hipMemcpyAsync(x->idx_descs_dev, stage_host, sizeof(IndexDesc) * nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(x->idx_copy_done[stage], s));
    int id_bits = 0;
    while ((1 << id_bits) < nb) id_bits++;
    // LH_SORT=generic: the one-segment 64-bit sort over the whole concatenated array instead of the segmented one (A/B);
    // LH_SORT=check: both, compared element by element (tests: two independent code paths must give the same stable order)
    static const int sort_cfg = []() { const char* e = getenv("LH_SORT"); return !e ? 0 : (strcmp(e, "generic") == 0 ? 1 : (strcmp(e, "check") == 0 ? 2 : 0)); }();
    { ProfScope p(x, "index_bbox_keys", 16.0 * total * 2, s);
      launch_index_keys(x->idx_descs_dev, nb, max_n, x->idx_bbox, x->k32a, sort_cfg ? x->k64a : nullptr, sort_cfg ? x->v32a : nullptr, s); }
    {
      if (sort_cfg == 1) {
        ProfScope p(x, "index_radix_sort", 12.0 * total * 2 * 4, s);
        sort_pairs_u64(x->sort64_temp, x->sort64_temp_bytes, x->k64a, x->k64b, x->v32a, x->v32b, (int)total, 32 + id_bits, s);
      } else {
        std::vector<uint64_t> kref;
        std::vector<uint32_t> vref;
        if (sort_cfg == 2) {  // reference first
          sort_pairs_u64(x->sort64_temp, x->sort64_temp_bytes, x->k64a, x->k64b, x->v32a, x->v32b, (int)total, 32 + id_bits, s);
          kref.resize(total); vref.resize(total);
          HIPCHK(hipMemcpyAsync(kref.data(), x->k64b, sizeof(uint64_t) * total, hipMemcpyDeviceToHost, s));
          HIPCHK(hipMemcpyAsync(vref.data(), x->v32b, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
        }
        { ProfScope p(x, "index_radix_sort", 8.0 * total * 3 * 2, s);
          segsort_pairs(x->idx_descs_dev, nb, max_n, x->k32a, x->k32b, x->k64b, x->v32b, x->rs_hist, s); }
        if (sort_cfg == 2) {
          std::vector<uint64_t> kk(total);
          std::vector<uint32_t> vv(total);
          HIPCHK(hipMemcpyAsync(kk.data(), x->k64b, sizeof(uint64_t) * total, hipMemcpyDeviceToHost, s));
          HIPCHK(hipMemcpyAsync(vv.data(), x->v32b, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
          long bad = 0;
          for (long i = 0; i < total; i++)
            if (kk[i] != kref[i] || vv[i] != vref[i]) bad++;
          if (bad) {
            fprintf(stderr, "[locus_hip] LH_SORT=check: %ld of %ld sorted elements differ between the segmented and the one-segment sort\n", bad, total);
            return LH_EDEVICE;
          }
        }
      }
    }
    { ProfScope p(x, "index_leaves", 8.0 * total * 3 + 48.0 * total, s); launch_index_leaves(x->idx_descs_dev, nb, ts, x->v32b, x->idx_bbox, s); }
    { ProfScope p(x, "index_box_tables", 24.0 * total, s); launch_index_trees(x->idx_descs_dev, nb, max_n, ts, s, 0); }
    { ProfScope p(x, "index_radix_tree", 8.0 * total, s); launch_index_trees(x->idx_descs_dev, nb, max_n, ts, s, 1); }
    { ProfScope p(x, "index_nodes", 32.0 * total, s); launch_index_trees(x->idx_descs_dev, nb, max_n, ts, s, 2); }
    HIPCHK(hipEventRecord(x->idx_build_done, s));
    HIPCHK(hipGetLastError());
    for (int k = 0; k < nb; k++) clouds[o + k]->has_index = true;
  }
  return LH_OK;
}
static lh_status cloud_build_index(lh_cloud* c) { return build_indices(c->ctx, &c, 1); }

static lh_status cloud_ensure_cov(lh_cloud* c, int k, double eps) {
  if (c->cov6 && c->cov_k == k && c->cov_eps == eps) return LH_OK;
  if (k > c->n || k > 64 || k < 1) return LH_EINVAL;  // gicp.hpp:72-79
  if (!c->has_index) { lh_status st = cloud_build_index(c); if (st) return st; }
  if (!c->cov6) HIPCHK(lhMalloc(&c->cov6, sizeof(double) * 6 * (size_t)c->n_pad));
  { ProfScope p(c->ctx, "knn_cov", (16.0 + 20 * 16.0 + 48.0) * c->n); launch_knn_cov(c->xyz, c->n, c->n_pad, c->view(), k, eps, c->cov6, c->ctx->stream); }
  HIPCHK(hipGetLastError());
  c->cov_k = k;
  c->cov_eps = eps;
  return LH_OK;
}

// ---------------------------------------------------------------------------------------------------------
lh_status Workspace::ensure(lh_ctx* c, int n) {
  if (n <= cap) return LH_OK;
  (void)hipStreamSynchronize(c->stream);  // a slot may belong to either scheduler group: nothing may still use the old buffers
  c->sync_side_streams();
  (void)lhFree(corr); (void)lhFree(maha6); (void)lhFree(prev_nn); (void)lhFree(out_xyz); (void)lhFree(cert); (void)lhFree(rec);
  corr = nullptr; maha6 = nullptr; prev_nn = nullptr; out_xyz = nullptr; cert = nullptr; rec = nullptr; cap = 0;
  int ncap = round_up(n, 256);
  HIPCHK(hipMalloc(&corr, sizeof(float4) * (size_t)ncap));
  HIPCHK(hipMalloc(&maha6, sizeof(double) * 6 * (size_t)ncap));
  HIPCHK(hipMalloc(&prev_nn, sizeof(int32_t) * (size_t)ncap));
  HIPCHK(hipMalloc(&out_xyz, sizeof(float4) * (size_t)ncap));
  HIPCHK(hipMalloc(&cert, sizeof(float4) * (size_t)ncap));
  HIPCHK(hipMalloc(&rec, sizeof(float4) * 2 * (size_t)ncap));
  if (!stats) { HIPCHK(hipMalloc(&stats, 16)); HIPCHK(hipMemset(stats, 0, 16)); }
  cap = ncap;
  n_pad = ncap;
  return LH_OK;
}
void Workspace::release() {
  (void)lhFree(corr); (void)lhFree(maha6); (void)lhFree(prev_nn); (void)lhFree(out_xyz); (void)lhFree(cert); (void)lhFree(rec); (void)lhFree(stats);
  corr = nullptr; maha6 = nullptr; prev_nn = nullptr; out_xyz = nullptr; cert = nullptr; rec = nullptr; stats = nullptr; cap = 0;
}

static lh_status ctx_ensure_slots(lh_ctx* c, int n_slots, int max_n) {
  size_t per_slot = std::max<size_t>((size_t)cost_blocks(max_n) * COST_NSUM, (size_t)FINAL_CHUNKS * MOM_ROW);
  int mom_stride = sweep_rows(max_n) * MOM_ROW;  // one partial row per 256-point workgroup of the sweep + the walk rows
  int mask_stride = ((max_n + 255) / 256) * 4;
  if (n_slots <= c->n_slots && per_slot <= c->partials_per_slot && mom_stride <= c->mom_stride) return LH_OK;
  (void)hipStreamSynchronize(c->stream);
  c->sync_side_streams();
  n_slots = std::max(n_slots, c->n_slots);
  per_slot = std::max(per_slot, c->partials_per_slot);
  mom_stride = std::max(mom_stride, c->mom_stride);
  mask_stride = std::max(mask_stride, c->mask_stride);
  (void)lhFree(c->descs_dev);
  (void)lhFree(c->mom_partials_dev);
  (void)lhFree(c->wmask_dev);
  (void)lhFree(c->states_dev);
  (void)lhFree(c->chunks_dev);
  if (c->descs_host) (void)hipHostFree(c->descs_host);
  if (c->partials_host) (void)hipHostFree(c->partials_host);
  if (c->states_host) (void)hipHostFree(c->states_host);
  if (c->states_init) (void)hipHostFree(c->states_init);
  HIPCHK(hipMalloc(&c->descs_dev, sizeof(PairDesc) * n_slots));
  HIPCHK(hipMalloc(&c->mom_partials_dev, sizeof(double) * (size_t)mom_stride * n_slots));
  HIPCHK(hipMalloc(&c->wmask_dev, sizeof(unsigned long long) * (size_t)mask_stride * n_slots));
  HIPCHK(hipMemset(c->wmask_dev, 0, sizeof(unsigned long long) * (size_t)mask_stride * n_slots));
  HIPCHK(hipMalloc(&c->states_dev, sizeof(OuterState) * n_slots));
  HIPCHK(hipMalloc(&c->chunks_dev, sizeof(double) * (size_t)FINAL_CHUNKS * MOM_ROW * n_slots));
  HIPCHK(hipHostMalloc(&c->states_host, sizeof(OuterState) * n_slots, hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&c->states_init, sizeof(OuterState) * n_slots, hipHostMallocDefault));
  for (int k = 0; k < lh_ctx::MAX_GROUPS; k++)
    if (!c->group_ev[k]) HIPCHK(hipEventCreateWithFlags(&c->group_ev[k], hipEventDisableTiming));
  c->mom_stride = mom_stride;
  c->mask_stride = mask_stride;
  HIPCHK(hipHostMalloc(&c->descs_host, sizeof(PairDesc) * n_slots, hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&c->partials_host, sizeof(double) * per_slot * n_slots, hipHostMallocDefault));
  c->n_slots = n_slots;
  c->partials_per_slot = per_slot;
  return LH_OK;
}

// ---------------------------------------------------------------------------------------------------------
// one alignment = one coroutine
enum Req { REQ_NONE = 0, REQ_SWEEP, REQ_COST, REQ_DONE };

struct Task;
static thread_local Task* g_boot_task = nullptr;

struct Task {
  // inputs
  lh_gicp_params P;
  lh_cloud *src = nullptr, *tgt = nullptr;
  Workspace* ws = nullptr;
  float guess[16];
  bool guess_is_identity = true;
  int slot = 0;
  hipStream_t stream = nullptr;  // the stream of the scheduler group that owns the task
  lh_gicp_trace* trace = nullptr;
  lh_cloud* aligned = nullptr;   // batch API: receives final_transformation_ * input (gicp.hpp:586) when the pair retires
  // coroutine (host-driven loop)
  ucontext_t ctx, sched;
  std::vector<char> stack;
  Req req = REQ_NONE;
  float req_T12[12];
  double res_sums[COST_NSUM];
  MomentModel mom;  // cost_mode 1 on the host: filled by the scheduler after each sweep
  bool sweep_bytes_pending = false;
  bool count_stats = false;  // debug sweeps only
  bool first_sweep = true;   // cold: gets a seed pre-pass
  long last_walks = -1;      // tree walks of the previous fused sweep (instrumentation: how many certificates failed)
  // device-driven loop
  lh_gicp_trace* trace_dev = nullptr;
  int enq_iters = 0;         // outer iterations enqueued so far
  int sweeps_done = 0;       // host-driven loop: sweeps launched so far (the device-driven loop counts enq_iters)
  // the loop's state (host-driven: advanced by run(); device-driven: the last download of the pair's device state)
  OuterState os;
  // outputs
  lh_gicp_result result;

  void yield(Req r) {
    req = r;
    swapcontext(&ctx, &sched);
  }
  void resume() {
    swapcontext(&sched, &ctx);
  }
  static void T16_to_T12(const float* T16, float* T12) {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) T12[r * 4 + c] = T16[c * 4 + r];
  }
  // cost_mode 0: one fused device pass per evaluation (gicp.hpp:362-402), reference arithmetic; libm on the host like the oracle
  struct DevicePass {
    Task* t;
    void operator()(const double x[6], const Trig& tg, double sums13[13], double* count) {
      float T16[16];
      apply_state_trig(x, tg, T16);   // base_transformation_ = I (gicp.hpp:435, 367-368)
      T16_to_T12(T16, t->req_T12);
      t->yield(REQ_COST);
      memcpy(sums13, t->res_sums, sizeof(double) * 13);
      *count = t->res_sums[13];
    }
  };

  // final_transformation_ = previous_transformation_ * guess (gicp.hpp:583), float; result fields from the loop state
  void finish_result() {
    memset(&result, 0, sizeof(result));
    result.fitness = NAN;
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) {
        float sm = 0.0f;
        for (int k = 0; k < 4; k++) sm += os.prev[k * 4 + r] * guess[c * 4 + k];
        result.T[c * 4 + r] = sm;
      }
    result.converged = os.converged;
    result.iterations = os.iter;
    result.n_correspondences_last = os.n_corr_last;
    result.cost_passes = os.passes;
    result.status = os.status == 0 ? LH_OK : (os.status == -4 ? LH_ETOO_FEW_CORR : (os.status == -6 ? LH_ENO_NN : LH_ESOLVER));  // the exception the reference caught (gicp.hpp:542-547)
    if (os.status == -6) {  // a source point without a nearest neighbour: computeTransformation returned at gicp.hpp:504-506 and
      memcpy(result.T, I16, sizeof(I16));   // final_transformation_ is still what pcl::Registration::align reset it to, converged_ false
      result.converged = 0;
      result.n_correspondences_last = 0;
    }
  }

  // computeTransformation (gicp.hpp:406-617), host-driven; covariances / index were prepared by the caller
  void run() {
    outer_state_init(&os);  // pcl::Registration::align resets transformation_ to identity
    if (trace) trace->n_iters = 0;
    const OuterParams OP{P.max_iterations, P.max_inner_iterations, P.rotation_epsilon, P.transformation_epsilon};
    while (!os.done) {
      T16_to_T12(os.T, req_T12);
      yield(REQ_SWEEP);                                   // gicp.hpp:464-498 (transform_R is formed in the kernel from T and the guess)
      const int before = os.passes, it = os.iter;
      double k_t;
      if (P.cost_mode == 1) {  // every evaluation of this outer iteration comes from the 74 moments of the sweep: no device pass
        typedef MomentPass<PortableMath> Pass;
        typedef CostEval<Pass, PortableMath> Fn;
        Pass pass{&mom};
        Fn fn;               // new correspondences: a fresh functor cache
        fn.pass = pass;
        outer_step<Fn, PortableMath>(&fn, OP, &os);
        k_t = mom.count();
      } else {
        typedef CostEval<DevicePass, LibmMath> Fn;
        DevicePass pass{this};
        Fn fn;
        fn.pass = pass;
        outer_step<Fn, LibmMath>(&fn, OP, &os);
        k_t = fn.count();
      }
      os.corr_sum += k_t;
      if (trace && os.status == 0 && it < LH_MAX_TRACE) {
        memcpy(trace->T[it], os.T, sizeof(os.T));
        trace->n_corr[it] = os.n_corr_last;
        trace->n_passes[it] = os.passes - before;
        trace->n_inner[it] = os.n_inner;
        trace->f_end[it] = os.f_end;
        trace->delta[it] = os.delta;
        trace->n_iters = it + 1;
      }
    }
    finish_result();
    yield(REQ_DONE);
  }

  static void entry() {
    Task* t = g_boot_task;
    t->run();
    for (;;) t->yield(REQ_DONE);
  }
  void start() {
    stack.resize(256 * 1024);
    getcontext(&ctx);
    ctx.uc_stack.ss_sp = stack.data();
    ctx.uc_stack.ss_size = stack.size();
    ctx.uc_link = &sched;
    makecontext(&ctx, (void (*)())entry, 0);
    g_boot_task = this;
    req = REQ_NONE;
    first_sweep = true;
    sweeps_done = 0;
    resume();  // runs until the first request
  }
};

// prepare device state of one pair in its slot: index, covariances, output cloud, descriptor
static lh_status task_prepare(lh_ctx* c, Task* t, bool rebuild_index, bool upload_desc = true) {
  lh_cloud *src = t->src, *tgt = t->tgt;
  if (!src || !tgt || src->n <= 0 || tgt->n <= 0) return LH_EINVAL;
  const lh_gicp_params& P = t->P;
  if (!P.recompute_source_cov && !src->nrm) return LH_EINVAL;
  if (!P.recompute_target_cov && !tgt->nrm) return LH_EINVAL;
  lh_status st;
  if (rebuild_index || !tgt->has_index) { st = cloud_build_index(tgt); if (st) return st; }
  if (P.recompute_target_cov) { st = cloud_ensure_cov(tgt, P.k_correspondences, P.gicp_epsilon); if (st) return st; }
  if (P.recompute_source_cov) { st = cloud_ensure_cov(src, P.k_correspondences, P.gicp_epsilon); if (st) return st; }
  st = t->ws->ensure(c, src->n);
  if (st) return st;
  hipStream_t ts = t->stream ? t->stream : c->stream;
  t->guess_is_identity = memcmp(t->guess, I16, sizeof(I16)) == 0;
  const float4* out = src->xyz;
  if (!t->guess_is_identity) {  // pcl::transformPointCloud(output, output, guess) (gicp.hpp:440)
    float T12[12];
    Task::T16_to_T12(t->guess, T12);
    ProfScope p(c, "transform", 32.0 * src->n, ts);
    launch_transform(src->xyz, nullptr, src->n, T12, t->ws->out_xyz, nullptr, ts);
    out = t->ws->out_xyz;
  }
  if (t->count_stats) {  // debug sweeps run without the seed pre-pass: start from "no candidate"
    ProfScope p(c, "fill", 4.0 * src->n);
    launch_fill_i32(t->ws->prev_nn, src->n, -1, ts);
  }
  PairDesc& d = c->descs_host[t->slot];
  d.src = out;
  d.src_nrm = P.recompute_source_cov ? nullptr : src->nrm;
  d.src_cov6 = P.recompute_source_cov ? src->cov6 : nullptr;
  d.tgt_xyz = tgt->xyz;
  d.tgt_nrm = P.recompute_target_cov ? nullptr : tgt->nrm;
  d.tgt_cov6 = P.recompute_target_cov ? tgt->cov6 : nullptr;
  d.tgt_sorted = tgt->sorted;
  d.tgt_nodes = tgt->nodes();
  d.tgt_hdr = tgt->hdr();
  d.prev_nn = t->ws->prev_nn;
  d.cert = t->ws->cert;
  d.rec = t->ws->rec;
  d.stats = t->count_stats ? t->ws->stats : nullptr;
  d.corr = t->ws->corr;
  d.maha6 = t->ws->maha6;
  d.n = src->n;
  d.n_pad = t->ws->n_pad;
  d.m = tgt->n;
  d.m_pad = tgt->n_pad;
  d.src_cov_pad = src->n_pad;
  d.corr_dist2 = P.corr_dist * P.corr_dist;  // gicp.hpp:438
  d.gicp_eps = P.gicp_epsilon;
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) d.guess3[r * 3 + cc] = (double)t->guess[cc * 4 + r];
  d.guess_identity = 1;
  for (int k = 0; k < 9; k++)
    if (d.guess3[k] != ((k % 4 == 0) ? 1.0 : 0.0)) d.guess_identity = 0;
  d.max_iterations = P.max_iterations;
  d.max_inner_iterations = P.max_inner_iterations;
  d.rotation_epsilon = P.rotation_epsilon;
  d.transformation_epsilon = P.transformation_epsilon;
  d.trace = t->trace_dev;
  if (upload_desc) HIPCHK(hipMemcpyAsync(&c->descs_dev[t->slot], &d, sizeof(PairDesc), hipMemcpyHostToDevice, ts));   // (the device-driven scheduler uploads a group's descriptors in one copy)
  HIPCHK(hipGetLastError());
  return LH_OK;
}

// ---- scheduler ------------------------------------------------------------------------------------------------
// In-flight pairs are split into groups (two half-batches when >= 16 pairs are in flight, each with its own HIP
// stream): while the host delivers results / runs the BFGS solves of one group, the other group's kernels keep the
// GPU busy.  Slots, descriptors, partial-sum buffers are per slot, so groups never share mutable device state; the
// index-build scratch is shared and ordered across streams by an event.
struct Group {
  hipStream_t stream = nullptr;
  std::vector<Task*> active, sweeps, moms, costs;
  std::vector<int> free_slots;
  bool inflight = false;
};

// Is the pair's k-th sweep (0-based) launched in the two-launch form (k_late + k_walk)?  A fixed rule of the pair's own
// parameters and k, so that both loop flavours, any batching and any number of GPUs add the same partial rows in the same order.
static bool sweep_is_split(const Task* t, int k) {
  return t->P.cost_mode == 1 && !t->P.recompute_source_cov && !t->P.recompute_target_cov && t->guess_is_identity && t->src->nrm &&
         t->tgt->nrm && t->ws->rec && k >= sweep_split_from();
}

static lh_status group_launch(lh_ctx* c, Group& g) {
  hipStream_t st = g.stream;
  g.sweeps.clear(); g.moms.clear(); g.costs.clear();
  // phase 1: sweeps (+ seed pre-pass for cold pairs).  Sweeps are held until every pair of the group has finished its
  // BFGS solve (cost_mode 0 pairs need different numbers of cost passes), so they always go out as ONE wide launch.
  bool any_cost = false;
  for (Task* t : g.active)
    if (t->req == REQ_COST) any_cost = true;
  if (!any_cost)
    for (Task* t : g.active)
      if (t->req == REQ_SWEEP) g.sweeps.push_back(t);
  for (size_t o = 0; o < g.sweeps.size(); o += MAX_JOBS) {
    SweepArgs a;
    a.njobs = (int)std::min<size_t>(MAX_JOBS, g.sweeps.size() - o);
    a.bpj = 0;
    a.max_depth = 0;
    a.pad = 0;
    int max_n = 0;
    double bytes = 0;
    for (int j = 0; j < a.njobs; j++) {
      Task* t = g.sweeps[o + j];
      a.job[j].slot = t->slot;
      a.job[j].pad = 0;
      memcpy(a.job[j].T, t->req_T12, sizeof(t->req_T12));
      max_n = std::max(max_n, t->src->n);
      bytes += 20.0 * t->src->n;  // SURVEY 8d B_nn = 20 N + 232 K_t; the K_t term is added when the count is known
      t->sweep_bytes_pending = true;
    }
    {  // cold tasks (first sweep of a pair): seed pre-pass so the sweep starts warm
      SweepArgs sa;
      sa.njobs = 0;
      sa.max_depth = a.max_depth;
      sa.pad = 0;
      int smax = 0;
      for (int j = 0; j < a.njobs; j++) {
        Task* t = g.sweeps[o + j];
        if (t->first_sweep) {
          sa.job[sa.njobs++] = a.job[j];
          smax = std::max(smax, t->src->n);
          t->first_sweep = false;
        }
      }
      if (sa.njobs > 0) {
        ProfScope p(c, "nn_seed", 0.0, st);
        launch_seed(c->descs_dev, sa, smax, st);
      }
    }
    bool all_fused = true;
    for (int j = 0; j < a.njobs; j++)
      if (g.sweeps[o + j]->P.cost_mode != 1) all_fused = false;
    if (all_fused) {
      // cost_mode 1: sweep and moment reduction in ONE kernel; M and the correspondences never reach HBM
      CostArgs ca;
      ca.njobs = a.njobs;
      ca.pad = 0;
      for (int j = 0; j < a.njobs; j++) {
        Task* t = g.sweeps[o + j];
        ca.job[j].slot = t->slot;
        ca.job[j].out_offset = (int)((size_t)t->slot * c->partials_per_slot);
        memcpy(ca.job[j].T, t->req_T12, sizeof(t->req_T12));
        g.moms.push_back(t);
      }
      ProfScope p(c, "nn_sweep", bytes, st);
      bool normals_only = true;
      uint32_t split_mask = 0u;
      for (int j = 0; j < a.njobs; j++) {
        Task* t = g.sweeps[o + j];
        if (t->P.recompute_source_cov || t->P.recompute_target_cov) normals_only = false;
        if (sweep_is_split(t, t->sweeps_done)) split_mask |= 1u << j;
        t->sweeps_done++;
      }
      launch_sweep_fused(c->descs_dev, a, split_mask, max_n, c->mom_partials_dev, c->mom_stride, nullptr, normals_only, c->wmask_dev, c->mask_stride, st);
      launch_moments_final(c->descs_dev, ca, c->mom_partials_dev, c->mom_stride, c->partials_host, nullptr, c->wmask_dev, c->mask_stride, st);
    } else {
      {
        ProfScope p(c, "nn_sweep", bytes, st);
        launch_sweep(c->descs_dev, a, max_n, st);
      }
      // mixed batch: cost_mode 1 pairs get a separate moment pass over the stored correspondences
      CostArgs ca;
      ca.njobs = 0;
      ca.pad = 0;
      int mmax = 0;
      for (int j = 0; j < a.njobs; j++) {
        Task* t = g.sweeps[o + j];
        if (t->P.cost_mode != 1) continue;
        ca.job[ca.njobs].slot = t->slot;
        ca.job[ca.njobs].out_offset = (int)((size_t)t->slot * c->partials_per_slot);
        memcpy(ca.job[ca.njobs].T, t->req_T12, sizeof(t->req_T12));
        ca.njobs++;
        mmax = std::max(mmax, t->src->n);
        g.moms.push_back(t);
      }
      if (ca.njobs > 0) {
        ProfScope p(c, "cost_moments", 0.0, st);
        launch_moments(c->descs_dev, ca, mmax, c->mom_partials_dev, c->mom_stride, c->partials_host, st);
      }
    }
  }
  for (Task* t : g.sweeps)
    if (t->P.cost_mode != 1) t->resume();  // each now yields its first COST request (or DONE)
  // phase 2: per-evaluation cost passes (cost_mode 0)
  for (Task* t : g.active)
    if (t->req == REQ_COST) g.costs.push_back(t);
  for (size_t o = 0; o < g.costs.size(); o += MAX_JOBS) {
    CostArgs a;
    a.njobs = (int)std::min<size_t>(MAX_JOBS, g.costs.size() - o);
    a.pad = 0;
    int max_n = 0;
    for (int j = 0; j < a.njobs; j++) {
      Task* t = g.costs[o + j];
      a.job[j].slot = t->slot;
      a.job[j].out_offset = (int)((size_t)t->slot * c->partials_per_slot);
      memcpy(a.job[j].T, t->req_T12, sizeof(t->req_T12));
      max_n = std::max(max_n, t->src->n);
    }
    ProfScope p(c, "cost_fdf", 0.0, st);
    launch_cost(c->descs_dev, a, max_n, c->mom_partials_dev, c->mom_stride, c->partials_host, st);
  }
  HIPCHK(hipGetLastError());
  g.inflight = !g.costs.empty() || !g.sweeps.empty();
  return LH_OK;
}

static lh_status group_collect(lh_ctx* c, Group& g) {
  if (g.inflight) HIPCHK(hipStreamSynchronize(g.stream));
  g.inflight = false;
  for (Task* t : g.costs) {
    const double* part = c->partials_host + (size_t)t->slot * c->partials_per_slot;   // the 14 sums, added in block order by k_cost_final
    double S[COST_NSUM];
    for (int k = 0; k < COST_NSUM; k++) S[k] = part[k];
    if (c->reduce_fn && c->reduce_fn(S, COST_NSUM, c->reduce_user) != 0) return LH_EDEVICE;
    memcpy(t->res_sums, S, sizeof(S));
    if (c->prof) {  // algorithmic bytes with the measured K_t: B_fdf = 108 K_t, B_nn += 232 K_t
      c->prof_entries[c->prof_entry("cost_fdf")].bytes += 108.0 * S[13];
      if (t->sweep_bytes_pending) c->prof_entries[c->prof_entry("nn_sweep")].bytes += 232.0 * S[13];
    }
    t->sweep_bytes_pending = false;
  }
  if (!g.costs.empty()) {  // every pair's BFGS now takes its next step (up to its next evaluation request): independent, on the host pool
    if (!c->pool) {
      const char* e = getenv("LH_HOST_THREADS");
      int nt = e ? atoi(e) : 8;
      c->pool = new HostPool(std::max(0, nt - 1));
    }
    std::vector<Task*>& costs = g.costs;
    c->pool->parallel_for((int)costs.size(), [&costs](int i) { costs[i]->resume(); });
  }
  for (Task* t : g.moms) {  // deliver the moments; the task then runs its whole BFGS solve on the host
    const double* part = c->partials_host + (size_t)t->slot * c->partials_per_slot;  // FINAL_CHUNKS x 74 chunk sums
    double* S = t->mom.S;
    for (int k = 0; k < MOM_NSUM; k++) S[k] = 0.0;
    double walks = 0.0;
    for (int ch = 0; ch < FINAL_CHUNKS; ch++) {  // fixed order => bitwise reproducible
      for (int k = 0; k < MOM_NSUM; k++) S[k] += part[ch * MOM_ROW + k];
      walks += part[ch * MOM_ROW + MOM_NSUM];
    }
    t->last_walks = (long)walks;
    {  // LH_WALK_LOG=1: tree walks of every sweep of every pair (stderr; instrumentation)
      static const bool wlog = []() { const char* e = getenv("LH_WALK_LOG"); return e && atoi(e) != 0; }();
      if (wlog) fprintf(stderr, "[lh walks] slot %d sweep %d walks %ld\n", t->slot, t->sweeps_done - 1, t->last_walks);
    }
    if (c->reduce_fn && c->reduce_fn(t->mom.S, MOM_NSUM, c->reduce_user) != 0) return LH_EDEVICE;
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 4; cc++) t->mom.T0[cc * 4 + r] = t->req_T12[r * 4 + cc];
    t->mom.T0[3] = t->mom.T0[7] = t->mom.T0[11] = 0.f; t->mom.T0[15] = 1.f;
    t->mom.prepare();
    if (c->prof)  // fused K4+K5': algorithmic bytes B_nn + one B_fdf = 20 N + (232 + 108) K_t
      c->prof_entries[c->prof_entry("nn_sweep")].bytes += 340.0 * S[73];
    t->sweep_bytes_pending = false;
  }
  if (!g.moms.empty()) {  // the BFGS solves of the group's pairs are independent: run them on the host pool
    if (!c->pool) {
      const char* e = getenv("LH_HOST_THREADS");
      int nt = e ? atoi(e) : 8;
      c->pool = new HostPool(std::max(0, nt - 1));
    }
    std::vector<Task*>& moms = g.moms;
    c->pool->parallel_for((int)moms.size(), [&moms](int i) { moms[i]->resume(); });
  }
  g.costs.clear(); g.moms.clear(); g.sweeps.clear();
  for (size_t i = 0; i < g.active.size();) {  // retire finished pairs
    if (g.active[i]->req == REQ_DONE) {
      Task* t = g.active[i];
      if (t->aligned) {  // pcl::transformPointCloud(*input_, output, final_transformation_) (gicp.hpp:586), on the group's stream
        float T12[12];
        Task::T16_to_T12(t->result.T, T12);
        ProfScope p(c, "transform", 32.0 * t->src->n, g.stream);
        launch_transform_copy(t->src->xyz, t->aligned->nrm ? t->src->nrm : nullptr, t->aligned->intensity ? t->src->intensity : nullptr, t->src->n, T12,
                              t->aligned->xyz, t->aligned->nrm, t->aligned->intensity, g.stream);
      }
      g.free_slots.push_back(g.active[i]->slot);
      g.active.erase(g.active.begin() + i);
    } else
      i++;
  }
  return LH_OK;
}

// ---- device-driven loop (cost_mode 1) --------------------------------------------------------------------------------
// The whole outer loop of a pair lives on the GPU: every iteration is k_sweep_fused -> k_moments_final -> k_solve on the pair's
// device state, and the next sweep reads the transform k_solve left there.  The host only enqueues: ROUNDS iterations per
// group back to back, then one small download of the group's states to see which pairs have ended (converged, failed, or
// out of iterations); those retire (result, aligned output cloud), new pairs are admitted into their slots, and the next
// rounds go out.  Two groups on two streams as in the host-driven scheduler: one group's k_solve (a single wave per pair) and
// launch gaps are covered by the other group's sweeps.  Pairs that end early are skipped by the kernels until the host looks.
struct DevGroup {
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  std::vector<Task*> active;
  std::vector<int> free_slots;
  int slot_lo = 0, slot_hi = 0;  // this group's contiguous slot range
  bool pending = false;          // rounds are enqueued and a state download is in flight behind them
  std::vector<Task*> to_align;   // retired in this round, output cloud still to be written
};

// align()'s output clouds (gicp.hpp:586, pcl::transformPointCloud(*input_, output, final_transformation_)) of the pairs that retired
// together: one launch on the group's stream
static void dev_write_aligned(lh_ctx* c, DevGroup& g) {
  for (size_t o = 0; o < g.to_align.size(); o += MAX_XFORM_JOBS) {
    XformBatchArgs a;
    a.njobs = (int)std::min<size_t>(MAX_XFORM_JOBS, g.to_align.size() - o);
    a.pad = 0;
    int max_n = 0;
    double bytes = 0;
    for (int j = 0; j < a.njobs; j++) {
      Task* t = g.to_align[o + j];
      XformJob& x = a.job[j];
      x.in_xyz = t->src->xyz; x.out_xyz = t->aligned->xyz;
      x.in_nrm = t->aligned->nrm ? t->src->nrm : nullptr; x.out_nrm = t->aligned->nrm;
      x.in_int = t->aligned->intensity ? t->src->intensity : nullptr; x.out_int = t->aligned->intensity;
      x.n = t->src->n; x.pad = 0;
      Task::T16_to_T12(t->result.T, x.T);
      max_n = std::max(max_n, x.n);
      bytes += 32.0 * x.n;
    }
    ProfScope p(c, "transform", bytes, g.stream);
    launch_transform_copy_batch(a, max_n, g.stream);
  }
  g.to_align.clear();
}

static lh_status dev_retire(lh_ctx* c, DevGroup& g, Task* t) {
  t->os = c->states_host[t->slot];
  t->finish_result();
  if (c->prof)  // fused K4+K5': algorithmic bytes B_nn + one B_fdf per iteration = 20 N + (232 + 108) K_t
    c->prof_entries[c->prof_entry("nn_sweep")].bytes += 340.0 * t->os.corr_sum;
  if (t->aligned) g.to_align.push_back(t);  // its output cloud goes out with the other pairs that retire in this round
  if (t->trace && t->trace_dev) {
    HIPCHK(hipMemcpyAsync(t->trace, t->trace_dev, sizeof(lh_gicp_trace), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  if (t->trace_dev) { (void)lhFree(t->trace_dev); t->trace_dev = nullptr; }
  return LH_OK;
}

// enqueue `rounds` outer iterations for every active pair of the group, then the download of the group's states
static lh_status dev_enqueue(lh_ctx* c, DevGroup& g, int rounds) {
  hipStream_t st = g.stream;
  for (int r = 0; r < rounds; r++) {
    for (size_t o = 0; o < g.active.size(); o += MAX_JOBS) {
      SweepArgs a;
      CostArgs ca;
      SolveArgs sa;
      a.njobs = (int)std::min<size_t>(MAX_JOBS, g.active.size() - o);
      a.bpj = 0; a.max_depth = 0; a.pad = 0;
      ca.njobs = a.njobs; ca.pad = 0;
      sa.njobs = a.njobs;
      int max_n = 0;
      double bytes = 0;
      bool normals_only = true;
      uint32_t split_mask = 0u;
      SweepArgs seed;
      seed.njobs = 0; seed.max_depth = 0; seed.pad = 0; seed.bpj = 0;
      int smax = 0;
      for (int j = 0; j < a.njobs; j++) {
        Task* t = g.active[o + j];
        a.job[j].slot = t->slot;
        a.job[j].pad = 0;
        Task::T16_to_T12(I16, a.job[j].T);  // only the seed pass of a cold pair reads it (transformation_ = I); sweeps read the device state
        ca.job[j].slot = t->slot;
        ca.job[j].out_offset = t->slot * (FINAL_CHUNKS * MOM_ROW);
        memcpy(ca.job[j].T, a.job[j].T, sizeof(a.job[j].T));
        sa.slot[j] = t->slot;
        max_n = std::max(max_n, t->src->n);
        if (t->P.recompute_source_cov || t->P.recompute_target_cov) normals_only = false;
        if (t->enq_iters < t->P.max_iterations) bytes += 20.0 * t->src->n;  // SURVEY 8d B_nn = 20 N + 232 K_t; the K_t terms are added at retirement
        if (t->first_sweep) {  // cold pair: seed pre-pass so its first sweep starts warm
          seed.job[seed.njobs++] = a.job[j];
          smax = std::max(smax, t->src->n);
          t->first_sweep = false;
        }
        if (sweep_is_split(t, t->enq_iters)) split_mask |= 1u << j;
        t->enq_iters++;
      }
      if (seed.njobs > 0) {
        ProfScope p(c, "nn_seed", 0.0, st);
        launch_seed(c->descs_dev, seed, smax, st);
      }
      {
        ProfScope p(c, "nn_sweep", bytes, st);
        launch_sweep_fused(c->descs_dev, a, split_mask, max_n, c->mom_partials_dev, c->mom_stride, c->states_dev, normals_only, c->wmask_dev, c->mask_stride, st);
        launch_moments_final(c->descs_dev, ca, c->mom_partials_dev, c->mom_stride, c->chunks_dev, c->states_dev, c->wmask_dev, c->mask_stride, st);
      }
      {
        ProfScope p(c, "bfgs_solve", 0.0, st);
        launch_solve(c->descs_dev, sa, c->chunks_dev, FINAL_CHUNKS * MOM_ROW, c->states_dev, st);
      }
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->states_host + g.slot_lo, c->states_dev + g.slot_lo, sizeof(OuterState) * (size_t)(g.slot_hi - g.slot_lo),
                        hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(g.ev, st));
  g.pending = true;
  return LH_OK;
}

static lh_status run_tasks_device(lh_ctx* c, std::vector<Task*>& tasks, int in_flight, bool rebuild_index, std::vector<Workspace>* slot_ws) {
  static const int rounds_cfg = []() { const char* e = getenv("LH_DEVICE_ROUNDS"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
  // Groups: a pair's solve (one wave, tens of sequential cost evaluations) takes about as long as its sweep, so with more groups
  // in flight there is always somebody's sweep to run beside the other groups' solves.  Profiling keeps one group so that the
  // HIP-event times of the launches do not overlap.
  static const int groups_cfg = []() { const char* e = getenv("LH_DEVICE_GROUPS"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > lh_ctx::MAX_GROUPS ? lh_ctx::MAX_GROUPS : v); }();
  // one group per MAX_JOBS (32) pairs in flight -- a group's launch covers all its pairs -- up to 32 groups = streams: with 256 in
  // flight, eight groups of 32 ran 14 % more pairs/s than four of 64 (each chain is half latency: solve, start-up, lone searches)
  int G = groups_cfg ? groups_cfg : (in_flight >= 64 ? std::min(lh_ctx::MAX_GROUPS, in_flight / MAX_JOBS) : (in_flight >= 16 ? 2 : 1));
  if (c->prof) G = 1;
  G = std::max(1, std::min(G, in_flight));
  hipStream_t* extra[lh_ctx::MAX_GROUPS - 1] = {&c->stream2, &c->stream3, &c->stream4};
  for (int k = 0; k < lh_ctx::MAX_GROUPS - 4; k++) extra[3 + k] = &c->stream_more[k];
  for (int gi = 1; gi < G; gi++)
    if (!*extra[gi - 1]) HIPCHK(hipStreamCreateWithFlags(extra[gi - 1], hipStreamNonBlocking));
  DevGroup groups[lh_ctx::MAX_GROUPS];
  groups[0].stream = c->stream;
  for (int gi = 1; gi < G; gi++) groups[gi].stream = *extra[gi - 1];
  {
    int per = (in_flight + G - 1) / G, s = 0;
    for (int gi = 0; gi < G; gi++) {
      groups[gi].ev = c->group_ev[gi];
      groups[gi].slot_lo = s;
      for (int k = 0; k < per && s < in_flight; k++, s++) groups[gi].free_slots.push_back(s);
      groups[gi].slot_hi = s;
    }
  }
  // LH_HOST_PROF=1: where the scheduling thread's time goes (stderr, per batch): waiting for the GPU vs feeding it
  static const bool host_prof = []() { const char* e = getenv("LH_HOST_PROF"); return e && atoi(e) != 0; }();
  double hp_wait = 0, hp_retire = 0, hp_admit = 0, hp_enq = 0, hp_build = 0, hp_prep = 0;
  auto hp_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double hp_t0 = hp_now();
  size_t next = 0;
  lh_status err = LH_OK;
  const uint64_t epoch = ++c->epoch;
  auto fail = [&](lh_status st) {
    (void)hipStreamSynchronize(c->stream);
    c->sync_side_streams();
    for (Task* t : tasks)
      if (t->trace_dev) { (void)lhFree(t->trace_dev); t->trace_dev = nullptr; }
    return st;
  };
  auto busy = [&]() {
    for (int gi = 0; gi < G; gi++)
      if (!groups[gi].active.empty()) return true;
    return false;
  };
  while (next < tasks.size() || busy()) {
    for (int gi = 0; gi < G; gi++) {
      DevGroup& g = groups[gi];
      lh_status st;
      double hp_a = hp_now();
      if (g.pending) {  // wait for THIS group's rounds; the other group's are still queued / running
        if (hipEventSynchronize(g.ev) != hipSuccess) return fail(LH_EDEVICE);   // (through fail(): the other groups' streams drain, the device traces are handed back)
        hp_wait += hp_now() - hp_a;
        hp_a = hp_now();
        g.pending = false;
        for (size_t i = 0; i < g.active.size();) {
          Task* t = g.active[i];
          const OuterState& os = c->states_host[t->slot];
          if (os.done) {
            st = dev_retire(c, g, t);
            if (st) return fail(st);
            g.free_slots.push_back(t->slot);
            g.active.erase(g.active.begin() + i);
          } else
            i++;
        }
        dev_write_aligned(c, g);
        hp_retire += hp_now() - hp_a;
      }
      hp_a = hp_now();
      // Admission is group-synchronous: new pairs enter a group only when ALL its pairs have retired, so that a group's pairs stay at
      // the same iteration -- every launch is one kernel over all of them (a mixed group launches the fused sweep for its young
      // pairs and k_late + k_walk for the others, each half empty), and index builds / seed passes always cover a whole group.  The
      // slots of early finishers wait (mean 18.5 of 20 iterations on the bench pairs); measured on a 512-pair queue, 128 in flight:
      // 6 490 -> 7 000 pairs/s.  LH_ADMIT=slot restores slot-by-slot admission.
      static const bool admit_by_slot = []() { const char* e = getenv("LH_ADMIT"); return e && strcmp(e, "slot") == 0; }();
      if (next < tasks.size() && !g.free_slots.empty() && (admit_by_slot || g.active.empty())) {  // admit: the NN indexes of all newly admitted targets are built together
        std::vector<lh_cloud*> to_build;
        size_t nn = next;
        bool built_elsewhere = false;
        for (size_t k = 0; k < g.free_slots.size() && nn < tasks.size(); k++, nn++) {
          lh_cloud* tg = tasks[nn]->tgt;
          if (!tg || tg->n <= 0) continue;
          // a target is (re)built once per call: a cloud shared by pairs of several groups (a scan-to-submap batch) was built by the
          // first group that admitted one of its pairs -- rebuilding it in place here would rewrite the tree under that group's sweeps
          if (tg->built_epoch == epoch && tg->has_index) { built_elsewhere = true; continue; }
          if ((rebuild_index || !tg->has_index) && std::find(to_build.begin(), to_build.end(), tg) == to_build.end()) to_build.push_back(tg);
        }
        if (!to_build.empty()) {
          const double hb = hp_now();
          st = build_indices(c, to_build.data(), (int)to_build.size(), g.stream);
          if (st) return fail(st);
          for (lh_cloud* tg : to_build) tg->built_epoch = epoch;
          hp_build += hp_now() - hb;
        } else if (built_elsewhere && c->idx_build_done) {
          // builds are chained through idx_build_done (shared scratch), so the latest record covers every earlier build of this call
          if (hipStreamWaitEvent(g.stream, c->idx_build_done, 0) != hipSuccess) return fail(LH_EDEVICE);
        }
        const double hpp = hp_now();
        std::vector<int> admitted;
        while (next < tasks.size() && !g.free_slots.empty()) {
          Task* t = tasks[next++];
          t->slot = g.free_slots.back();
          g.free_slots.pop_back();
          t->stream = g.stream;
          if (slot_ws) t->ws = &(*slot_ws)[t->slot];
          t->trace_dev = nullptr;
          st = LH_OK;
          if (t->trace) {
            if (lhMalloc(&t->trace_dev, sizeof(lh_gicp_trace)) != hipSuccess) st = LH_ENOMEM;
            else if (hipMemsetAsync(t->trace_dev, 0, sizeof(int), g.stream) != hipSuccess) st = LH_EDEVICE;  // n_iters = 0
            t->trace->n_iters = 0;
          }
          if (!st) st = task_prepare(c, t, false, false);
          if (!st) {  // the pair's loop state: transformation_ = I, nothing done yet (pcl::Registration::align); uploaded below with the others
            outer_state_init(&c->states_init[t->slot]);
            admitted.push_back(t->slot);
          }
          if (st) {
            if (t->trace_dev) { (void)lhFree(t->trace_dev); t->trace_dev = nullptr; }
            memset(&t->result, 0, sizeof(t->result));
            memcpy(t->result.T, I16, sizeof(I16));
            t->result.status = st;
            t->result.fitness = NAN;
            g.free_slots.push_back(t->slot);
            err = st;
            continue;
          }
          t->first_sweep = true;
          t->enq_iters = 0;
          g.active.push_back(t);
        }
        hp_prep += hp_now() - hpp;
        if (!admitted.empty()) {
          // ONE copy for the group's descriptors (the host copies of the slots that keep running are unchanged) and one per run of
          // admitted slots for the loop states: a copy is a small kernel on the group's stream, and three per pair were 840 per step
          if (hipMemcpyAsync(&c->descs_dev[g.slot_lo], &c->descs_host[g.slot_lo], sizeof(PairDesc) * (size_t)(g.slot_hi - g.slot_lo), hipMemcpyHostToDevice,
                             g.stream) != hipSuccess)
            return fail(LH_EDEVICE);
          std::sort(admitted.begin(), admitted.end());
          for (size_t a0 = 0; a0 < admitted.size();) {
            size_t a1 = a0 + 1;
            while (a1 < admitted.size() && admitted[a1] == admitted[a1 - 1] + 1) a1++;
            if (hipMemcpyAsync(&c->states_dev[admitted[a0]], &c->states_init[admitted[a0]], sizeof(OuterState) * (a1 - a0), hipMemcpyHostToDevice, g.stream) !=
                hipSuccess)
              return fail(LH_EDEVICE);
            a0 = a1;
          }
        }
      }
      hp_admit += hp_now() - hp_a;
      hp_a = hp_now();
      if (!g.active.empty()) {
        // how many iterations before the host looks again: no pair needs more than what is left of its max_iterations
        int need = 0;
        for (Task* t : g.active) need = std::max(need, t->P.max_iterations - t->enq_iters);
        st = dev_enqueue(c, g, std::max(1, std::min(rounds_cfg, need)));
        if (st) return fail(st);
      }
      hp_enq += hp_now() - hp_a;
    }
  }
  if (host_prof)
    fprintf(stderr, "[lh host] %zu pairs, %d groups: total %.3f ms = wait %.3f + retire %.3f + admit %.3f (index builds %.3f, pair set-up %.3f) + enqueue %.3f\n", tasks.size(), G,
            1e3 * (hp_now() - hp_t0), 1e3 * hp_wait, 1e3 * hp_retire, 1e3 * hp_admit, 1e3 * hp_build, 1e3 * hp_prep, 1e3 * hp_enq);
  return err;
}

// run a set of tasks to completion, at most `in_flight` concurrently
static lh_status run_tasks_host(lh_ctx* c, std::vector<Task*>& tasks, int in_flight, bool rebuild_index, std::vector<Workspace>* slot_ws) {
  // Groups: one per MAX_JOBS (32) pairs in flight like the device-driven loop, each on its own stream -- while the host thread delivers
  // one group's sums and resumes its solves, the other groups' kernels keep the GPU busy.  With two groups (round 2) the reference-
  // arithmetic mode, whose every cost evaluation is a launch + a synchronisation, left the GPU idle 40 % of the time (18 502 k_cost
  // launches per 512-pair step).  Profiling keeps one group so HIP-event times do not overlap.
  static const int host_groups_cfg = []() { const char* e = getenv("LH_HOST_GROUPS"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > lh_ctx::MAX_GROUPS ? lh_ctx::MAX_GROUPS : v); }();
  int G = host_groups_cfg ? host_groups_cfg : (in_flight >= 64 ? std::min(16, in_flight / MAX_JOBS) : (in_flight >= 16 ? 2 : 1));
  if (c->prof) G = 1;
  G = std::max(1, std::min(G, in_flight));
  hipStream_t* extra[lh_ctx::MAX_GROUPS - 1] = {&c->stream2, &c->stream3, &c->stream4};
  for (int k = 0; k < lh_ctx::MAX_GROUPS - 4; k++) extra[3 + k] = &c->stream_more[k];
  for (int gi = 1; gi < G; gi++)
    if (!*extra[gi - 1]) HIPCHK(hipStreamCreateWithFlags(extra[gi - 1], hipStreamNonBlocking));
  std::vector<Group> groups(G);
  groups[0].stream = c->stream;
  for (int gi = 1; gi < G; gi++) groups[gi].stream = *extra[gi - 1];
  {
    int per = (in_flight + G - 1) / G, s = 0;
    for (int gi = 0; gi < G; gi++)
      for (int k = 0; k < per && s < in_flight; k++, s++) groups[gi].free_slots.push_back(s);
  }
  size_t next = 0;
  lh_status err = LH_OK;
  const uint64_t epoch = ++c->epoch;
  auto busy = [&]() {
    for (int gi = 0; gi < G; gi++)
      if (!groups[gi].active.empty() || groups[gi].inflight) return true;
    return false;
  };
  // error exit: the other scheduler group may still have kernels queued that read or write pooled device buffers (index
  // build, sweeps); nothing may be handed back to the pool, or to the caller, before both streams have drained
  auto fail = [&](lh_status st) {
    (void)hipStreamSynchronize(c->stream);
    c->sync_side_streams();
    return st;
  };
  while (next < tasks.size() || busy()) {
    for (int gi = 0; gi < G; gi++) {
      Group& g = groups[gi];
      lh_status st = group_collect(c, g);  // waits for THIS group's kernels; the other group's are still queued/running
      if (st) return fail(st);
      // admit new pairs: the NN indexes of all newly admitted targets are built together (batched launches + one sort)
      if (next < tasks.size() && !g.free_slots.empty()) {
        std::vector<lh_cloud*> to_build;
        size_t nn = next;
        bool built_elsewhere = false;
        for (size_t k = 0; k < g.free_slots.size() && nn < tasks.size(); k++, nn++) {
          lh_cloud* tg = tasks[nn]->tgt;
          if (!tg || tg->n <= 0) continue;
          if (tg->built_epoch == epoch && tg->has_index) { built_elsewhere = true; continue; }   // built by the other group in this call (see run_tasks_device)
          if ((rebuild_index || !tg->has_index) && std::find(to_build.begin(), to_build.end(), tg) == to_build.end()) to_build.push_back(tg);
        }
        if (!to_build.empty()) {
          st = build_indices(c, to_build.data(), (int)to_build.size(), g.stream);
          if (st) return fail(st);
          for (lh_cloud* tg : to_build) tg->built_epoch = epoch;
        } else if (built_elsewhere && c->idx_build_done) {
          if (hipStreamWaitEvent(g.stream, c->idx_build_done, 0) != hipSuccess) return fail(LH_EDEVICE);
        }
        while (next < tasks.size() && !g.free_slots.empty()) {
          Task* t = tasks[next++];
          t->slot = g.free_slots.back();
          g.free_slots.pop_back();
          t->stream = g.stream;
          if (slot_ws) t->ws = &(*slot_ws)[t->slot];  // batch mode: workspaces belong to slots
          st = task_prepare(c, t, false);
          if (st) {
            memset(&t->result, 0, sizeof(t->result));
            memcpy(t->result.T, I16, sizeof(I16));
            t->result.status = st;
            t->result.fitness = NAN;
            g.free_slots.push_back(t->slot);
            err = st;
            continue;
          }
          t->start();
          g.active.push_back(t);
        }
      }
      st = group_launch(c, g);
      if (st) return fail(st);
    }
  }
  return err;
}

// Where the loop between two sweeps runs in cost_mode 1 (lh_gicp_params.solver): on the device (k_solve) the host is out of the
// loop and throughput no longer depends on it -- the choice for batches; on the host one outer iteration costs a sync and a
// few microseconds of BFGS on a CPU core, against ~3 us per cost evaluation on a single GPU wave -- the choice for one pair at
// a time (measured: 1.2 ms vs 2.5 ms per 100k-point pair at 20 iterations).  solver = 0 picks by the number of pairs in
// flight.  The host loop is also taken when something needs the host inside the loop: the source-sharded pair's SUM hook (its
// sums cross ranks through a host callback) and the debug-statistics sweeps.  Both loops give bit-identical results.
constexpr int DEVICE_LOOP_MIN_IN_FLIGHT = 8;
static lh_status run_tasks(lh_ctx* c, std::vector<Task*>& tasks, int in_flight, bool rebuild_index, std::vector<Workspace>* slot_ws = nullptr) {
  bool device_loop = !c->reduce_fn;
  bool forced = false;
  for (Task* t : tasks) {
    if (t->P.cost_mode != 1 || t->P.solver == 1 || t->count_stats || t->P.max_iterations < 1) device_loop = false;
    if (t->P.solver == 2) forced = true;
  }
  if (device_loop && !forced && std::min<size_t>(in_flight, tasks.size()) < (size_t)DEVICE_LOOP_MIN_IN_FLIGHT) device_loop = false;
  return device_loop ? run_tasks_device(c, tasks, in_flight, rebuild_index, slot_ws) : run_tasks_host(c, tasks, in_flight, rebuild_index, slot_ws);
}

// ---------------------------------------------------------------------------------------------------------
struct lh_gicp {
  lh_ctx* ctx = nullptr;
  lh_gicp_params P;
  lh_cloud *src = nullptr, *tgt = nullptr;
  bool own_src = false, own_tgt = false;
  Workspace ws;
  Task task;
  float last_T[16];
  bool have_result = false;
  // debug sweep state
  bool dbg_ready = false, dbg_prepared = false;
};

// ---------------------------------------------------------------------------------------------------------
// host <-> device cloud conversion
static lh_status upload_view(lh_ctx* c, const lh_cloud_view* v, lh_cloud** out) {
  if (!v || !v->base || v->count == 0 || v->stride < 12) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->device));
  lh_cloud* cl = new lh_cloud();
  cl->ctx = c;
  cl->n = (int)v->count;
  cl->n_pad = round_up(cl->n, 256);
  size_t n = v->count;
  bool has_n = v->off_normal != UINT32_MAX, has_i = v->off_intensity != UINT32_MAX;
  std::vector<float> xyz(n * 4), nrm(has_n ? n * 4 : 0), inten(has_i ? n : 0);
  const char* base = (const char*)v->base;
  for (size_t i = 0; i < n; i++) {
    const char* p = base + i * v->stride;
    const float* f = (const float*)(p + v->off_xyz);
    xyz[4 * i] = f[0]; xyz[4 * i + 1] = f[1]; xyz[4 * i + 2] = f[2]; xyz[4 * i + 3] = 1.0f;
    if (has_n) {
      const float* g = (const float*)(p + v->off_normal);
      nrm[4 * i] = g[0]; nrm[4 * i + 1] = g[1]; nrm[4 * i + 2] = g[2];
      nrm[4 * i + 3] = (v->off_curvature != UINT32_MAX) ? *(const float*)(p + v->off_curvature) : 0.0f;
    }
    if (has_i) inten[i] = *(const float*)(p + v->off_intensity);
  }
  hipError_t e = lhMalloc(&cl->xyz, sizeof(float4) * (size_t)cl->n_pad);
  if (e == hipSuccess && has_n) e = lhMalloc(&cl->nrm, sizeof(float4) * (size_t)cl->n_pad);
  if (e == hipSuccess && has_i) e = lhMalloc(&cl->intensity, sizeof(float) * (size_t)cl->n_pad);
  if (e == hipSuccess) e = hipMemcpyAsync(cl->xyz, xyz.data(), sizeof(float) * 4 * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && has_n) e = hipMemcpyAsync(cl->nrm, nrm.data(), sizeof(float) * 4 * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && has_i) e = hipMemcpyAsync(cl->intensity, inten.data(), sizeof(float) * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // staging vectors die at scope exit
  if (e != hipSuccess) {
    fprintf(stderr, "[locus_hip] cloud upload failed: %s\n", hipGetErrorString(e));
    cloud_free(cl);
    return e == hipErrorOutOfMemory ? LH_ENOMEM : LH_EDEVICE;
  }
  *out = cl;
  return LH_OK;
}

static void fill_T12(const float* T16, float* T12) { Task::T16_to_T12(T16, T12); }

// =========================================================================================================
extern "C" {

int lh_abi_version(void) { return LH_ABI_VERSION; }

const char* lh_status_string(lh_status s) {
  switch (s) {
    case LH_OK: return "ok";
    case LH_EINVAL: return "invalid argument";
    case LH_ENOMEM: return "out of memory";
    case LH_EDEVICE: return "HIP device error / no device";
    case LH_ETOO_FEW_CORR: return "fewer than 4 correspondences";
    case LH_ESOLVER: return "BFGS solver did not converge";
    case LH_ENO_NN: return "no nearest neighbour";
    default: return "unknown";
  }
}

void lh_default_gicp_params(lh_gicp_params* p) {
  if (!p) return;
  p->max_iterations = 200;           // gicp.h:129
  p->max_inner_iterations = 20;      // gicp.h:121
  p->corr_dist = 5.0;                // gicp.h:131
  p->transformation_epsilon = 5e-4;  // gicp.h:130
  p->rotation_epsilon = 2e-3;        // gicp.h:119
  p->gicp_epsilon = 1e-3;            // gicp.h:118
  p->k_correspondences = 20;         // gicp.h:112
  p->recompute_source_cov = 0;       // gicp.h:115
  p->recompute_target_cov = 0;       // gicp.h:116
  p->num_threads = 1;
  p->enable_timing = 0;
  p->cost_mode = 1;
  p->solver = 0;
}

lh_status lh_create(lh_ctx** out, int device_id) {
  if (!out) return LH_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "[locus_hip] no HIP device available: this library has no CPU fallback\n");
    return LH_EDEVICE;
  }
  if (device_id < 0 || device_id >= ndev) return LH_EINVAL;
  HIPCHK(hipSetDevice(device_id));
  lh_ctx* c = new lh_ctx();
  c->device = device_id;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc(&c->bbox, sizeof(uint32_t) * 8));
  *out = c;
  return LH_OK;
}

void lh_destroy(lh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->sync_side_streams();
  delete c->pool;
  c->pool = nullptr;
  for (auto& w : c->slot_ws) w.release();
  c->slot_ws.clear();
  c->prof_flush();
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  (void)lhFree(c->keys0); (void)lhFree(c->keys1); (void)lhFree(c->vals0); (void)lhFree(c->vals1);
  (void)lhFree(c->k64a); (void)lhFree(c->k64b); (void)lhFree(c->v32a); (void)lhFree(c->v32b); (void)lhFree(c->sort64_temp);
  (void)lhFree(c->tree_tmp); (void)lhFree(c->scan_tmp); (void)lhFree(c->k32a); (void)lhFree(c->k32b); (void)lhFree(c->rs_hist);
  (void)lhFree(c->idx_bbox); (void)lhFree(c->idx_descs_dev);
  if (c->idx_descs_host) (void)hipHostFree(c->idx_descs_host);
  for (hipEvent_t e : c->idx_copy_done)
    if (e) (void)hipEventDestroy(e);
  if (c->idx_build_done) (void)hipEventDestroy(c->idx_build_done);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
  for (hipStream_t s : c->stream_more)
    if (s) (void)hipStreamDestroy(s);
  (void)lhFree(c->sort_temp); (void)lhFree(c->bbox); (void)lhFree(c->descs_dev); (void)lhFree(c->mom_partials_dev); (void)lhFree(c->wmask_dev);
  (void)lhFree(c->states_dev); (void)lhFree(c->chunks_dev);
  if (c->states_host) (void)hipHostFree(c->states_host);
  if (c->states_init) (void)hipHostFree(c->states_init);
  for (int k = 0; k < lh_ctx::MAX_GROUPS; k++)
    if (c->group_ev[k]) (void)hipEventDestroy(c->group_ev[k]);
  if (c->descs_host) (void)hipHostFree(c->descs_host);
  if (c->partials_host) (void)hipHostFree(c->partials_host);
  if (c->small_host) (void)hipHostFree(c->small_host);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

lh_status lh_synchronize(lh_ctx* c) {
  if (!c) return LH_EINVAL;
  HIPCHK(hipStreamSynchronize(c->stream));
  return LH_OK;
}

// ---- clouds ----------------------------------------------------------------------------------------------
lh_status lh_cloud_create(lh_ctx* ctx, const lh_cloud_view* view, lh_cloud** out) {
  if (!ctx || !out) return LH_EINVAL;
  return upload_view(ctx, view, out);
}
void lh_cloud_destroy(lh_cloud* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  cloud_free(c);
}
uint32_t lh_cloud_size(const lh_cloud* c) { return c ? (uint32_t)c->n : 0; }
lh_status lh_cloud_build_index(lh_cloud* c) {
  if (!c) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->ctx->device));
  return cloud_build_index(c);
}
lh_status lh_cloud_drop_index(lh_cloud* c) {
  if (!c) return LH_EINVAL;
  c->has_index = false;
  c->cov_k = 0;
  return LH_OK;
}
lh_status lh_cloud_download(const lh_cloud* c, void* out_base, uint32_t stride, uint32_t off_xyz, uint32_t off_normal,
                            uint32_t off_intensity, uint32_t off_curvature) {
  if (!c || !out_base || stride < 12) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->ctx->device));
  size_t n = (size_t)c->n;
  std::vector<float> xyz(n * 4), nrm, inten;
  HIPCHK(hipMemcpyAsync(xyz.data(), c->xyz, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, c->ctx->stream));
  if (c->nrm && off_normal != UINT32_MAX) {
    nrm.resize(n * 4);
    HIPCHK(hipMemcpyAsync(nrm.data(), c->nrm, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, c->ctx->stream));
  }
  if (c->intensity && off_intensity != UINT32_MAX) {
    inten.resize(n);
    HIPCHK(hipMemcpyAsync(inten.data(), c->intensity, sizeof(float) * n, hipMemcpyDeviceToHost, c->ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(c->ctx->stream));
  char* base = (char*)out_base;
  for (size_t i = 0; i < n; i++) {
    char* p = base + i * stride;
    memcpy(p + off_xyz, &xyz[4 * i], 12);
    if (!nrm.empty()) {
      memcpy(p + off_normal, &nrm[4 * i], 12);
      if (off_curvature != UINT32_MAX) memcpy(p + off_curvature, &nrm[4 * i + 3], 4);
    }
    if (!inten.empty()) memcpy(p + off_intensity, &inten[i], 4);
  }
  return LH_OK;
}
lh_status lh_cloud_transform(const lh_cloud* in, const float T[16], int with_normals, lh_cloud** out) {
  if (!in || !T || !out) return LH_EINVAL;
  lh_ctx* c = in->ctx;
  HIPCHK(hipSetDevice(c->device));
  DevGuard guard;
  lh_cloud* o = (*out == in) ? const_cast<lh_cloud*>(in) : new lh_cloud();
  if (o != in) {
    guard.cloud = o;  // released again on any failure below
    o->ctx = c; o->n = in->n; o->n_pad = in->n_pad;
    HIPCHK(lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad));
    if (in->nrm) HIPCHK(lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad));
    if (in->intensity) {
      HIPCHK(lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad));
      HIPCHK(hipMemcpyAsync(o->intensity, in->intensity, sizeof(float) * (size_t)in->n, hipMemcpyDeviceToDevice, c->stream));
    }
    if (in->nrm && !with_normals) HIPCHK(hipMemcpyAsync(o->nrm, in->nrm, sizeof(float4) * (size_t)in->n, hipMemcpyDeviceToDevice, c->stream));
  }
  float T12[12];
  fill_T12(T, T12);
  { ProfScope p(c, "transform", (with_normals ? 64.0 : 32.0) * in->n);
    launch_transform(in->xyz, with_normals ? in->nrm : nullptr, in->n, T12, o->xyz, with_normals ? o->nrm : nullptr, c->stream); }
  HIPCHK(hipGetLastError());
  o->has_index = false;
  o->cov_k = 0;
  (void)guard.keep_cloud();
  *out = o;
  return LH_OK;
}

// points [first, first+count) as a new cloud (device-to-device): the source shard of a rank, and the pieces
// PointCloudMerger concatenates (lh_cloud_concat)
lh_status lh_cloud_slice(const lh_cloud* in, uint32_t first, uint32_t count, lh_cloud** out) {
  if (!in || !out || count == 0 || (uint64_t)first + count > (uint64_t)in->n) return LH_EINVAL;
  lh_ctx* c = in->ctx;
  HIPCHK(hipSetDevice(c->device));
  lh_cloud* o = new lh_cloud();
  o->ctx = c; o->n = (int)count; o->n_pad = round_up(o->n, 256);
  hipError_t e = lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad);
  if (e == hipSuccess && in->nrm) e = lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad);
  if (e == hipSuccess && in->intensity) e = lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad);
  if (e != hipSuccess) { cloud_free(o); return LH_ENOMEM; }
  e = hipMemcpyAsync(o->xyz, in->xyz + first, sizeof(float4) * (size_t)count, hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess && in->nrm) e = hipMemcpyAsync(o->nrm, in->nrm + first, sizeof(float4) * (size_t)count, hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess && in->intensity) e = hipMemcpyAsync(o->intensity, in->intensity + first, sizeof(float) * (size_t)count, hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { cloud_free(o); return LH_EDEVICE; }
  *out = o;
  return LH_OK;
}
// PointCloudMerger concatenation (PointCloudMerger.cc:158-159, `*merged = *a + *b`): points of the inputs in order; normals /
// intensity are kept only if every input has them
lh_status lh_cloud_concat(lh_cloud* const* parts, int n_parts, lh_cloud** out) {
  if (!parts || n_parts < 1 || !out || !parts[0]) return LH_EINVAL;
  lh_ctx* c = parts[0]->ctx;
  uint64_t total = 0;
  bool nrm = true, inten = true;
  for (int i = 0; i < n_parts; i++) {
    if (!parts[i] || parts[i]->ctx != c) return LH_EINVAL;
    total += (uint64_t)parts[i]->n;
    nrm = nrm && parts[i]->nrm;
    inten = inten && parts[i]->intensity;
  }
  if (total == 0 || total > 0x7fffff00ull) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->device));
  lh_cloud* o = new lh_cloud();
  o->ctx = c; o->n = (int)total; o->n_pad = round_up(o->n, 256);
  hipError_t e = lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad);
  if (e == hipSuccess && nrm) e = lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad);
  if (e == hipSuccess && inten) e = lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad);
  if (e != hipSuccess) { cloud_free(o); return LH_ENOMEM; }
  size_t at = 0;
  for (int i = 0; i < n_parts && e == hipSuccess; i++) {
    size_t k = (size_t)parts[i]->n;
    if (!k) continue;
    e = hipMemcpyAsync(o->xyz + at, parts[i]->xyz, sizeof(float4) * k, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && nrm) e = hipMemcpyAsync(o->nrm + at, parts[i]->nrm, sizeof(float4) * k, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && inten) e = hipMemcpyAsync(o->intensity + at, parts[i]->intensity, sizeof(float) * k, hipMemcpyDeviceToDevice, c->stream);
    at += k;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { cloud_free(o); return LH_EDEVICE; }
  *out = o;
  return LH_OK;
}

// ---- registration object ----------------------------------------------------------------------------------
lh_status lh_gicp_create(lh_ctx* ctx, const lh_gicp_params* p, lh_gicp** out) {
  if (!ctx || !out) return LH_EINVAL;
  lh_gicp* g = new lh_gicp();
  g->ctx = ctx;
  if (p) g->P = *p; else lh_default_gicp_params(&g->P);
  memcpy(g->last_T, I16, sizeof(I16));
  *out = g;
  return LH_OK;
}
void lh_gicp_destroy(lh_gicp* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  if (g->own_src) cloud_free(g->src);
  if (g->own_tgt && g->tgt != g->src) cloud_free(g->tgt);
  g->ws.release();
  delete g;
}
lh_status lh_gicp_set_params(lh_gicp* g, const lh_gicp_params* p) {
  if (!g || !p) return LH_EINVAL;
  g->P = *p;
  return LH_OK;
}
static void gicp_drop_src(lh_gicp* g) {
  if (g->own_src && g->src && g->src != g->tgt) { (void)hipStreamSynchronize(g->ctx->stream); cloud_free(g->src); }
  g->src = nullptr; g->own_src = false;
}
static void gicp_drop_tgt(lh_gicp* g) {
  if (g->own_tgt && g->tgt && g->tgt != g->src) { (void)hipStreamSynchronize(g->ctx->stream); cloud_free(g->tgt); }
  g->tgt = nullptr; g->own_tgt = false;
}
lh_status lh_gicp_set_source(lh_gicp* g, const lh_cloud_view* v) {
  if (!g) return LH_EINVAL;
  lh_cloud* c = nullptr;
  lh_status st = upload_view(g->ctx, v, &c);
  if (st) return st;
  gicp_drop_src(g);
  g->src = c; g->own_src = true; g->have_result = false; g->dbg_ready = false; g->dbg_prepared = false;
  return LH_OK;
}
lh_status lh_gicp_set_target(lh_gicp* g, const lh_cloud_view* v) {
  if (!g) return LH_EINVAL;
  lh_cloud* c = nullptr;
  lh_status st = upload_view(g->ctx, v, &c);
  if (st) return st;
  gicp_drop_tgt(g);
  g->tgt = c; g->own_tgt = true; g->have_result = false; g->dbg_ready = false; g->dbg_prepared = false;
  return LH_OK;
}
lh_status lh_gicp_set_source_cloud(lh_gicp* g, lh_cloud* c) {
  if (!g || !c || c->ctx != g->ctx) return LH_EINVAL;
  gicp_drop_src(g);
  g->src = c; g->own_src = false; g->have_result = false; g->dbg_ready = false; g->dbg_prepared = false;
  return LH_OK;
}
lh_status lh_gicp_set_target_cloud(lh_gicp* g, lh_cloud* c) {
  if (!g || !c || c->ctx != g->ctx) return LH_EINVAL;
  gicp_drop_tgt(g);
  g->tgt = c; g->own_tgt = false; g->have_result = false; g->dbg_ready = false; g->dbg_prepared = false;
  return LH_OK;
}
lh_status lh_gicp_promote_source_to_target(lh_gicp* g) {
  if (!g || !g->src) return LH_EINVAL;
  lh_cloud* s = g->src;
  bool own = g->own_src;
  g->src = nullptr; g->own_src = false;
  gicp_drop_tgt(g);
  g->tgt = s; g->own_tgt = own;
  g->tgt->has_index = false;  // align() rebuilds the index (target changed)
  g->have_result = false; g->dbg_ready = false; g->dbg_prepared = false;
  return LH_OK;
}

lh_status lh_gicp_align(lh_gicp* g, const float guess[16], lh_gicp_result* out, lh_gicp_trace* trace, void* aligned_out,
                        uint32_t stride, uint32_t off_xyz) {
  if (!g || !out) return LH_EINVAL;
  lh_ctx* c = g->ctx;
  HIPCHK(hipSetDevice(c->device));
  memset(out, 0, sizeof(*out));
  memcpy(out->T, I16, sizeof(I16));
  out->fitness = NAN;
  if (!g->src || !g->tgt) { out->status = LH_EINVAL; return LH_EINVAL; }
  lh_status st = ctx_ensure_slots(c, 1, std::max(g->src->n, 1));
  if (st) { out->status = st; return st; }
  g->dbg_prepared = false;
  Task& t = g->task;
  t.count_stats = false;
  t.P = g->P; t.src = g->src; t.tgt = g->tgt; t.ws = &g->ws; t.trace = trace;
  memcpy(t.guess, guess ? guess : I16, sizeof(I16));
  std::vector<Task*> tasks{&t};
  st = run_tasks(c, tasks, 1, /*rebuild_index=*/!g->tgt->has_index);
  *out = t.result;
  if (st) { out->status = st; return st; }
  memcpy(g->last_T, out->T, sizeof(I16));
  g->have_result = true;
  g->dbg_ready = true;
  if (aligned_out) {  // pcl::transformPointCloud(*input_, output, final_transformation_) (gicp.hpp:586)
    float T12[12];
    // (on the no-neighbour return of gicp.hpp:504-506 line 586 is never reached: `output` is still guess * input from line 440)
    fill_T12(out->status == LH_ENO_NN ? t.guess : out->T, T12);
    { ProfScope p(c, "transform", 32.0 * g->src->n); launch_transform(g->src->xyz, nullptr, g->src->n, T12, g->ws.out_xyz, nullptr, c->stream); }
    std::vector<float> host((size_t)g->src->n * 4);
    HIPCHK(hipMemcpyAsync(host.data(), g->ws.out_xyz, sizeof(float) * host.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < g->src->n; i++) memcpy((char*)aligned_out + (size_t)i * stride + off_xyz, &host[4 * (size_t)i], 12);
  }
  return (lh_status)out->status;
}

static lh_status nn1_device(lh_ctx* c, lh_cloud* target, const float4* q, int nq, const float* T16, int32_t* idx, float* d2,
                            double* fitness_sum) {
  if (!target->has_index) { lh_status st = cloud_build_index(target); if (st) return st; }
  int32_t* d_idx = nullptr;
  float* d_d2 = nullptr;
  double* d_part = nullptr;
  DevGuard guard;
  HIPCHK(guard.alloc(&d_idx, sizeof(int32_t) * (size_t)nq));
  HIPCHK(guard.alloc(&d_d2, sizeof(float) * (size_t)nq));
  float T12[12];
  if (T16) fill_T12(T16, T12);
  { ProfScope p(c, "nn1", 24.0 * nq); launch_nn1(q, nq, T16 ? T12 : nullptr, target->view(), d_idx, d_d2, c->stream); }
  lh_status rc = LH_OK;
  if (fitness_sum) {
    int nb = sum_blocks(nq);
    rc = ctx_ensure_small(c, (size_t)nb * 2);
    if (!rc) {
      HIPCHK(guard.alloc(&d_part, sizeof(double) * 2 * (size_t)nb));
      launch_sum_f32(d_d2, d_idx, nq, d_part, c->stream);
      HIPCHK(hipMemcpyAsync(c->small_host, d_part, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      double s = 0, k = 0;
      for (int b = 0; b < nb; b++) { s += c->small_host[2 * b]; k += c->small_host[2 * b + 1]; }
      fitness_sum[0] = s;   // sum of d2 over the queries that found a neighbour ...
      fitness_sum[1] = k;   // ... and how many did (a non-finite query point finds none)
    }
  }
  if (idx) HIPCHK(hipMemcpyAsync(idx, d_idx, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
  if (d2) HIPCHK(hipMemcpyAsync(d2, d_d2, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return rc;
}

lh_status lh_set_allreduce(lh_ctx* ctx, lh_allreduce_fn fn, void* user) {
  if (!ctx) return LH_EINVAL;
  ctx->reduce_fn = fn;
  ctx->reduce_user = user;
  return LH_OK;
}

lh_status lh_gicp_fitness(lh_gicp* g, double* fitness) {
  if (!g || !fitness || !g->src || !g->tgt || !g->have_result) return LH_EINVAL;
  HIPCHK(hipSetDevice(g->ctx->device));
  double sn[2] = {0.0, 0.0};
  lh_status st = nn1_device(g->ctx, g->tgt, g->src->xyz, g->src->n, g->last_T, nullptr, nullptr, sn);
  if (st) return st;
  if (g->ctx->reduce_fn && g->ctx->reduce_fn(sn, 2, g->ctx->reduce_user) != 0) return LH_EDEVICE;
  // mean over the queries that found a neighbour (every finite query does: max_range = DBL_MAX); none -> DBL_MAX like PCL
  *fitness = sn[1] > 0 ? sn[0] / sn[1] : DBL_MAX;
  return LH_OK;
}

lh_status lh_nn1(lh_gicp* g, const lh_cloud_view* q, int32_t* idx, float* d2) {
  if (!g || !g->tgt || !q) return LH_EINVAL;
  HIPCHK(hipSetDevice(g->ctx->device));
  lh_cloud* qc = nullptr;
  lh_status st = upload_view(g->ctx, q, &qc);
  if (st) return st;
  st = nn1_device(g->ctx, g->tgt, qc->xyz, qc->n, nullptr, idx, d2, nullptr);
  cloud_free(qc);
  return st;
}
lh_status lh_nn1_cloud(lh_cloud* target, const lh_cloud* q, int32_t* idx, float* d2) {
  if (!target || !q || target->ctx != q->ctx) return LH_EINVAL;
  HIPCHK(hipSetDevice(target->ctx->device));
  return nn1_device(target->ctx, target, q->xyz, q->n, nullptr, idx, d2, nullptr);
}
lh_status lh_knn_cloud(lh_cloud* target, const lh_cloud* q, int k, int32_t* idx, float* d2) {
  if (!target || !q || target->ctx != q->ctx || k < 1 || k > 64) return LH_EINVAL;
  lh_ctx* c = target->ctx;
  HIPCHK(hipSetDevice(c->device));
  if (!target->has_index) { lh_status st = cloud_build_index(target); if (st) return st; }
  int32_t* d_idx = nullptr;
  float* d_d2 = nullptr;
  size_t cnt = (size_t)q->n * k;
  HIPCHK(lhMalloc(&d_idx, sizeof(int32_t) * cnt));
  HIPCHK(lhMalloc(&d_d2, sizeof(float) * cnt));
  { ProfScope p(c, "knn", (16.0 + 8.0 * k) * q->n); launch_knn(q->xyz, q->n, target->view(), k, d_idx, d_d2, c->stream); }
  HIPCHK(hipGetLastError());
  if (idx) HIPCHK(hipMemcpyAsync(idx, d_idx, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, c->stream));
  if (d2) HIPCHK(hipMemcpyAsync(d2, d_d2, sizeof(float) * cnt, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)lhFree(d_idx); (void)lhFree(d_d2);
  return LH_OK;
}

// a cloud shaped like `in` (same fields, same size) whose contents are about to be overwritten
static lh_status cloud_like(const lh_cloud* in, lh_cloud** out) {
  lh_cloud* o = new lh_cloud();
  o->ctx = in->ctx; o->n = in->n; o->n_pad = in->n_pad;
  hipError_t e = lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad);
  if (e == hipSuccess && in->nrm) e = lhMalloc(&o->nrm, sizeof(float4) * (size_t)o->n_pad);
  if (e == hipSuccess && in->intensity) e = lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad);
  if (e != hipSuccess) { cloud_free(o); return LH_ENOMEM; }
  *out = o;
  return LH_OK;
}

lh_status lh_gicp_align_batch_out(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src, lh_cloud* const* tgt,
                                  const float* guesses, lh_gicp_result* out, lh_cloud** aligned, int max_in_flight) {
  if (!ctx || !p || n_pairs < 0 || !src || !tgt || !out) return LH_EINVAL;
  if (n_pairs == 0) return LH_OK;
  HIPCHK(hipSetDevice(ctx->device));
  int in_flight = max_in_flight > 0 ? max_in_flight : 64;  // default: enough pairs in flight for four scheduler groups
  in_flight = std::min(in_flight, n_pairs);
  int max_n = 1;
  for (int i = 0; i < n_pairs; i++) {
    if (!src[i] || !tgt[i] || src[i]->ctx != ctx || tgt[i]->ctx != ctx) return LH_EINVAL;
    if (aligned && aligned[i] && (aligned[i]->ctx != ctx || aligned[i]->n != src[i]->n || aligned[i] == src[i] || aligned[i] == tgt[i])) return LH_EINVAL;
    max_n = std::max(max_n, src[i]->n);
  }
  if (aligned)  // a caller-supplied output cloud is overwritten: whatever index / covariances it carried describe the OLD coordinates
    for (int i = 0; i < n_pairs; i++)
      if (aligned[i]) { aligned[i]->has_index = false; aligned[i]->cov_k = 0; }
  lh_status st = ctx_ensure_slots(ctx, in_flight, max_n);
  if (st) return st;
  if ((int)ctx->slot_ws.size() < in_flight) ctx->slot_ws.resize(in_flight);  // grow-only, reused across calls, freed by lh_destroy
  std::vector<int> created;  // entries of `aligned` made by this call: released again if an allocation fails
  if (aligned)
    for (int i = 0; i < n_pairs; i++)
      if (!aligned[i]) {
        st = cloud_like(src[i], &aligned[i]);
        if (st) {
          for (int k : created) { cloud_free(aligned[k]); aligned[k] = nullptr; }
          aligned[i] = nullptr;
          return st;
        }
        created.push_back(i);
      }
  std::vector<Task> tasks(n_pairs);
  std::vector<Task*> ptrs(n_pairs);
  for (int i = 0; i < n_pairs; i++) {
    Task& t = tasks[i];
    t.P = *p; t.src = src[i]; t.tgt = tgt[i]; t.trace = nullptr;
    t.aligned = aligned ? aligned[i] : nullptr;
    memcpy(t.guess, guesses ? guesses + 16 * (size_t)i : I16, sizeof(I16));
    ptrs[i] = &t;
  }
  st = run_tasks(ctx, ptrs, in_flight, /*rebuild_index=*/true, &ctx->slot_ws);
  for (int i = 0; i < n_pairs; i++) out[i] = tasks[i].result;
  if (aligned) {  // the output clouds were written on the scheduler streams: complete before the caller touches them
    (void)hipStreamSynchronize(ctx->stream);
    ctx->sync_side_streams();
  }
  return st;
}

lh_status lh_gicp_align_batch(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src, lh_cloud* const* tgt,
                              const float* guesses, lh_gicp_result* out, int max_in_flight) {
  return lh_gicp_align_batch_out(ctx, p, n_pairs, src, tgt, guesses, out, nullptr, max_in_flight);
}

int lh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// Independent scan pairs over several GPUs from ONE process (SURVEY 8b "shards over visible GPUs", 8e): the pairs need no
// exchange step, so there is no collective -- one host thread per device drives that device's context(s) with the pairs whose
// clouds live there.  Contexts that share a device are served by the same thread, one after the other (the buffer pool's
// one-context-at-a-time contract, locus_hip.h).
lh_status lh_gicp_align_batch_multi(int n_ctx, lh_ctx* const* ctxs, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src,
                                    lh_cloud* const* tgt, const float* guesses, lh_gicp_result* out, lh_cloud** aligned, int max_in_flight) {
  if (n_ctx < 1 || !ctxs || !p || n_pairs < 0 || !src || !tgt || !out) return LH_EINVAL;
  for (int d = 0; d < n_ctx; d++) {
    if (!ctxs[d]) return LH_EINVAL;
    for (int e = 0; e < d; e++)
      if (ctxs[e] == ctxs[d]) return LH_EINVAL;
  }
  if (n_pairs == 0) return LH_OK;
  std::vector<std::vector<int>> mine(n_ctx);
  for (int i = 0; i < n_pairs; i++) {
    if (!src[i] || !tgt[i] || src[i]->ctx != tgt[i]->ctx) return LH_EINVAL;
    int d = 0;
    while (d < n_ctx && ctxs[d] != src[i]->ctx) d++;
    if (d == n_ctx) return LH_EINVAL;  // a pair whose clouds live on none of the given contexts
    mine[d].push_back(i);
  }
  std::vector<lh_status> rc(n_ctx, LH_OK);
  auto run_ctx = [&](int d) {
    const std::vector<int>& idx = mine[d];
    if (idx.empty()) return;
    const int k = (int)idx.size();
    std::vector<lh_cloud*> s(k), t(k), a(k, nullptr);
    std::vector<float> g;
    std::vector<lh_gicp_result> r(k);
    if (guesses) g.resize((size_t)16 * k);
    for (int j = 0; j < k; j++) {
      s[j] = src[idx[j]]; t[j] = tgt[idx[j]];
      if (aligned) a[j] = aligned[idx[j]];
      if (guesses) memcpy(&g[(size_t)16 * j], guesses + (size_t)16 * idx[j], sizeof(float) * 16);
    }
    rc[d] = lh_gicp_align_batch_out(ctxs[d], p, k, s.data(), t.data(), guesses ? g.data() : nullptr, r.data(), aligned ? a.data() : nullptr, max_in_flight);
    for (int j = 0; j < k; j++) {
      out[idx[j]] = r[j];
      if (aligned) aligned[idx[j]] = a[j];
    }
  };
  std::vector<int> devices;  // one thread per distinct device
  for (int d = 0; d < n_ctx; d++)
    if (std::find(devices.begin(), devices.end(), ctxs[d]->device) == devices.end()) devices.push_back(ctxs[d]->device);
  auto run_device = [&](int dev) {
    for (int d = 0; d < n_ctx; d++)
      if (ctxs[d]->device == dev) run_ctx(d);
  };
  std::vector<std::thread> threads;
  for (size_t k = 1; k < devices.size(); k++) threads.emplace_back(run_device, devices[k]);
  run_device(devices[0]);  // the calling thread takes the first device
  for (auto& th : threads) th.join();
  for (int d = 0; d < n_ctx; d++)
    if (rc[d]) return rc[d];
  return LH_OK;
}

// the same from host-resident clouds (what a C++ ROS node holds: PCL point arrays): pair i goes to context i * n_ctx / n_pairs
// (contiguous blocks, so that consecutive scans of an odometry stream -- target of pair i = source of pair i - 1 -- are
// uploaded once per device), is aligned there, and only the results come back
lh_status lh_gicp_align_batch_multi_views(int n_ctx, lh_ctx* const* ctxs, const lh_gicp_params* p, int n_pairs, const lh_cloud_view* src,
                                          const lh_cloud_view* tgt, const float* guesses, lh_gicp_result* out, int max_in_flight) {
  if (n_ctx < 1 || !ctxs || !p || n_pairs < 0 || !src || !tgt || !out) return LH_EINVAL;
  for (int d = 0; d < n_ctx; d++)
    if (!ctxs[d]) return LH_EINVAL;
  if (n_pairs == 0) return LH_OK;
  std::vector<lh_cloud*> S(n_pairs, nullptr), T(n_pairs, nullptr), owned;
  lh_status st = LH_OK;
  auto same = [](const lh_cloud_view& a, const lh_cloud_view& b) {
    return a.base == b.base && a.count == b.count && a.stride == b.stride && a.off_xyz == b.off_xyz && a.off_normal == b.off_normal;
  };
  for (int i = 0; i < n_pairs && !st; i++) {
    const int d = (int)(((long)i * n_ctx) / n_pairs);
    const bool chained = i > 0 && (int)(((long)(i - 1) * n_ctx) / n_pairs) == d && same(tgt[i], src[i - 1]);
    st = upload_view(ctxs[d], &src[i], &S[i]);
    if (!st) owned.push_back(S[i]);
    if (!st) {
      if (chained) T[i] = S[i - 1];  // the previous scan is already on this device
      else { st = upload_view(ctxs[d], &tgt[i], &T[i]); if (!st) owned.push_back(T[i]); }
    }
  }
  if (!st) st = lh_gicp_align_batch_multi(n_ctx, ctxs, p, n_pairs, S.data(), T.data(), guesses, out, nullptr, max_in_flight);
  for (lh_cloud* c : owned) {
    (void)hipSetDevice(c->ctx->device);
    cloud_free(c);
  }
  return st;
}

// ---- building blocks for parity tests ---------------------------------------------------------------------
lh_status lh_cov_knn(lh_cloud* c, int k, double gicp_epsilon, double* cov9_out) {
  if (!c || !cov9_out) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->ctx->device));
  lh_status st = cloud_ensure_cov(c, k, gicp_epsilon);
  if (st) return st;
  std::vector<double> planes((size_t)6 * c->n_pad);
  HIPCHK(hipMemcpyAsync(planes.data(), c->cov6, sizeof(double) * planes.size(), hipMemcpyDeviceToHost, c->ctx->stream));
  HIPCHK(hipStreamSynchronize(c->ctx->stream));
  for (int i = 0; i < c->n; i++) {
    double s6[6];
    for (int q = 0; q < 6; q++) s6[q] = planes[(size_t)q * c->n_pad + i];
    sym6_to_mat9(s6, cov9_out + 9 * (size_t)i);
  }
  return LH_OK;
}

lh_status lh_gicp_debug_sweep(lh_gicp* g, const float T[16], const float guess[16], int32_t* tgt_idx, double* maha9) {
  if (!g || !g->src || !g->tgt || !T) return LH_EINVAL;
  lh_ctx* c = g->ctx;
  HIPCHK(hipSetDevice(c->device));
  lh_status st = ctx_ensure_slots(c, 1, g->src->n);
  if (st) return st;
  Task& t = g->task;
  t.P = g->P; t.src = g->src; t.tgt = g->tgt; t.ws = &g->ws; t.trace = nullptr; t.slot = 0;
  memcpy(t.guess, guess ? guess : I16, sizeof(I16));
  t.count_stats = true;
  t.stream = nullptr;
  if (!g->dbg_prepared) {  // first debug sweep after a change of clouds: cold state; later ones are warm (like align's sweeps)
    st = task_prepare(c, &t, !g->tgt->has_index);
    if (st) return st;
    g->dbg_prepared = true;
  }
  SweepArgs a;
  a.njobs = 1; a.bpj = 0; a.max_depth = 0; a.pad = 0; a.job[0].slot = 0; a.job[0].pad = 0;
  Task::T16_to_T12(T, a.job[0].T);
  { ProfScope p(c, "nn_sweep", 252.0 * g->src->n); launch_sweep(c->descs_dev, a, g->src->n, c->stream); }
  HIPCHK(hipGetLastError());
  int n = g->src->n;
  std::vector<float> corr((size_t)n * 4);
  std::vector<double> planes((size_t)6 * g->ws.n_pad);
  HIPCHK(hipMemcpyAsync(corr.data(), g->ws.corr, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(planes.data(), g->ws.maha6, sizeof(double) * planes.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) {
    int32_t j;
    memcpy(&j, &corr[4 * (size_t)i + 3], 4);
    if (tgt_idx) tgt_idx[i] = j < 0 ? -1 : j;   // (-2 marks "no neighbour at all" for the cost kernels; unmatched either way)
    if (maha9) {
      double s6[6];
      for (int q = 0; q < 6; q++) s6[q] = planes[(size_t)q * g->ws.n_pad + i];
      sym6_to_mat9(s6, maha9 + 9 * (size_t)i);
    }
  }
  g->dbg_ready = true;
  return LH_OK;
}

lh_status lh_gicp_debug_sweep_fused(lh_gicp* g, const float T[16], int sweep_index, int32_t* tgt_idx, uint64_t* walks, double* sums74) {
  if (!g || !g->src || !g->tgt || !T || sweep_index < 0) return LH_EINVAL;
  if (!g->src->nrm || !g->tgt->nrm) return LH_EINVAL;   // the cost_mode 1 kernels of the production configuration: covariances from normals
  lh_ctx* c = g->ctx;
  HIPCHK(hipSetDevice(c->device));
  lh_status st = ctx_ensure_slots(c, 1, g->src->n);
  if (st) return st;
  Task& t = g->task;
  t.P = g->P; t.P.cost_mode = 1; t.P.recompute_source_cov = 0; t.P.recompute_target_cov = 0;
  t.src = g->src; t.tgt = g->tgt; t.ws = &g->ws; t.trace = nullptr; t.slot = 0;
  memcpy(t.guess, I16, sizeof(I16));
  t.count_stats = false;
  t.stream = nullptr;
  if (sweep_index == 0 || !g->dbg_prepared) {  // cold: descriptor, index, and the seed pass below
    st = task_prepare(c, &t, !g->tgt->has_index);
    if (st) return st;
    g->dbg_prepared = true;
  }
  SweepArgs a;
  a.njobs = 1; a.bpj = 0; a.max_depth = 0; a.pad = 0; a.job[0].slot = 0; a.job[0].pad = 0;
  Task::T16_to_T12(T, a.job[0].T);
  if (sweep_index == 0) {
    SweepArgs sa = a;
    launch_seed(c->descs_dev, sa, g->src->n, c->stream);
  }
  CostArgs ca;
  ca.njobs = 1; ca.pad = 0; ca.job[0].slot = 0; ca.job[0].out_offset = 0;
  memcpy(ca.job[0].T, a.job[0].T, sizeof(a.job[0].T));
  launch_sweep_fused(c->descs_dev, a, sweep_is_split(&t, sweep_index) ? 1u : 0u, g->src->n, c->mom_partials_dev, c->mom_stride, nullptr, true, c->wmask_dev,
                     c->mask_stride, c->stream);
  launch_moments_final(c->descs_dev, ca, c->mom_partials_dev, c->mom_stride, c->partials_host, nullptr, c->wmask_dev, c->mask_stride, c->stream);
  HIPCHK(hipGetLastError());
  if (tgt_idx) HIPCHK(hipMemcpyAsync(tgt_idx, g->ws.prev_nn, sizeof(int32_t) * (size_t)g->src->n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double w = 0.0;
  for (int ch = 0; ch < FINAL_CHUNKS; ch++) w += c->partials_host[ch * MOM_ROW + MOM_NSUM];
  if (walks) *walks = (uint64_t)w;
  if (sums74)
    for (int k = 0; k < MOM_NSUM; k++) {
      double v = 0.0;
      for (int ch = 0; ch < FINAL_CHUNKS; ch++) v += c->partials_host[ch * MOM_ROW + k];
      sums74[k] = v;
    }
  g->dbg_ready = false;   // (the correspondence buffers of lh_gicp_debug_sweep were not written)
  return LH_OK;
}

lh_status lh_gicp_debug_stats(lh_gicp* g, uint64_t out[2], int reset) {
  if (!g || !out) return LH_EINVAL;
  out[0] = out[1] = 0;
  if (!g->ws.stats) return LH_OK;  // nothing swept yet
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  HIPCHK(hipMemcpy(out, g->ws.stats, 16, hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(g->ws.stats, 0, 16));
  return LH_OK;
}

lh_status lh_debug_traversal_stats(lh_cloud* target, const lh_cloud* q, const float T[16], const int32_t* cand, int leaf_prescan,
                                   uint64_t out[5]) {
  if (!target || !q || !out || target->ctx != q->ctx) return LH_EINVAL;
  lh_ctx* c = target->ctx;
  HIPCHK(hipSetDevice(c->device));
  if (!target->has_index) { lh_status st = cloud_build_index(target); if (st) return st; }
  unsigned long long* d = nullptr;
  HIPCHK(lhMalloc(&d, 8 * 24));
  HIPCHK(hipMemsetAsync(d, 0, 8 * 24, c->stream));
  float T12[12];
  if (T) fill_T12(T, T12);
  int32_t* d_cand = nullptr;
  if (cand) {
    HIPCHK(lhMalloc(&d_cand, sizeof(int32_t) * (size_t)q->n));
    HIPCHK(hipMemcpyAsync(d_cand, cand, sizeof(int32_t) * (size_t)q->n, hipMemcpyHostToDevice, c->stream));
  }
  launch_nn1_stats(q->xyz, q->n, T ? T12 : nullptr, target->view(), target->xyz, d_cand, d, c->stream);
  (void)leaf_prescan;  // an experiment of the implicit-tree layout; ignored
  HIPCHK(hipMemcpyAsync(out, d, 40, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (getenv("LH_STATS_LEVELS")) {
    unsigned long long lv[24];
    HIPCHK(hipMemcpy(lv, d, sizeof(lv), hipMemcpyDeviceToHost));
    fprintf(stderr, "[locus_hip] node visits per query by level:");
    for (int l = 0; l < MAX_DEPTH; l++) fprintf(stderr, " %.2f", (double)lv[8 + l] / q->n);
    fprintf(stderr, "\n");
  }
  (void)lhFree(d);
  (void)lhFree(d_cand);
  return LH_OK;
}

lh_status lh_gicp_debug_cost(lh_gicp* g, const double x[6], double* f, double g6[6], double sums13[13], int* m) {
  if (!g || !g->dbg_ready || !x) return LH_EINVAL;
  lh_ctx* c = g->ctx;
  HIPCHK(hipSetDevice(c->device));
  float T16[16];
  apply_state(x, T16);
  CostArgs a;
  a.njobs = 1; a.pad = 0; a.job[0].slot = 0; a.job[0].out_offset = 0;
  Task::T16_to_T12(T16, a.job[0].T);
  { ProfScope p(c, "cost_fdf", 108.0 * g->src->n); launch_cost(c->descs_dev, a, g->src->n, c->mom_partials_dev, c->mom_stride, c->partials_host, c->stream); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  double S[COST_NSUM];
  for (int k = 0; k < COST_NSUM; k++) S[k] = c->partials_host[k];
  if (sums13) memcpy(sums13, S, sizeof(double) * 13);
  if (m) *m = (int)S[13];
  double ff = 0, gg[6] = {0, 0, 0, 0, 0, 0};
  if (S[13] > 0) cost_finish(S, S[13], x, &ff, gg);
  if (f) *f = ff;
  if (g6) memcpy(g6, gg, sizeof(gg));
  return LH_OK;
}

// ---- K8 / H2 -----------------------------------------------------------------------------------------------
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
  if (!c || k < 3 || k > 64) return LH_EINVAL;
  lh_ctx* x = c->ctx;
  HIPCHK(hipSetDevice(x->device));
  if (!c->has_index) { lh_status st = cloud_build_index(c); if (st) return st; }
  if (!c->nrm) HIPCHK(lhMalloc(&c->nrm, sizeof(float4) * (size_t)c->n_pad));
  { ProfScope p(x, "knn_normals", (16.0 + 16.0 * k + 16.0) * c->n); launch_knn_normals(c->xyz, c->n, c->view(), k, c->nrm, x->stream); }
  HIPCHK(hipGetLastError());
  return LH_OK;
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
static lh_status voxel_grid_device(lh_ctx* c, const float4* d_in, int n, float leaf, int limit_axis, double lo, double hi,
                                   float4** d_out, uint32_t* total_out, const float4* d_nrm = nullptr, float4** d_out_nrm = nullptr) {
  *d_out = nullptr;
  *total_out = 0;
  if (d_out_nrm) *d_out_nrm = nullptr;
  VoxelSegments vs;
  lh_status st = voxel_segments(c, d_in, n, leaf, limit_axis, lo, hi, &vs);
  if (st) return st;
  if (vs.total > 0) {
    if (lhMalloc(d_out, sizeof(float4) * (size_t)vs.total) != hipSuccess) { vs.release(); return LH_ENOMEM; }
    if (d_nrm && d_out_nrm && lhMalloc(d_out_nrm, sizeof(float4) * (size_t)round_up((int)vs.total, 256)) != hipSuccess) {  // n_pad entries, like every cloud's normals
      (void)lhFree(*d_out); *d_out = nullptr; vs.release(); return LH_ENOMEM;
    }
    ProfScope p(c, "voxel_centroids", (d_nrm ? 64.0 : 32.0) * n);
    launch_voxel_centroids(d_in, d_nrm, c->keys1, c->vals1, vs.heads, vs.rank, n, *d_out, d_out_nrm ? *d_out_nrm : nullptr, vs.total, c->stream);
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
__global__ void __launch_bounds__(256) k_pack_xyzi(const float4* __restrict__ xyz, const float* __restrict__ inten, int n, float4* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = xyz[i];
  out[i] = make_float4(p.x, p.y, p.z, inten ? inten[i] : 0.0f);
}
__global__ void __launch_bounds__(256) k_unpack_xyzi(const float4* __restrict__ in, int n, float4* __restrict__ xyz, float* __restrict__ inten) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in[i];
  xyz[i] = make_float4(p.x, p.y, p.z, 1.0f);
  inten[i] = p.w;
}
static lh_status cloud_voxel_grid(const lh_cloud* in, float leaf, int limit_axis, double lo, double hi, bool all_fields, lh_cloud** out) {
  if (!in || !out || !(leaf > 0.0f) || limit_axis > 2 || in->n <= 0) return LH_EINVAL;
  if (all_fields && !in->nrm) return LH_EINVAL;  // the PointXYZINormal flavour needs the normal / curvature fields
  lh_ctx* c = in->ctx;
  HIPCHK(hipSetDevice(c->device));
  float4 *d_in = nullptr, *d_out = nullptr, *d_out_nrm = nullptr;
  HIPCHK(lhMalloc(&d_in, sizeof(float4) * (size_t)in->n));
  hipLaunchKernelGGL(k_pack_xyzi, dim3((in->n + 255) / 256), dim3(256), 0, c->stream, in->xyz, in->intensity, in->n, d_in);
  uint32_t total = 0;
  lh_status st = voxel_grid_device(c, d_in, in->n, leaf, limit_axis, lo, hi, &d_out, &total, all_fields ? in->nrm : nullptr,
                                   all_fields ? &d_out_nrm : nullptr);
  (void)lhFree(d_in);
  DevGuard guard;
  guard.bufs.push_back(d_out);
  if (st) { (void)lhFree(d_out_nrm); return st; }
  if (total == 0) { (void)lhFree(d_out_nrm); return LH_EINVAL; }  // every point was filtered out: no cloud to return
  lh_cloud* o = new lh_cloud();
  guard.cloud = o;
  o->ctx = c;
  o->n = (int)total;
  o->n_pad = round_up(o->n, 256);
  o->nrm = d_out_nrm;
  HIPCHK(lhMalloc(&o->xyz, sizeof(float4) * (size_t)o->n_pad));
  HIPCHK(lhMalloc(&o->intensity, sizeof(float) * (size_t)o->n_pad));
  hipLaunchKernelGGL(k_unpack_xyzi, dim3((o->n + 255) / 256), dim3(256), 0, c->stream, d_out, o->n, o->xyz, o->intensity);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = guard.keep_cloud();
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
__global__ void __launch_bounds__(256) k_gather_cloud(const float4* __restrict__ xyz, const float4* __restrict__ nrm, const float* __restrict__ inten,
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
