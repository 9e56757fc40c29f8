// lh_runtime.hpp -- the host runtime's shared types (one lh_ctx per GPU, device clouds, the scheduler's Task) and the functions the
// translation units of liblocus_hip.so call across: lh_pool.hip (device-memory pool, context scratch), lh_index.hip (K2 index build,
// k-NN covariances), lh_sched.hip (the two schedulers: host-driven coroutines, device-driven loop), lh_api.hip (C ABI: context, clouds,
// registration object, batches, debug entry points), lh_filters.hip (C ABI: K8 / K3 / K1 filters, NDT, body filter, local map,
// profiling).  Internal: nothing here is part of the ABI (include/locus_hip.h is).
#pragma once
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


// ---- device-memory pool (lh_pool.hip) -----------------------------------------------------------------------------------------
hipError_t lhMallocRaw(void** p, size_t bytes);
template <class T>
static hipError_t lhMalloc(T** p, size_t bytes) { return lhMallocRaw(reinterpret_cast<void**>(p), bytes); }
hipError_t lhFree(void* p);

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
  // (generation << 32) | next index: an index is claimed by a compare-exchange on the WHOLE word, so a worker that is still leaving the
  // previous parallel_for can never take (or skip) an index of the next one -- with a bare counter it could read the new n_items before
  // the counter was reset, run an index that would be handed out again later, and resume the same coroutine twice
  std::atomic<uint64_t> ticket{0};
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
  void drain(uint64_t gen, int n) {
    for (;;) {
      uint64_t t = ticket.load();
      if ((t >> 32) != (gen & 0xffffffffull)) return;   // a later parallel_for has begun: nothing of ours is left
      const uint32_t idx = (uint32_t)t;
      if ((int)idx >= n) return;
      if (!ticket.compare_exchange_weak(t, t + 1)) continue;
      fn((int)idx);   // (fn is not reassigned before every claimed index of this generation has finished: pending > 0 until then)
      std::lock_guard<std::mutex> l(m);
      if (--pending == 0) cv_done.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      int n;
      {
        std::unique_lock<std::mutex> l(m);
        cv_work.wait(l, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
        n = n_items;
      }
      drain(seen, n);
    }
  }
  void parallel_for(int n, std::function<void(int)> f) {
    if (n <= 0) return;
    if (workers.empty() || n == 1) { for (int i = 0; i < n; i++) f(i); return; }
    uint64_t gen;
    {
      std::lock_guard<std::mutex> l(m);
      fn = std::move(f);
      n_items = n;
      pending = n;
      gen = ++generation;
      ticket.store((gen & 0xffffffffull) << 32);
    }
    cv_work.notify_all();
    drain(gen, n);
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
  float* rec = nullptr;       // the neighbour prev_nn points at, gathered: positions then normals, packed triples (PairDesc::rec)
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
  // Batched index build scratch, in IDX_SETS independent sets: a build uses one set from its first launch to its last, and builds on the same
  // set are chained by the set's event.  The scheduler gives every group its own set (two groups share one beyond sixteen), so the sixteen
  // builds that open a 512-pair step run side by side on their groups' streams instead of one after the other on ONE shared scratch --
  // measured: the 20-iteration headline unchanged (the GPU is full either way), production stopping 20.7 k -> 21.1 k pairs/s
  // (round 4; about 0.3 GB per set at 32 x 100 k points, allocated on first use).
  struct IndexScratch {
    uint64_t *k64a = nullptr, *k64b = nullptr;
    uint32_t *v32a = nullptr, *v32b = nullptr, *bbox = nullptr;
    uint64_t *k32a = nullptr, *k32b = nullptr;   // the build's radix sort: (key, index) pairs in flight between its passes
    uint32_t* rs_hist = nullptr;                 // ... and its per-tile digit tables
    void* sort64_temp = nullptr;
    size_t sort64_temp_bytes = 0;
    char* tree_tmp = nullptr;        // TreeScratch arrays (TREE_SCRATCH_BYTES_PER_POINT per point)
    int cap = 0;
    static constexpr int STAGES = 4; // staging ring of the build's descriptors: a build never waits for the upload of the build before it
    IndexDesc *descs_dev = nullptr, *descs_host = nullptr;   // host: STAGES x MAX_INDEX_BATCH entries (pinned)
    hipEvent_t copy_done[STAGES] = {};
    hipEvent_t build_done = nullptr;
    int stage = 0;
  };
  static constexpr int IDX_SETS = 16;
  IndexScratch idx_sets[IDX_SETS];
  // K3, the block k-NN search over a batch of clouds (lh_index.hip knn_block_batch): the launch's descriptor table and its redo list
  KnnCloudDesc* knn_descs_dev = nullptr;   // [MAX_INDEX_BATCH]
  uint32_t* knn_redo_cnt = nullptr;
  uint2* knn_redo = nullptr;
  long knn_redo_cap = 0;                    // points the redo list has room for
  float* knn_soa = nullptr;                 // the batch's sorted coordinates as separate x / y / z arrays (3 x (points + LEAF_CAP per cloud))
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
  hipEvent_t entry_ev = nullptr;       // fork point of a scheduler run: the side streams wait for what earlier calls left on `stream`
  // batch mode: one workspace per scheduler slot.  They live here (not in a thread-local) so that they are tied to this
  // context's device, reused by every thread that drives the context, and released by lh_destroy.
  std::vector<Workspace> slot_ws;
  // misc pinned scratch for small downloads
  double* small_host = nullptr;
  size_t small_host_doubles = 0;
  // source-sharded single pair: in-place sum of the cost/moment sums over the ranks that hold the other shards
  lh_allreduce_fn reduce_fn = nullptr;
  void* reduce_user = nullptr;
  lh_device_allreduce_fn dev_reduce_fn = nullptr;   // ... and its device-side form: the device-driven loop sums the chunk sums over the ranks in HBM, on the iteration's own stream
  void* dev_reduce_user = nullptr;
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
  NodeX* node_buf = nullptr;   // element 0 holds the TreeHeader, the start grid follows (GRID_NODEX slots), then the nodes
  int index_cap = 0;           // points the index buffers were allocated for
  NodeX* nodes() const { return node_buf ? node_buf + 1 + GRID_NODEX : nullptr; }
  TreeHeader* hdr() const { return reinterpret_cast<TreeHeader*>(node_buf); }
  // k-NN covariances (6 planes of n_pad doubles), valid for (cov_k, cov_eps)
  double* cov6 = nullptr;
  int cov_k = 0;
  double cov_eps = 0;
  uint64_t built_epoch = 0;    // the batch call (lh_ctx::epoch) that last built this cloud's index: a target shared by pairs of several
                               // scheduler groups is rebuilt ONCE per call, not once per group admission under the other groups' sweeps
  TreeView view() const { return TreeView{sorted, nodes(), hdr(), n}; }
};


static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
void cloud_free(lh_cloud* c);

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


void runtime_check_streams(lh_ctx* c, int groups);   // lh_api.hip: one-time report when the runtime serialises the scheduler's streams
lh_status ctx_ensure_scratch(lh_ctx* c, int n);
lh_status ctx_ensure_small(lh_ctx* c, size_t doubles);
// K8 with both reductions ended on the device (lh_filters.hip): enqueues the chain on s; out21 (device) = the 21 unique entries of Ap
lh_status p2plane_information_device(const float4* qxyz, int n, const float4* ref_nrm, const int32_t* corr, double* scratch /* 21 sum_blocks(n) + 4 doubles */,
                                     double* out21, hipStream_t s);
lh_status ctx_ensure_slots(lh_ctx* c, int n_slots, int max_n);
// K2 (lh_index.hip): the NN indexes of several clouds by the same launches; k-NN covariances of a cloud
lh_status build_indices(lh_ctx* x, lh_cloud* const* clouds, int n_clouds, hipStream_t s_in = nullptr, int set = 0);   // set: which of lh_ctx::idx_sets the build uses
// Makes stream s wait for every index build enqueued so far, whatever set and stream it used.
static inline lh_status wait_index_builds(lh_ctx* x, hipStream_t s) {
  for (lh_ctx::IndexScratch& X : x->idx_sets)
    if (X.build_done && hipStreamWaitEvent(s, X.build_done, 0) != hipSuccess) return LH_EDEVICE;
  return LH_OK;
}
// Calls on a context are issued in order (include/locus_hip.h): whatever earlier calls enqueued on the primary stream without waiting for it
// (lh_normals_knn_batch: index build + normals; uploads) is complete
// before a scheduler group's side stream (non-blocking, otherwise unordered against `stream`) reads a tree, a normal or a cloud.
static inline lh_status fork_side_streams(lh_ctx* x, hipStream_t* const* side, int n_side) {
  if (n_side <= 0) return LH_OK;
  if (!x->entry_ev && hipEventCreateWithFlags(&x->entry_ev, hipEventDisableTiming) != hipSuccess) return LH_EDEVICE;
  if (hipEventRecord(x->entry_ev, x->stream) != hipSuccess) return LH_EDEVICE;
  for (int k = 0; k < n_side; k++)
    if (hipStreamWaitEvent(*side[k], x->entry_ev, 0) != hipSuccess) return LH_EDEVICE;
  return LH_OK;   // (index builds inside a scheduler run are complete when the run returns: every pair retires behind its group's build)
}
static inline lh_status cloud_build_index(lh_cloud* c) { return build_indices(c->ctx, &c, 1); }
extern std::atomic<bool> g_small_index;   // clouds of <= SMALL_INDEX_MAX_N points take the one-launch build (lh_index_small.hip); off: the general build for all
lh_status cloud_ensure_cov(lh_cloud* c, int k, double eps);
// K3 for a batch of clouds: ONE index build for those that have none and ONE block k-NN launch per MAX_INDEX_BATCH clouds.
// mode = KNN_MODE_NORMALS (fills lh_cloud::nrm), KNN_MODE_COV (lh_cloud::cov6) or KNN_MODE_RAW (one cloud: idx_dev / d2_dev, n * k each)
lh_status knn_block_batch(lh_ctx* x, lh_cloud* const* clouds, int n_clouds, int k, int mode, double eps, int32_t* idx_dev = nullptr,
                          float* d2_dev = nullptr);

// one alignment = one coroutine
enum Req { REQ_NONE = 0, REQ_SWEEP, REQ_COST, REQ_DONE };

struct Task;
inline thread_local Task* g_boot_task = nullptr;

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
    const OuterParams OP{P.max_iterations, P.max_inner_iterations, P.rotation_epsilon, P.transformation_epsilon, P.bfgs_quad_curv};
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


// lh_sched.hip
lh_status task_prepare(lh_ctx* c, Task* t, bool rebuild_index, bool upload_desc = true);
bool sweep_is_split(const Task* t, int k);
lh_status run_tasks(lh_ctx* c, std::vector<Task*>& tasks, int in_flight, bool rebuild_index, std::vector<Workspace>* slot_ws = nullptr);

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


// host <-> device cloud conversion (lh_api.hip)
lh_status upload_view(lh_ctx* c, const lh_cloud_view* v, lh_cloud** out, bool sync = true);
static inline void fill_T12(const float* T16, float* T12) { Task::T16_to_T12(T16, T12); }
