// lh_api.hip -- C ABI (include/locus_hip.h) of the MI355X GICP hot path: context, device clouds, the registration object, batches, debug
// entry points.  (Filters, NDT, local map, profiling: lh_filters.hip.  Runtime behind them: lh_runtime.hpp, lh_sched.hip, lh_index.hip, lh_pool.hip.)
//
// Runtime model: one lh_ctx per GPU (one HIP stream).  An alignment is a stackful coroutine (ucontext) that runs
// the reference's computeTransformation control flow (gicp.hpp:406-617) and yields two kinds of device requests:
//   SWEEP(T)  -> k_sweep   (NN + Mahalanobis, gicp.hpp:464-498)
//   COST(x)   -> k_cost    (fused f/df pass, gicp.hpp:362-402)
// The scheduler resumes every in-flight pair, batches their requests into ONE launch per kind (grid.y = pair),
// synchronises once per round and feeds the results back.  A single lh_gicp_align is the same machinery with one
// task.  No CPU fallback exists: without a HIP device every entry point returns LH_EDEVICE.
#include <atomic>
#include <chrono>
#include <mutex>
#include "lh_runtime.hpp"

// host <-> device cloud conversion
// sync = false: the copy and the unpack are only queued (on the context's stream); the caller synchronises once for many clouds and
// must keep the host array alive until then
lh_status upload_view(lh_ctx* c, const lh_cloud_view* v, lh_cloud** out, bool sync) {
  if (!v || !v->base || v->count == 0 || v->stride < 12) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->device));
  const bool has_n = v->off_normal != UINT32_MAX, has_i = v->off_intensity != UINT32_MAX;
  const size_t n = v->count;
  // every field that will be read must lie inside a point
  if ((size_t)v->off_xyz + 12 > v->stride || (has_n && (size_t)v->off_normal + 12 > v->stride) || (has_i && (size_t)v->off_intensity + 4 > v->stride) ||
      (has_n && v->off_curvature != UINT32_MAX && (size_t)v->off_curvature + 4 > v->stride))
    return LH_EINVAL;
  lh_cloud* cl = new lh_cloud();
  cl->ctx = c;
  cl->n = (int)n;
  cl->n_pad = round_up(cl->n, 256);
  // the host array goes over AS IT IS (one copy) and is taken apart on the device (k_unpack_view).  Only the bytes up to the end of the LAST
  // point's last field are read: a strided view whose final point ends the caller's buffer (a column slice of a wider record, a stride
  // larger than the fields used) is not over-read by the tail of a stride.
  size_t used = (size_t)v->off_xyz + 12;
  if (has_n) used = std::max(used, (size_t)v->off_normal + 12);
  if (has_i) used = std::max(used, (size_t)v->off_intensity + 4);
  if (has_n && v->off_curvature != UINT32_MAX) used = std::max(used, (size_t)v->off_curvature + 4);
  const size_t raw_bytes = (n - 1) * (size_t)v->stride + used;
  void* raw = nullptr;
  hipError_t e = lhMalloc(&raw, n * (size_t)v->stride);
  if (e == hipSuccess) e = lhMalloc(&cl->xyz, sizeof(float4) * (size_t)cl->n_pad);
  if (e == hipSuccess && has_n) e = lhMalloc(&cl->nrm, sizeof(float4) * (size_t)cl->n_pad);
  if (e == hipSuccess && has_i) e = lhMalloc(&cl->intensity, sizeof(float) * (size_t)cl->n_pad);
  if (e == hipSuccess) e = hipMemcpyAsync(raw, v->base, raw_bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    launch_unpack_view(raw, cl->n, v->stride, v->off_xyz, has_n ? v->off_normal : 0u, has_i ? v->off_intensity : 0u, has_n ? v->off_curvature : UINT32_MAX,
                       cl->xyz, cl->nrm, cl->intensity, c->stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess && sync) e = hipStreamSynchronize(c->stream);
  (void)lhFree(raw);   // (parked: the pool hands it out again in stream order, behind the unpack)
  if (e != hipSuccess) {
    fprintf(stderr, "[locus_hip] cloud upload failed: %s\n", hipGetErrorString(e));
    (void)hipStreamSynchronize(c->stream);
    cloud_free(cl);
    return e == hipErrorOutOfMemory ? LH_ENOMEM : LH_EDEVICE;
  }
  *out = cl;
  return LH_OK;
}


// =========================================================================================================
// ---- the stream-concurrency probe (hidden: only the C ABI below is exported) ----
// one wave that stays resident for `ticks` of the 100-MHz wall clock and says when it ran: the probe's unit of work
__global__ void __launch_bounds__(64) k_probe_spin(unsigned long long ticks, unsigned long long* __restrict__ span) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (span && threadIdx.x == 0) { span[0] = t0; span[1] = wall_clock64(); }
}
static std::mutex g_probe_mu;
static double g_probe_concurrency[64] = {};   // per device; 0 = not measured yet, < 0 = the probe failed (remembered: a failing probe is not paid for again)
// How many of sixteen one-wave kernels on the CONTEXT'S OWN sixteen streams are resident at the same instant (their own wall-clock stamps; the
// largest overlap of the sixteen intervals).  The scheduler's streams, not new ones: the runtime hands hardware queues to streams as they are
// created, and a process with more streams than queues runs far worse than one with fewer (round 4: 24 groups 3.8 k pairs/s against 12.2 k
// with 16) -- a probe that created sixteen streams of its own next to the scheduler's cost exactly that.
static double probe_stream_concurrency(lh_ctx* c) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  const int device = c->device;
  if (device >= 0 && device < 64 && g_probe_concurrency[device] != 0) return g_probe_concurrency[device] > 0 ? g_probe_concurrency[device] : 0.0;
  auto failed = [&]() { if (device >= 0 && device < 64) g_probe_concurrency[device] = -1.0; return 0.0; };
  constexpr int NS = 16;
  hipStream_t* side[lh_ctx::MAX_GROUPS - 1] = {&c->stream2, &c->stream3, &c->stream4};
  for (int k = 0; k < lh_ctx::MAX_GROUPS - 4; k++) side[3 + k] = &c->stream_more[k];
  hipStream_t st[NS];
  st[0] = c->stream;
  for (int k = 1; k < NS; k++) {
    if (!*side[k - 1] && hipStreamCreateWithFlags(side[k - 1], hipStreamNonBlocking) != hipSuccess) return failed();
    st[k] = *side[k - 1];
  }
  unsigned long long* span = nullptr;
  if (hipMalloc(&span, sizeof(unsigned long long) * 2 * NS) != hipSuccess) return failed();
  const unsigned long long ticks = 200000ull;   // 2 ms of the 100-MHz clock: sixteen launches are issued in a fraction of that
  bool ok = true;
  for (int k = 0; k < NS; k++) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, st[k], 100ull, (unsigned long long*)nullptr);   // warm-up: code object, queues
  for (int k = 0; k < NS; k++) ok = ok && hipStreamSynchronize(st[k]) == hipSuccess;
  for (int k = 0; k < NS; k++) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, st[k], ticks, span + 2 * k);
  for (int k = 0; k < NS; k++) ok = ok && hipStreamSynchronize(st[k]) == hipSuccess;
  unsigned long long h[2 * NS] = {};
  ok = ok && hipMemcpy(h, span, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(span);
  (void)hipGetLastError();
  double conc = 0.0;
  if (ok) {   // the largest number of intervals [start, end) that contain a common instant: try every start
    int best = 0;
    for (int a = 0; a < NS; a++) {
      int n = 0;
      for (int b = 0; b < NS; b++) n += (h[2 * b] <= h[2 * a] && h[2 * a] < h[2 * b + 1]) ? 1 : 0;
      best = std::max(best, n);
    }
    conc = (double)best;
  }
  if (device >= 0 && device < 64) g_probe_concurrency[device] = conc > 0 ? conc : -1.0;
  return conc;
}
// called by the scheduler the first time a batch spreads over more than four streams: ONE measurement per device and process (2-8 ms: sixteen
// 2-ms kernels and their synchronisation), success or failure; every later batch returns at the flag
void runtime_check_streams(lh_ctx* c, int groups) {
  static std::atomic<bool> said{false};
  static std::atomic<unsigned long long> checked{0ull};   // one bit per device
  if (groups <= 4 || said.load()) return;
  const unsigned long long bit = (c->device >= 0 && c->device < 64) ? (1ull << c->device) : 0ull;
  if (bit && (checked.fetch_or(bit) & bit)) return;
  const double conc = probe_stream_concurrency(c);
  if (conc > 0 && conc < 0.7 * std::min(groups, 16) && !said.exchange(true)) {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    fprintf(stderr, "[locus_hip] of %d scheduler streams only about %.0f run at once: the HIP runtime gave this process too few hardware queues (GPU_MAX_HW_QUEUES=%s, read at "
                    "the process's first HIP call).  Batches of >= 64 pairs run about 20 %% slower than they could: call lh_runtime_init(0) before the first HIP "
                    "call, or export GPU_MAX_HW_QUEUES=24 (INTEGRATION.md section 5).\n", groups, conc, e ? e : "unset");
  }
}

#pragma GCC visibility push(default)   // the C ABI is the library's ONLY exported surface (the TUs are compiled -fvisibility=hidden)
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
  p->bfgs_quad_curv = 0;             // GSL's curvature test (see the header)
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

// The scheduler keeps up to sixteen groups of pairs in flight, each on its own HIP stream, and counts on their kernels overlapping (one
// group's single-workgroup k_solve under the other groups' sweeps).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues --
// four by default -- and streams that share a queue run one after the other.  Measured on the 512-pair bench queue: 4 / 8 / 16 / 24 / 32 / 64
// queues -> 9 740 / 10 150 / 10 990 / 12 220 / 12 140 / 12 180 pairs/s (round 4).  The runtime reads the variable when it initialises, i.e. at
// the process's first HIP call.  Round 4 set it from a load-time constructor; a library must not edit its host's environment behind its
// back (setenv races getenv in other threads, and it changes the queues of every other HIP user of the process -- the kind of side effect
// SURVEY 8b told this build not to copy from omp_set_num_threads, gicp.h:138).  Now: lh_runtime_init() is the explicit form, called by the
// host before its first HIP call and before it starts threads; lh_runtime_info() MEASURES what the process got, and the scheduler says so
// once on stderr when a batch wants more concurrent streams than the runtime gives it.
lh_status lh_runtime_init(int max_hw_queues) {
  if (max_hw_queues < 0 || max_hw_queues > 128) return LH_EINVAL;
  char buf[16];
  snprintf(buf, sizeof buf, "%d", max_hw_queues ? max_hw_queues : 24);
  return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? LH_OK : LH_EINVAL;   // (overwrite = 0: a value the deployment chose stays)
}

lh_status lh_runtime_info(lh_ctx* c, lh_runtime_info_t* out) {
  if (!c || !out) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->device));
  memset(out, 0, sizeof *out);
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  out->hw_queues_env = e ? atoi(e) : -1;
  out->streams_probed = 16;
  out->stream_concurrency = probe_stream_concurrency(c);
  if (!(out->stream_concurrency > 0)) return LH_EDEVICE;
  out->adequate = out->stream_concurrency >= 12.0;
  return LH_OK;
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
  for (lh_ctx::IndexScratch& X : c->idx_sets) {
    (void)lhFree(X.k64a); (void)lhFree(X.k64b); (void)lhFree(X.v32a); (void)lhFree(X.v32b); (void)lhFree(X.sort64_temp);
    (void)lhFree(X.tree_tmp); (void)lhFree(X.k32a); (void)lhFree(X.k32b); (void)lhFree(X.rs_hist);
    (void)lhFree(X.bbox); (void)lhFree(X.descs_dev);
    if (X.descs_host) (void)hipHostFree(X.descs_host);
    for (hipEvent_t e : X.copy_done)
      if (e) (void)hipEventDestroy(e);
    if (X.build_done) (void)hipEventDestroy(X.build_done);
  }
  (void)lhFree(c->knn_descs_dev); (void)lhFree(c->knn_redo_cnt); (void)lhFree(c->knn_redo); (void)lhFree(c->knn_soa);
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
  if (c->entry_ev) (void)hipEventDestroy(c->entry_ev);
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
// debug: the index as it lies in HBM (tests compare the one-launch build of small clouds with the general build, byte for byte)
lh_status lh_debug_index_dump(lh_cloud* c, float* sorted4, void* nodes, uint32_t nodes_capacity, void* header64) {
  if (!c || !sorted4 || !nodes || !header64) return LH_EINVAL;
  HIPCHK(hipSetDevice(c->ctx->device));
  if (!c->has_index) { lh_status st = cloud_build_index(c); if (st) return st; }
  const uint32_t nn = std::min<uint32_t>(nodes_capacity, (uint32_t)c->n);
  HIPCHK(hipMemcpyAsync(sorted4, c->sorted, sizeof(float4) * ((size_t)c->n + LEAF_CAP), hipMemcpyDeviceToHost, c->ctx->stream));
  HIPCHK(hipMemcpyAsync(nodes, c->nodes(), sizeof(NodeX) * (size_t)nn, hipMemcpyDeviceToHost, c->ctx->stream));
  HIPCHK(hipMemcpyAsync(header64, c->hdr(), sizeof(TreeHeader), hipMemcpyDeviceToHost, c->ctx->stream));
  HIPCHK(hipStreamSynchronize(c->ctx->stream));
  return LH_OK;
}
int lh_debug_small_index(int enable) {
  const bool was = g_small_index.load();
  if (enable >= 0) g_small_index.store(enable != 0);
  return was ? 1 : 0;
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

lh_status lh_set_device_allreduce(lh_ctx* ctx, lh_device_allreduce_fn fn, void* user) {
  if (!ctx) return LH_EINVAL;
  ctx->dev_reduce_fn = fn;
  ctx->dev_reduce_user = user;
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

static lh_status cloud_like(const lh_cloud* in, lh_cloud** out);
// PointCloudLocalization::MeasurementUpdate's device work in one call (PointCloudLocalization.cc:305-336, 398-421, 469-486, 694-750): see locus_hip.h
static lh_status measurement_update_impl(lh_gicp* g, const float guess[16], int want_information, double icp_max_covariance, lh_measurement* out,
                                         int32_t* corr, void* aligned_out, uint32_t stride, uint32_t off_xyz, uint32_t off_normal, lh_cloud** aligned_cloud) {
  if (!g || !out) return LH_EINVAL;
  if (aligned_cloud) *aligned_cloud = nullptr;
  lh_ctx* c = g->ctx;
  memset(out, 0, sizeof(*out));
  if (!g->src || !g->tgt) { out->result.status = LH_EINVAL; return LH_EINVAL; }
  if (want_information && !g->tgt->nrm) return LH_EINVAL;   // Ap reads the reference's normals
  lh_status st = lh_gicp_align(g, guess, &out->result, nullptr, nullptr, 0, 0);   // icp_->align (:309)
  if (st != LH_OK && st != LH_ETOO_FEW_CORR && st != LH_ESOLVER && st != LH_ENO_NN) return st;   // (the reference goes on with whatever transform align left)
  const lh_status align_status = st;
  const int n = g->src->n;
  if (n <= 0) return align_status;
  HIPCHK(hipSetDevice(c->device));
  const int nb = sum_blocks(n);
  DevGuard guard;
  float4 *d_xyz = nullptr, *d_nrm = nullptr;
  int32_t* d_idx = nullptr;
  float* d_d2 = nullptr;
  double* d_scr = nullptr;
  if (aligned_cloud) {   // the aligned query stays on the device: a cloud shaped like the source (every other field copied, like PCL's `output = input`)
    st = cloud_like(g->src, &guard.cloud);
    if (st) return st;
    d_xyz = guard.cloud->xyz;
    d_nrm = guard.cloud->nrm;
    if (g->src->intensity) HIPCHK(hipMemcpyAsync(guard.cloud->intensity, g->src->intensity, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  } else {
    HIPCHK(guard.alloc(&d_xyz, sizeof(float4) * (size_t)n));
    if (g->src->nrm) HIPCHK(guard.alloc(&d_nrm, sizeof(float4) * (size_t)n));
  }
  HIPCHK(guard.alloc(&d_idx, sizeof(int32_t) * (size_t)n));
  HIPCHK(guard.alloc(&d_d2, sizeof(float) * (size_t)n));
  HIPCHK(guard.alloc(&d_scr, sizeof(double) * ((size_t)nb * 21 + 4 + 21)));
  st = ctx_ensure_small(c, 21);
  if (st) return st;
  // pcl::transformPointCloudWithNormals(*query, *aligned_query, T) (:325): points and normals, on the device
  float T12[12];
  fill_T12(out->result.T, T12);
  { ProfScope p(c, "transform", 64.0 * n); launch_transform(g->src->xyz, g->src->nrm, n, T12, d_xyz, d_nrm, c->stream); }
  // the ungated 1-NN of every aligned point in the reference's tree (:327-336); the index is the one align() built
  if (!g->tgt->has_index) { st = cloud_build_index(g->tgt); if (st) return st; }
  { ProfScope p(c, "nn1", 24.0 * n); launch_nn1(d_xyz, n, nullptr, g->tgt->view(), d_idx, d_d2, c->stream); }
  // ComputePoint2PlaneICPCovariance / ComputeIcpObservability: normalizePCloud(query) + ComputeAp_ForPoint2PlaneICP (:469-486, 723-750)
  double* d_out21 = d_scr + (size_t)nb * 21 + 4;
  if (want_information) {
    ProfScope p(c, "p2plane_Ap", 40.0 * n);
    st = p2plane_information_device(g->src->xyz, n, g->tgt->nrm, d_idx, d_scr, d_out21, c->stream);
    if (st) return st;
    HIPCHK(hipMemcpyAsync(c->small_host, d_out21, sizeof(double) * 21, hipMemcpyDeviceToHost, c->stream));
  }
  // what the caller asked back: the correspondences, the aligned cloud -- and ONE wait for everything
  if (corr) HIPCHK(hipMemcpyAsync(corr, d_idx, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  std::vector<float> hx, hn;
  if (aligned_out) {
    hx.resize((size_t)n * 4);
    HIPCHK(hipMemcpyAsync(hx.data(), d_xyz, sizeof(float) * hx.size(), hipMemcpyDeviceToHost, c->stream));
    if (d_nrm && off_normal != 0xffffffffu) {
      hn.resize((size_t)n * 4);
      HIPCHK(hipMemcpyAsync(hn.data(), d_nrm, sizeof(float) * hn.size(), hipMemcpyDeviceToHost, c->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (aligned_out)
    for (int i = 0; i < n; i++) {
      memcpy((char*)aligned_out + (size_t)i * stride + off_xyz, &hx[4 * (size_t)i], 12);
      if (!hn.empty()) memcpy((char*)aligned_out + (size_t)i * stride + off_normal, &hn[4 * (size_t)i], 12);
    }
  if (want_information) {
    int t = 0;
    for (int r = 0; r < 6; r++)
      for (int cc = r; cc < 6; cc++) { out->Ap[r * 6 + cc] = c->small_host[t]; out->Ap[cc * 6 + r] = c->small_host[t]; t++; }
    out->have_information = 1;
    out->covariance_ok = lh_icp_covariance(out->Ap, icp_max_covariance, out->covariance, &out->condition_number) == LH_OK ? 1 : 0;
  }
  if (aligned_cloud) *aligned_cloud = guard.keep_cloud();
  return align_status;
}
lh_status lh_gicp_measurement_update(lh_gicp* g, const float guess[16], int want_information, double icp_max_covariance, lh_measurement* out,
                                     int32_t* corr, void* aligned_out, uint32_t stride, uint32_t off_xyz, uint32_t off_normal) {
  return measurement_update_impl(g, guess, want_information, icp_max_covariance, out, corr, aligned_out, stride, off_xyz, off_normal, nullptr);
}
lh_status lh_gicp_measurement_update_cloud(lh_gicp* g, const float guess[16], int want_information, double icp_max_covariance, lh_measurement* out,
                                           int32_t* corr, lh_cloud** aligned) {
  return measurement_update_impl(g, guess, want_information, icp_max_covariance, out, corr, nullptr, 0, 0, 0xffffffffu, aligned);
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
  if (q == target && k <= KNN_BLOCK_MAX_K) {   // a cloud against itself: the block search (one wave per 64 Morton-consecutive queries)
    lh_status st = knn_block_batch(c, &target, 1, k, KNN_MODE_RAW, 0.0, d_idx, d_d2);
    if (st) { (void)lhFree(d_idx); (void)lhFree(d_d2); return st; }
  } else {
    ProfScope p(c, "knn", (16.0 + 8.0 * k) * q->n);
    launch_knn(q->xyz, q->n, target->view(), k, d_idx, d_d2, c->stream);
  }
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

static lh_status align_batch_impl(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src, lh_cloud* const* tgt,
                                  const float* guesses, lh_gicp_result* out, lh_cloud** aligned, int max_in_flight, bool rebuild_index) {
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
  st = run_tasks(ctx, ptrs, in_flight, rebuild_index, &ctx->slot_ws);
  for (int i = 0; i < n_pairs; i++) out[i] = tasks[i].result;
  if (aligned) {  // the output clouds were written on the scheduler streams: complete before the caller touches them
    (void)hipStreamSynchronize(ctx->stream);
    ctx->sync_side_streams();
  }
  return st;
}

lh_status lh_gicp_align_batch_out(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src, lh_cloud* const* tgt,
                                  const float* guesses, lh_gicp_result* out, lh_cloud** aligned, int max_in_flight) {
  return align_batch_impl(ctx, p, n_pairs, src, tgt, guesses, out, aligned, max_in_flight, /*rebuild_index=*/true);
}

lh_status lh_gicp_align_batch(lh_ctx* ctx, const lh_gicp_params* p, int n_pairs, lh_cloud* const* src, lh_cloud* const* tgt,
                              const float* guesses, lh_gicp_result* out, int max_in_flight) {
  return lh_gicp_align_batch_out(ctx, p, n_pairs, src, tgt, guesses, out, nullptr, max_in_flight);
}

// The odometry stream (PointCloudOdometry.cc:237-322 over a queue of scans): pair i aligns scans[i + 1] (query) to scans[i] (reference,
// `copyPointCloud(*query_, *reference_)` of the previous update).  A scan's index is built ONCE -- by this call, or before it by the
// normal filter (lh_normals_knn_batch) -- and stays with the cloud; lh_gicp_align_batch rebuilds every target like initCompute does.
// The index is a function of the cloud alone, so the results are those of lh_gicp_align_batch, bit for bit.
lh_status lh_gicp_align_stream(lh_ctx* ctx, const lh_gicp_params* p, int n_scans, lh_cloud* const* scans, const float* guesses,
                               lh_gicp_result* out, int max_in_flight) {
  if (!scans || n_scans < 2) return LH_EINVAL;
  return align_batch_impl(ctx, p, n_scans - 1, scans + 1, scans, guesses, out, nullptr, max_in_flight, /*rebuild_index=*/false);
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
    st = upload_view(ctxs[d], &src[i], &S[i], /*sync=*/false);   // queued: the caller's arrays live until this call returns
    if (!st) owned.push_back(S[i]);
    if (!st) {
      if (chained) T[i] = S[i - 1];  // the previous scan is already on this device
      else { st = upload_view(ctxs[d], &tgt[i], &T[i], false); if (!st) owned.push_back(T[i]); }
    }
  }
  for (int d = 0; d < n_ctx; d++) {   // the uploads of every context, once
    (void)hipSetDevice(ctxs[d]->device);
    if (hipStreamSynchronize(ctxs[d]->stream) != hipSuccess && !st) st = LH_EDEVICE;
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
    a.job[0].pad = 1;   // the cold sweep: seeds in prev_nn, certificates / records not read
  }
  a.job[0].pad |= sweep_greedy_flag(sweep_index);
  CostArgs ca;
  ca.njobs = 1; ca.pad = 0; ca.job[0].slot = 0; ca.job[0].out_offset = 0;
  memcpy(ca.job[0].T, a.job[0].T, sizeof(a.job[0].T));
  launch_sweep_fused(c->descs_dev, a, sweep_is_split(&t, sweep_index) ? 1u : 0u, g->src->n, c->mom_partials_dev, c->mom_stride, nullptr, true, c->wmask_dev,
                     c->mask_stride, c->stream);
  ca.pad = (sweep_is_split(&t, sweep_index) || sweep_coop(true)) ? 0 : 1;   // whose rows the final sum adds: the fused sweep's or k_late's + k_walk's
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


}  // extern "C"
#pragma GCC visibility pop
