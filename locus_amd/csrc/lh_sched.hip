// lh_sched.hip -- the two schedulers of the registration path (see lh_runtime.hpp for Task): the host-driven loop (an alignment is a
// stackful coroutine that yields SWEEP / COST requests; cost_mode 0, single pairs, the source-sharded pair) and the device-driven loop
// (cost_mode 1 batches: k_sweep_fused / k_late + k_walk -> k_moments_final -> k_solve enqueued back to back, the host only looks).
#include "lh_runtime.hpp"

lh_status Workspace::ensure(lh_ctx* c, int n) {
  if (n <= cap) return LH_OK;
  (void)hipStreamSynchronize(c->stream);  // a slot may belong to either scheduler group: nothing may still use the old buffers
  c->sync_side_streams();
  (void)lhFree(corr); (void)lhFree(maha6); (void)lhFree(prev_nn); (void)lhFree(out_xyz); (void)lhFree(cert); (void)lhFree(rec);
  corr = nullptr; maha6 = nullptr; prev_nn = nullptr; out_xyz = nullptr; cert = nullptr; rec = nullptr; cap = 0;
  // a quarter of headroom (round 5): a stream's scans differ by a few per cent from one to the next (the adaptive voxel filter aims at a point
  // COUNT, Locus.cc:780-810), and with exact capacities every scan that was a little larger than all before it re-allocated the slot's six buffers
  // -- 2 ms of hipFree / hipMalloc on a 0.27-ms update (the production leg's p-max), 3.4 instead of 1.4 ms per 218 k-point frame of configs[4]
  int ncap = round_up(n + n / 4, 256);
  HIPCHK(hipMalloc(&corr, sizeof(float4) * (size_t)ncap));
  HIPCHK(hipMalloc(&maha6, sizeof(double) * 6 * (size_t)ncap));
  HIPCHK(hipMalloc(&prev_nn, sizeof(int32_t) * (size_t)ncap));
  HIPCHK(hipMalloc(&out_xyz, sizeof(float4) * (size_t)ncap));
  HIPCHK(hipMalloc(&cert, sizeof(float4) * (size_t)ncap));
  HIPCHK(hipMalloc(&rec, sizeof(float) * 6 * (size_t)ncap));   // two planes of packed triples (PairDesc::rec)
  if (!stats) { HIPCHK(hipMalloc(&stats, 16)); HIPCHK(hipMemset(stats, 0, 16)); }
  cap = ncap;
  n_pad = ncap;
  return LH_OK;
}
void Workspace::release() {
  (void)lhFree(corr); (void)lhFree(maha6); (void)lhFree(prev_nn); (void)lhFree(out_xyz); (void)lhFree(cert); (void)lhFree(rec); (void)lhFree(stats);
  corr = nullptr; maha6 = nullptr; prev_nn = nullptr; out_xyz = nullptr; cert = nullptr; rec = nullptr; stats = nullptr; cap = 0;
}

lh_status ctx_ensure_slots(lh_ctx* c, int n_slots, int max_n) {
  {  // already large enough for this call?  (checked on the exact size; an allocation below is made with a quarter of headroom, like the slots' own buffers)
    const size_t need_slot = std::max<size_t>((size_t)cost_blocks(max_n) * COST_NSUM, (size_t)FINAL_CHUNKS * MOM_ROW);
    if (n_slots <= c->n_slots && need_slot <= c->partials_per_slot && sweep_rows(max_n) * MOM_ROW <= c->mom_stride && ((max_n + 255) / 256) * 4 <= c->mask_stride) return LH_OK;
  }
  max_n = max_n + max_n / 4;
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
  // nothing is left half-described: if an allocation below fails, the context holds NO slot buffers and says so (n_slots = 0), so the next
  // call -- whatever it asks for -- allocates afresh instead of passing the capacity check above and launching on freed memory
  c->descs_dev = nullptr; c->mom_partials_dev = nullptr; c->wmask_dev = nullptr; c->states_dev = nullptr; c->chunks_dev = nullptr;
  c->descs_host = nullptr; c->partials_host = nullptr; c->states_host = nullptr; c->states_init = nullptr;
  c->n_slots = 0; c->partials_per_slot = 0; c->mom_stride = 0; c->mask_stride = 0;
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

// prepare device state of one pair in its slot: index, covariances, output cloud, descriptor
lh_status task_prepare(lh_ctx* c, Task* t, bool rebuild_index, bool upload_desc) {
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
  d.bfgs_quad_curv = P.bfgs_quad_curv; d.pad_b = 0;
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
bool sweep_is_split(const Task* t, int k) {
  return t->P.cost_mode == 1 && !t->P.recompute_source_cov && !t->P.recompute_target_cov && t->guess_is_identity && t->src->nrm &&
         t->tgt->nrm && t->ws->rec && k >= std::max(1, sweep_split_from());   // (never the cold sweep: k_late reads certificates and records)
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
          a.job[j].pad = 1;   // ... and the sweep that follows is the pair's cold one: it reads the seeds, not the slot's old certificates / records
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
        a.job[j].pad |= sweep_greedy_flag(t->sweeps_done);
        t->sweeps_done++;
      }
      {
        // (a second, nested scope by KIND of sweep -- every pair of the launch walks / certificate holders + walkers / both -- so that a profile
        // can put each kind against its own byte count: bench.py roofline.per_kernel)
        const uint32_t all = a.njobs >= 32 ? 0xffffffffu : ((1u << a.njobs) - 1u);
        ProfScope pk(c, split_mask == 0u ? "nn_sweep_allwalk" : (split_mask == all ? "nn_sweep_late_walk" : "nn_sweep_mixed"), 0.0, st);
        launch_sweep_fused(c->descs_dev, a, split_mask, max_n, c->mom_partials_dev, c->mom_stride, nullptr, normals_only, c->wmask_dev, c->mask_stride, st);
      }
      ca.pad = sweep_coop(normals_only) ? 0 : (int)(~split_mask & (a.njobs >= 32 ? 0xffffffffu : ((1u << a.njobs) - 1u)));   // the jobs whose rows the fused sweep left (one per 64 points)
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
          a.job[j].pad = 1;      // (the fused sweep's `cold` flag, lh_kernels.hip sweep_point)
          smax = std::max(smax, t->src->n);
          t->first_sweep = false;
        }
        if (sweep_is_split(t, t->enq_iters)) split_mask |= 1u << j;
        a.job[j].pad |= sweep_greedy_flag(t->enq_iters);
        t->enq_iters++;
      }
      if (seed.njobs > 0) {
        ProfScope p(c, "nn_seed", 0.0, st);
        launch_seed(c->descs_dev, seed, smax, st);
      }
      {
        ProfScope p(c, "nn_sweep", bytes, st);
        {
          const uint32_t all = a.njobs >= 32 ? 0xffffffffu : ((1u << a.njobs) - 1u);   // (nested scope by kind of sweep: see the host-driven loop)
          ProfScope pk(c, split_mask == 0u ? "nn_sweep_allwalk" : (split_mask == all ? "nn_sweep_late_walk" : "nn_sweep_mixed"), 0.0, st);
          launch_sweep_fused(c->descs_dev, a, split_mask, max_n, c->mom_partials_dev, c->mom_stride, c->states_dev, normals_only, c->wmask_dev, c->mask_stride, st);
        }
        ca.pad = sweep_coop(normals_only) ? 0 : (int)(~split_mask & (a.njobs >= 32 ? 0xffffffffu : ((1u << a.njobs) - 1u)));   // the jobs whose rows the fused sweep left (one per 64 points)
        launch_moments_final(c->descs_dev, ca, c->mom_partials_dev, c->mom_stride, c->chunks_dev, c->states_dev, c->wmask_dev, c->mask_stride, st);
      }
      if (c->dev_reduce_fn) {   // the source-sharded pair: the chunk sums are summed over the ranks where they lie, on this stream, before k_solve reads them
        ProfScope p(c, "moments_allreduce", 0.0, st);
        for (int j = 0; j < a.njobs; j++)
          if (c->dev_reduce_fn(c->chunks_dev + (size_t)sa.slot[j] * (FINAL_CHUNKS * MOM_ROW), FINAL_CHUNKS * MOM_ROW, (void*)st, c->dev_reduce_user) != 0) return LH_EDEVICE;
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
  static const int rounds_cfg = []() { const char* e = getenv("LH_DEVICE_ROUNDS"); int v = e ? atoi(e) : 6; return v < 1 ? 1 : v; }();   // (4 / 5 / 6 / 7 / 10 / 20 rounds: 9 800 / 9 940 / 10 010 / 9 990 / 9 990 / 9 890 pairs/s forced-20, natural convergence 19 650 / 19 700 / 19 670 / 19 670 / 19 330 / 18 530)
  // Groups: a pair's solve (one wave, tens of sequential cost evaluations) takes about as long as its sweep, so with more groups
  // in flight there is always somebody's sweep to run beside the other groups' solves.  Profiling keeps one group so that the
  // HIP-event times of the launches do not overlap.
  static const int groups_cfg = []() { const char* e = getenv("LH_DEVICE_GROUPS"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > lh_ctx::MAX_GROUPS ? lh_ctx::MAX_GROUPS : v); }();
  // one group per MAX_JOBS (32) pairs in flight -- a group's launch covers all its pairs -- up to 32 groups = streams: with 256 in
  // flight, eight groups of 32 ran 14 % more pairs/s than four of 64 (each chain is half latency: solve, start-up, lone searches)
  // With few pairs in flight the chains are the bound, not the GPU: three groups of 21-22 run 64 pairs 5 % faster than two of 32 (7 320 ->
  // 7 690 pairs/s), six of 21 run 128 pairs 2 % faster than four of 32; from 256 in flight on, full launches of 32 win (512: 16 groups
  // 8 917, 21 groups 8 509, 32 groups 7 554).
  int G = groups_cfg ? groups_cfg
                     : (in_flight > 128 ? std::min(lh_ctx::MAX_GROUPS, in_flight / MAX_JOBS) : (in_flight >= 64 ? (in_flight + 10) / 21 : (in_flight >= 16 ? 2 : 1)));
  if (c->prof) G = 1;
  G = std::max(1, std::min(G, in_flight));
  hipStream_t* extra[lh_ctx::MAX_GROUPS - 1] = {&c->stream2, &c->stream3, &c->stream4};
  for (int k = 0; k < lh_ctx::MAX_GROUPS - 4; k++) extra[3 + k] = &c->stream_more[k];
  for (int gi = 1; gi < G; gi++)
    if (!*extra[gi - 1]) HIPCHK(hipStreamCreateWithFlags(extra[gi - 1], hipStreamNonBlocking));
  { lh_status fst = fork_side_streams(c, extra, G - 1); if (fst) return fst; }
  runtime_check_streams(c, G);
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
          // each group builds in its own scratch set: the builds of the sixteen groups overlap instead of queueing on one scratch
          st = build_indices(c, to_build.data(), (int)to_build.size(), g.stream, gi);
          if (st) return fail(st);
          for (lh_cloud* tg : to_build) tg->built_epoch = epoch;
          hp_build += hp_now() - hb;
        }
        // a target another group built: that build was enqueued earlier in this call, on that group's stream and scratch set
        if (built_elsewhere && (st = wait_index_builds(c, g.stream))) return fail(st);
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
  { lh_status fst = fork_side_streams(c, extra, G - 1); if (fst) return fst; }
  runtime_check_streams(c, G);
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
          st = build_indices(c, to_build.data(), (int)to_build.size(), g.stream, gi);
          if (st) return fail(st);
          for (lh_cloud* tg : to_build) tg->built_epoch = epoch;
        }
        if (built_elsewhere && (st = wait_index_builds(c, g.stream))) return fail(st);
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
lh_status run_tasks(lh_ctx* c, std::vector<Task*>& tasks, int in_flight, bool rebuild_index, std::vector<Workspace>* slot_ws) {
  // a host SUM hook needs the host between the reduction and the solve; a device SUM hook keeps the loop on the device (and asks for it:
  // without the host hook a host-driven loop would solve on this rank's shard alone)
  bool device_loop = !c->reduce_fn || c->dev_reduce_fn;
  const bool forced_by_hook = c->dev_reduce_fn != nullptr;
  bool forced = false;
  for (Task* t : tasks) {
    if (t->P.cost_mode != 1 || t->P.solver == 1 || t->count_stats || t->P.max_iterations < 1) device_loop = false;
    if (t->P.solver == 2) forced = true;
  }
  if (device_loop && !forced && !forced_by_hook && std::min<size_t>(in_flight, tasks.size()) < (size_t)DEVICE_LOOP_MIN_IN_FLIGHT) device_loop = false;
  if (!device_loop && forced_by_hook && !c->reduce_fn) return LH_EINVAL;   // only a device hook, and a mode the device loop does not run (cost_mode 0, solver 1, debug counters): the shards could not be summed
  return device_loop ? run_tasks_device(c, tasks, in_flight, rebuild_index, slot_ws) : run_tasks_host(c, tasks, in_flight, rebuild_index, slot_ws);
}

