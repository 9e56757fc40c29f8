// lh_index.hip -- K2: the batched NN-index build (replaces tree_->setInputCloud of pcl::Registration::initCompute) and the k-NN
// covariances of a cloud (gicp.hpp:85-154), host side (see lh_runtime.hpp; kernels in lh_kernels.hip / lh_radix.hip).
#include <atomic>

#include "lh_runtime.hpp"

std::atomic<bool> g_small_index{[]() { const char* e = getenv("LH_SMALL_INDEX"); return !e || atoi(e) != 0; }()};

// K2: Morton sort + cell-aligned radix tree with 4-ary nodes (replaces tree_->setInputCloud of pcl::Registration::initCompute).
// All clouds of a batch are built by the same launches, one radix sort and one scan (see lh_kernels.hpp "K2 batched").
lh_status build_indices(lh_ctx* x, lh_cloud* const* clouds, int n_clouds, hipStream_t s_in, int set) {
  if (n_clouds <= 0) return LH_OK;
  hipStream_t s = s_in ? s_in : x->stream;
  lh_ctx::IndexScratch& X = x->idx_sets[((set % lh_ctx::IDX_SETS) + lh_ctx::IDX_SETS) % lh_ctx::IDX_SETS];   // the build scratch this call owns (lh_runtime.hpp)
  for (int o = 0; o < n_clouds; o += MAX_INDEX_BATCH) {
    int nb = std::min(MAX_INDEX_BATCH, n_clouds - o);
    long total = 0;
    int max_n = 0, tile0 = 0;
    if (!X.descs_dev) {
      HIPCHK(hipMalloc(&X.descs_dev, sizeof(IndexDesc) * MAX_INDEX_BATCH));
      HIPCHK(hipHostMalloc(&X.descs_host, sizeof(IndexDesc) * MAX_INDEX_BATCH * lh_ctx::IndexScratch::STAGES, hipHostMallocDefault));
      HIPCHK(hipMalloc(&X.bbox, sizeof(uint32_t) * 8 * MAX_INDEX_BATCH));
      launch_index_bbox_init(X.bbox, s);   // (every build then leaves the slots reset for the next one)
      for (int k = 0; k < lh_ctx::IndexScratch::STAGES; k++) HIPCHK(hipEventCreateWithFlags(&X.copy_done[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&X.build_done, hipEventDisableTiming));
    }
    // The descriptors are staged in a ring: the upload of a build is queued behind the previous build of the same set on the GPU, so
    // waiting for the PREVIOUS upload before refilling one staging buffer would tie the calling thread to the GPU's index builds (with
    // one shared scratch, round 3: 1.6 ms per group of 32, 25 of a 60-ms step).  Only the upload STAGES builds ago is waited for.
    const int stage = X.stage;
    X.stage = (X.stage + 1) % lh_ctx::IndexScratch::STAGES;
    IndexDesc* const stage_host = X.descs_host + (size_t)stage * MAX_INDEX_BATCH;
    HIPCHK(hipEventSynchronize(X.copy_done[stage]));   // (an event that was never recorded is complete)
    // the chunk's clouds: the large ones first (their sorted positions must be one contiguous run for the batched launches), then the small
    // ones, which are built by one launch of one workgroup each (lh_index_small.hip) on their own slices of the same scratch
    const bool small_path = g_small_index.load();   // (lh_debug_small_index / LH_SMALL_INDEX=0: every cloud through the general build -- A/B, tests)
    std::vector<lh_cloud*> ordered;
    for (int pass = 0; pass < 2; pass++)
      for (int k = 0; k < nb; k++) {
        lh_cloud* c = clouds[o + k];
        if (!c || c->n <= 0 || c->ctx != x) return LH_EINVAL;
        const bool small = small_path && c->n <= SMALL_INDEX_MAX_N;
        if (small == (pass == 1)) ordered.push_back(c);
      }
    int n_big = 0;
    long total_big = 0;
    for (int k = 0; k < nb; k++) {
      lh_cloud* c = ordered[k];
      if (c->n > (1 << MAX_POINT_BITS)) return LH_EINVAL;  // leaf references keep 27 bits of sorted position; the traversal stack is sized for it (lh_device.hpp)
      if (c->n > c->index_cap) {
        (void)hipStreamSynchronize(x->stream);
        x->sync_side_streams();
        (void)lhFree(c->sorted); (void)lhFree(c->node_buf);
        c->sorted = nullptr; c->node_buf = nullptr; c->index_cap = 0;
        HIPCHK(lhMalloc(&c->sorted, sizeof(float4) * ((size_t)c->n + LEAF_CAP)));
        HIPCHK(lhMalloc(&c->node_buf, sizeof(NodeX) * ((size_t)c->n + 1 + GRID_NODEX)));  // header + start grid + worst case: every point its own leaf
        c->index_cap = c->n;
      }
      IndexDesc& d = stage_host[k];
      d.xyz = c->xyz; d.sorted = c->sorted; d.nodes = c->nodes(); d.hdr = c->hdr(); d.pos = nullptr;
      d.n = c->n; d.offset = (int)total;
      d.tile0 = tile0; d.pad = 0;
      const bool small = small_path && c->n <= SMALL_INDEX_MAX_N;
      if (!small) {
        tile0 += segsort_tiles(c->n);
        n_big++;
        total_big += c->n;
        max_n = std::max(max_n, c->n);
      }
      total += c->n;
    }
    if (total > 0x7ffffff0L) return LH_EINVAL;
    if ((int)total > X.cap) {
      (void)hipStreamSynchronize(x->stream);
      x->sync_side_streams();
      (void)lhFree(X.k64a); (void)lhFree(X.k64b); (void)lhFree(X.v32a); (void)lhFree(X.v32b); (void)lhFree(X.sort64_temp);
      (void)lhFree(X.tree_tmp); (void)lhFree(X.k32a); (void)lhFree(X.k32b); (void)lhFree(X.rs_hist);
      // (a failed allocation below must not leave the old capacity standing over freed buffers)
      X.k64a = X.k64b = nullptr; X.v32a = X.v32b = nullptr; X.sort64_temp = nullptr; X.tree_tmp = nullptr;
      X.k32a = X.k32b = nullptr; X.rs_hist = nullptr; X.cap = 0;
      int cap = round_up((int)std::min<long>(total + total / 4, 0x7fffff00L), 1024);
      HIPCHK(hipMalloc(&X.k64a, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&X.k64b, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&X.v32a, sizeof(uint32_t) * (size_t)cap));
      HIPCHK(hipMalloc(&X.v32b, sizeof(uint32_t) * (size_t)cap));
      HIPCHK(hipMalloc(&X.k32a, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&X.k32b, sizeof(uint64_t) * (size_t)cap));
      HIPCHK(hipMalloc(&X.rs_hist, sizeof(uint32_t) * segsort_hist_elems(cap, MAX_INDEX_BATCH)));
      X.sort64_temp_bytes = sort64_temp_bytes(cap);
      HIPCHK(hipMalloc(&X.sort64_temp, X.sort64_temp_bytes ? X.sort64_temp_bytes : 16));
      HIPCHK(hipMalloc(&X.tree_tmp, TREE_SCRATCH_BYTES_PER_POINT * ((size_t)cap + 16) + 4096));
      HIPCHK(hipMemsetAsync(X.tree_tmp, 0, TREE_SCRATCH_BYTES_PER_POINT * ((size_t)cap + 16) + 4096, s));   // (the per-tile leaf counts must start at zero; every build leaves them so)
      X.cap = cap;
    }
    TreeScratch ts;
    {
      size_t cap = (size_t)X.cap + 16;
      char* p = X.tree_tmp;
      ts.lkey = reinterpret_cast<uint64_t*>(p); p += 8 * cap;       // 16-byte aligned arrays first (cap is a multiple of 16)
      ts.lbox = reinterpret_cast<float4*>(p); p += 32 * cap;
      ts.a1box = reinterpret_cast<float4*>(p); p += 32 * (cap / 32 + 16);
      ts.a2box = reinterpret_cast<float4*>(p); p += 32 * (cap / 1024 + 16);
      ts.ibox = reinterpret_cast<float4*>(p); p += 32 * cap;
      ts.ichild = reinterpret_cast<int32_t*>(p); p += 8 * cap;
      ts.irange = reinterpret_cast<int32_t*>(p); p += 8 * cap;
      ts.iparent = reinterpret_cast<int32_t*>(p); p += 4 * cap;
      ts.icom = reinterpret_cast<int32_t*>(p); p += 4 * cap;
      ts.flag = reinterpret_cast<uint32_t*>(p); p += 4 * cap;
      ts.lid = reinterpret_cast<uint32_t*>(p); p += 4 * cap;
      ts.lstart = reinterpret_cast<uint32_t*>(p); p += 4 * cap;
      ts.tsum = reinterpret_cast<uint32_t*>(p); p += 4 * (cap / 4096 + 16);
      ts.toff = reinterpret_cast<uint32_t*>(p); p += 4 * (cap / 4096 + 16);
      ts.keys = X.k64b;
      ts.total = (int)total_big;   // the batched launches cover the large clouds' positions [0, total_big)
    }
    HIPCHK(hipStreamWaitEvent(s, X.build_done, 0));  // the previous build of this set may still be running on another stream
    HIPCHK(hipMemcpyAsync(X.descs_dev, stage_host, sizeof(IndexDesc) * nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(X.copy_done[stage], s));
    int id_bits = 0;
    while ((1 << id_bits) < std::max(n_big, 1)) id_bits++;
    if (n_big > 0) {
    // LH_SORT=generic: the one-segment 64-bit sort over the whole concatenated array instead of the segmented one (A/B);
    // LH_SORT=check: both, compared element by element (tests: two independent code paths must give the same stable order)
    static const int sort_cfg = []() { const char* e = getenv("LH_SORT"); return !e ? 0 : (strcmp(e, "generic") == 0 ? 1 : (strcmp(e, "check") == 0 ? 2 : 0)); }();
    { ProfScope p(x, "index_bbox_keys", 16.0 * total_big * 2, s);
      launch_index_keys(X.descs_dev, n_big, max_n, X.bbox, X.k32a, sort_cfg ? X.k64a : nullptr, sort_cfg ? X.v32a : nullptr, s); }
    {
      if (sort_cfg == 1) {
        ProfScope p(x, "index_radix_sort", 12.0 * total_big * 2 * 4, s);
        sort_pairs_u64(X.sort64_temp, X.sort64_temp_bytes, X.k64a, X.k64b, X.v32a, X.v32b, (int)total_big, 32 + id_bits, s);
      } else {
        std::vector<uint64_t> kref;
        std::vector<uint32_t> vref;
        if (sort_cfg == 2) {  // reference first
          sort_pairs_u64(X.sort64_temp, X.sort64_temp_bytes, X.k64a, X.k64b, X.v32a, X.v32b, (int)total_big, 32 + id_bits, s);
          kref.resize(total_big); vref.resize(total_big);
          HIPCHK(hipMemcpyAsync(kref.data(), X.k64b, sizeof(uint64_t) * total_big, hipMemcpyDeviceToHost, s));
          HIPCHK(hipMemcpyAsync(vref.data(), X.v32b, sizeof(uint32_t) * total_big, hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
        }
        { ProfScope p(x, "index_radix_sort", 8.0 * total_big * 3 * 2, s);
          segsort_pairs(X.descs_dev, n_big, max_n, X.k32a, X.k32b, X.k64b, X.v32b, X.rs_hist, s); }
        if (sort_cfg == 2) {
          std::vector<uint64_t> kk(total_big);
          std::vector<uint32_t> vv(total_big);
          HIPCHK(hipMemcpyAsync(kk.data(), X.k64b, sizeof(uint64_t) * total_big, hipMemcpyDeviceToHost, s));
          HIPCHK(hipMemcpyAsync(vv.data(), X.v32b, sizeof(uint32_t) * total_big, hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
          long bad = 0;
          for (long i = 0; i < total_big; i++)
            if (kk[i] != kref[i] || vv[i] != vref[i]) bad++;
          if (bad) {
            fprintf(stderr, "[locus_hip] LH_SORT=check: %ld of %ld sorted elements differ between the segmented and the one-segment sort\n", bad, total_big);
            return LH_EDEVICE;
          }
        }
      }
    }
    { ProfScope p(x, "index_leaves", 8.0 * total_big * 3 + 48.0 * total_big, s); launch_index_leaves(X.descs_dev, n_big, ts, X.v32b, X.bbox, s); }
    { ProfScope p(x, "index_box_tables", 24.0 * total_big, s); launch_index_trees(X.descs_dev, n_big, max_n, ts, s, 0); }
    { ProfScope p(x, "index_radix_tree", 8.0 * total_big, s); launch_index_trees(X.descs_dev, n_big, max_n, ts, s, 1); }
    { ProfScope p(x, "index_nodes", 32.0 * total_big, s); launch_index_trees(X.descs_dev, n_big, max_n, ts, s, 2); }
    }
    if (nb > n_big) {   // the small clouds: the whole build in one launch, one workgroup per cloud
      ProfScope p(x, "index_small", 120.0 * (total - total_big), s);
      launch_index_small(X.descs_dev + n_big, nb - n_big, ts, s);
    }
    HIPCHK(hipEventRecord(X.build_done, s));
    HIPCHK(hipGetLastError());
    for (int k = 0; k < nb; k++) clouds[o + k]->has_index = true;
  }
  return LH_OK;
}

// K3 over a batch of clouds (normal_computation.cc:26-59 for every scan of a stream; gicp.hpp:85-154 in the recompute mode): the
// clouds without an index are built together (one batched build), then ONE launch of the block search serves up to MAX_INDEX_BATCH
// clouds -- a scan's tree is built once and stays with the cloud for the alignment that follows.
lh_status knn_block_batch(lh_ctx* x, lh_cloud* const* clouds, int n_clouds, int k, int mode, double eps, int32_t* idx_dev, float* d2_dev) {
  if (!x || !clouds || n_clouds <= 0 || k < 1 || k > 64) return LH_EINVAL;
  if (mode == KNN_MODE_RAW && n_clouds != 1) return LH_EINVAL;
  std::vector<lh_cloud*> need;
  for (int i = 0; i < n_clouds; i++) {
    lh_cloud* c = clouds[i];
    if (!c || c->ctx != x || c->n <= 0) return LH_EINVAL;
    if (mode == KNN_MODE_COV && k > c->n) return LH_EINVAL;   // gicp.hpp:72-79
    if (!c->has_index) need.push_back(c);
  }
  if (!need.empty()) {
    lh_status st = build_indices(x, need.data(), (int)need.size());
    if (st) return st;
  }
  for (int i = 0; i < n_clouds; i++) {
    lh_cloud* c = clouds[i];
    if (mode == KNN_MODE_NORMALS && !c->nrm) HIPCHK(lhMalloc(&c->nrm, sizeof(float4) * (size_t)c->n_pad));
    if (mode == KNN_MODE_COV && !c->cov6) HIPCHK(lhMalloc(&c->cov6, sizeof(double) * 6 * (size_t)c->n_pad));
  }
  const double model_bytes = mode == KNN_MODE_COV ? 16.0 + 16.0 * k + 48.0 : (mode == KNN_MODE_NORMALS ? 16.0 + 16.0 * k + 16.0 : 16.0 + 8.0 * k);
  const char* tag = mode == KNN_MODE_COV ? "knn_cov" : (mode == KNN_MODE_NORMALS ? "knn_normals" : "knn");
  if (k > KNN_BLOCK_MAX_K) {   // beyond the register lists of the block search: one query per lane, one cloud per launch
    for (int i = 0; i < n_clouds; i++) {
      lh_cloud* c = clouds[i];
      ProfScope p(x, tag, model_bytes * c->n);
      if (mode == KNN_MODE_NORMALS) launch_knn_normals(c->xyz, c->n, c->view(), k, c->nrm, x->stream);
      else if (mode == KNN_MODE_COV) launch_knn_cov(c->xyz, c->n, c->n_pad, c->view(), k, eps, c->cov6, x->stream);
      else launch_knn(c->xyz, c->n, c->view(), k, idx_dev, d2_dev, x->stream);
    }
    HIPCHK(hipGetLastError());
    return LH_OK;
  }
  if (!x->knn_descs_dev) {
    HIPCHK(lhMalloc(&x->knn_descs_dev, sizeof(KnnCloudDesc) * MAX_INDEX_BATCH));
    HIPCHK(lhMalloc(&x->knn_redo_cnt, 256));
  }
  for (int o = 0; o < n_clouds; o += MAX_INDEX_BATCH) {
    const int nb = std::min(MAX_INDEX_BATCH, n_clouds - o);
    KnnCloudDesc hd[MAX_INDEX_BATCH];
    long pts = 0;
    int max_n = 0;
    for (int i = 0; i < nb; i++) {
      lh_cloud* c = clouds[o + i];
      KnnCloudDesc& d = hd[i];
      d.pts = c->sorted; d.nodes = c->nodes(); d.hdr = c->hdr(); d.xyz = c->xyz;
      d.nrm = c->nrm; d.cov6 = c->cov6; d.idx = idx_dev; d.d2 = d2_dev;
      d.n = c->n; d.n_pad = c->n_pad;
      pts += c->n;
      max_n = std::max(max_n, c->n);
    }
    if (pts > x->knn_redo_cap) {
      (void)hipStreamSynchronize(x->stream);
      (void)lhFree(x->knn_redo); (void)lhFree(x->knn_soa);
      x->knn_redo = nullptr; x->knn_soa = nullptr; x->knn_redo_cap = 0;
      const long cap = pts + pts / 4 + 1024;
      HIPCHK(lhMalloc(&x->knn_redo, sizeof(uint2) * (size_t)cap));
      // a cloud's arrays take round_up16(n + LEAF_CAP) <= n + LEAF_CAP + 15 floats each: room for the padding of a full batch whatever its sizes
      HIPCHK(lhMalloc(&x->knn_soa, sizeof(float) * 3 * ((size_t)cap + (size_t)(LEAF_CAP + 16) * MAX_INDEX_BATCH)));
      x->knn_redo_cap = cap;
    }
    {
      size_t off = 0;   // every array starts on a 64-byte boundary
      for (int i = 0; i < nb; i++) {
        const size_t len = ((size_t)hd[i].n + LEAF_CAP + 15) & ~(size_t)15;
        hd[i].sx = x->knn_soa + off; hd[i].sy = hd[i].sx + len; hd[i].sz = hd[i].sy + len;
        off += 3 * len;
      }
    }
    // (pageable source: the copy is staged before the call returns, and it is queued behind the previous launch that reads the table)
    HIPCHK(hipMemcpyAsync(x->knn_descs_dev, hd, sizeof(KnnCloudDesc) * nb, hipMemcpyHostToDevice, x->stream));
    {
      ProfScope p(x, tag, model_bytes * pts);
      launch_knn_block(x->knn_descs_dev, nb, max_n, k, mode, eps, x->knn_redo_cnt, x->knn_redo, x->stream);
    }
    HIPCHK(hipGetLastError());
  }
  if (mode == KNN_MODE_COV)
    for (int i = 0; i < n_clouds; i++) { clouds[i]->cov_k = k; clouds[i]->cov_eps = eps; }
  return LH_OK;
}

lh_status cloud_ensure_cov(lh_cloud* c, int k, double eps) {
  if (c->cov6 && c->cov_k == k && c->cov_eps == eps) return LH_OK;
  if (k > c->n || k > 64 || k < 1) return LH_EINVAL;  // gicp.hpp:72-79
  return knn_block_batch(c->ctx, &c, 1, k, KNN_MODE_COV, eps);
}

