// Host model (no GPU): dependent steps of the sweeps' exact 1-NN walks when a query starts at the subtree of ITS OWN grid cell
// (a direct table: level-L cell of the Morton key grid -> the radix-tree node / leaf that holds exactly the points of that cell)
// and then visits the neighbour cells its best-distance ball reaches, against the walk from the root.  Per sweep: mean steps per
// query and the mean over waves (64 consecutive queries) of the per-wave maximum -- what a wave pays.
//   hipcc -O2 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -x hip tools/model/grid_start_model.cpp -o /tmp/gsm && /tmp/gsm /tmp/wm
//   (/tmp/wm: tgt.f32, src.f32 = a scan pair as float triples, poses.f32 = 3x4 row-major transforms, one per modelled sweep)
#define TRAVERSAL_CHECK_NO_MAIN
#include "../../tests/host_emu/traversal_check.cpp"
#include <cstring>
#include <string>

static std::vector<float> read_f32(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { printf("cannot open %s\n", path.c_str()); exit(1); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<float> v(sz / 4);
  if (fread(v.data(), 4, v.size(), f) != v.size()) exit(1);
  fclose(f);
  return v;
}


struct Walk { int nodes = 0, leaves = 0; };
// the step-wise walk of the sweeps (node_visit / scan_leaf / pop) from an arbitrary start reference
static void walk_from(const TreeView& tv, int32_t ref, float qx, float qy, float qz, Nn1CertCollector& c, Walk& w) {
  TreeHeader h = *tv.hdr;
  GridQuery gq = grid_query(h, qx, qy, qz);
  std::vector<uint64_t> st(LDS_STACK);
  WalkStack<LDS_STACK> ws(st.data(), 1);
  for (;;) {
    while (ref >= 0 && ref != NO_CHILD) { ref = node_visit(tv.nodes[ref], gq, h.scl2, c, ws); w.nodes++; }
    if (ref == NO_CHILD) break;
    scan_leaf(tv, ref, qx, qy, qz, c); w.leaves++;
    ref = ws.pop(c);
  }
}


// One wave of the fused sweep in lockstep ("while-while": every lane that holds a node takes node steps until none does, then every
// lane that holds a leaf scans it and pops): the number of node iterations and leaf iterations the WAVE executes, and the lane-steps
// that did useful work in them.
struct LaneState { Nn1CertCollector col; std::vector<uint64_t> mem; WalkStack<LDS_STACK> ws; GridQuery gq; float q[3]; int32_t ref;
                   LaneState() : col{inf_f(), 0x7fffffff, inf_f()}, mem(LDS_STACK), ws(mem.data(), 1), gq{0, 0, 0, 0.f}, ref(NO_CHILD) {} };
static int g_policy = 0;
static long g_hist[65];      // iterations by number of busy lanes
static long g_cut_iters[5], g_cut_lanes[5];   // iterations executed / lanes handed off if a wave stops once <= {0,2,4,8,16} lanes are busy
static int g_group = 64;   // lanes that share one instruction stream in the model (64 = a wave with one query per lane; 16 = a wave of sixteen
                           // 4-lane query groups would execute max-over-16 iterations: what a cooperative search would pay)
static void wave_lockstep_all(const TreeView& tv, const TreeHeader& h, std::vector<LaneState>& L, long& node_iters, long& leaf_iters, long& busy, bool hist);
static void wave_lockstep(const TreeView& tv, const TreeHeader& h, std::vector<LaneState>& L, long& node_iters, long& leaf_iters, long& busy, bool hist = false) {
  if (g_group >= (int)L.size()) { wave_lockstep_all(tv, h, L, node_iters, leaf_iters, busy, hist); return; }
  for (size_t b = 0; b < L.size(); b += g_group) {
    std::vector<LaneState> sub(g_group);
    for (int k = 0; k < g_group; k++) {   // (a copy whose stack lives in its own memory)
      const LaneState& o = L[b + k];
      LaneState& n = sub[k];
      n.col = o.col; n.gq = o.gq; n.ref = o.ref;
      n.q[0] = o.q[0]; n.q[1] = o.q[1]; n.q[2] = o.q[2];
      n.mem = o.mem;
      n.ws = WalkStack<LDS_STACK>(n.mem.data(), 1);
      n.ws.sp = o.ws.sp;
      memcpy(n.ws.spill, o.ws.spill, sizeof(n.ws.spill));
    }
    wave_lockstep_all(tv, h, sub, node_iters, leaf_iters, busy, hist);
    for (int k = 0; k < g_group; k++) L[b + k].col = sub[k].col;
  }
}
static void wave_lockstep_all(const TreeView& tv, const TreeHeader& h, std::vector<LaneState>& L, long& node_iters, long& leaf_iters, long& busy, bool hist) {
  const int cuts[5] = {0, 2, 4, 8, 16};
  bool cut_done[5] = {false, false, false, false, false};
  auto note = [&](int act) {
    if (!hist) return;
    g_hist[act]++;
    for (int k = 0; k < 5; k++) {
      if (!cut_done[k] && act <= cuts[k]) { cut_done[k] = true; g_cut_lanes[k] += act; }
      if (!cut_done[k]) g_cut_iters[k]++;
    }
  };
  if (g_policy == 0) {
  for (;;) {
    bool any = false;
    for (auto& l : L) any |= l.ref != NO_CHILD;
    if (!any) break;
    for (;;) {   // node phase
      int act = 0;
      for (auto& l : L) if (l.ref >= 0 && l.ref != NO_CHILD) act++;
      if (!act) break;
      node_iters++; busy += act; note(act);
      for (auto& l : L) if (l.ref >= 0 && l.ref != NO_CHILD) l.ref = node_visit(tv.nodes[l.ref], l.gq, h.scl2, l.col, l.ws);
    }
    int act = 0;
    for (auto& l : L) if (l.ref < 0) act++;
    if (act) {
      leaf_iters++; busy += act; note(act);
      for (auto& l : L) if (l.ref < 0) { scan_leaf(tv, l.ref, l.q[0], l.q[1], l.q[2], l.col); l.ref = l.ws.pop(l.col); }
    }
  }
  } else {
    bool last_leaf = true;
    for (;;) {
      int nn = 0, nl = 0;
      for (auto& l : L) { if (l.ref >= 0 && l.ref != NO_CHILD) nn++; else if (l.ref < 0) nl++; }
      if (!nn && !nl) break;
      bool do_node;
      if (g_policy == 1) do_node = nn * 1.0 >= nl * 1.0;           // majority vote
      else if (g_policy >= 10) do_node = nl == 0 || (nn > 0 && nn * 4 >= nl * (g_policy - 10));   // weighted vote: policy 10 + 4 w
      else if (g_policy == 3) do_node = nl == 0 || (nn > 0 && nn >= 3 * nl);   // leaves first unless nodes dominate 3:1
      else do_node = nn > 0 && (last_leaf || nl == 0);           // if-if: alternate
      if (do_node) {
        node_iters++; busy += nn; note(nn);
        for (auto& l : L) if (l.ref >= 0 && l.ref != NO_CHILD) l.ref = node_visit(tv.nodes[l.ref], l.gq, h.scl2, l.col, l.ws);
        last_leaf = false;
      } else {
        leaf_iters++; busy += nl; note(nl);
        for (auto& l : L) if (l.ref < 0) { scan_leaf(tv, l.ref, l.q[0], l.q[1], l.q[2], l.col); l.ref = l.ws.pop(l.col); }
        last_leaf = true;
      }
    }
  }
}

int main(int argc, char** argv) {
  std::string dir = argc > 1 ? argv[1] : "/tmp/wm";
  g_policy = argc > 2 ? atoi(argv[2]) : 0;
  g_group = argc > 3 ? atoi(argv[3]) : 64;
  auto tg = read_f32(dir + "/tgt.f32"), sr = read_f32(dir + "/src.f32"), po = read_f32(dir + "/poses.f32");
  int m = (int)tg.size() / 3, n = (int)sr.size() / 3, np = (int)po.size() / 12;
  std::vector<float4> tp(m);
  for (int i = 0; i < m; i++) tp[i] = make_float4(tg[3 * i], tg[3 * i + 1], tg[3 * i + 2], 1.f);
  HostTree t = build(tp);   // (fills the start grid with the product's grid_fill_child / grid_fill_root)
  TreeView tv = t.view();
  long set5 = 0, set4 = 0, set3 = 0, setf = 0;
  for (int k = 0; k < GRID_ENTRIES; k++)
    if (t.grid()[k] != GRID_EMPTY) (k < GRID_OFF5 ? setf : (k < GRID_OFF4 ? set5 : (k < GRID_OFF3 ? set4 : set3)))++;
  printf("target %d points, %d leaves; start grid (finest level %d): %ld / %ld / %ld cells set at levels 5 / 4 / 3, %ld at the finer levels\n", m, t.n_leaves, GRID_FINEST, set5, set4, set3, setf);
  const TreeHeader h = t.hdr();
  std::vector<int> prev(n, -1);
  for (int s = 0; s < np; s++) {
    const float* T = &po[12 * s];
    double sum_b = 0, sum_g = 0, wm_b = 0, wm_g = 0, sum_push = 0;
    long waves = 0, mism = 0, lbbad = 0, use_root = 0, bnodes = 0, bleaves = 0, gnodes = 0, gleaves = 0;
    int wb = 0, wg = 0;
    std::vector<LaneState> LA(64), LB(64), LC(64);
    int desc_nodes[64] = {0};
    long c_desc = 0, c_node = 0, c_leaf = 0, c_busy = 0, c_mism = 0;
    long a_node = 0, a_leaf = 0, a_busy = 0, g_node = 0, g_leaf = 0, g_busy = 0;
    for (int i = 0; i < n; i++) {
      float qx, qy, qz;
      xform_pt(T, sr[3 * i], sr[3 * i + 1], sr[3 * i + 2], qx, qy, qz);
      int cand = prev[i];
      if (s == 0) {   // seed pass: descent of every 4th point (from its own grid cell, like k_seed), shared by its group
        int g0 = i & ~3;
        if (i == g0) {
          Nn1Collector cd{inf_f(), 0x7fffffff};
          tree_descend<Nn1Collector, true>(tv, qx, qy, qz, cd);
          prev[g0] = cd.bi;
        }
        cand = prev[g0];
      }
      float cd2 = d2f(qx, qy, qz, tp[cand].x, tp[cand].y, tp[cand].z);
      {   // the same two walks as lanes of a lockstep wave
        const int ln = i & 63;
        LaneState &la = LA[ln], &lb = LB[ln];
        la.col = lb.col = Nn1CertCollector{cd2, cand, inf_f()};
        la.gq = lb.gq = grid_query(h, qx, qy, qz);
        la.q[0] = lb.q[0] = qx; la.q[1] = lb.q[1] = qy; la.q[2] = lb.q[2] = qz;
        la.ws.sp = lb.ws.sp = 0;
        la.ref = h.root;
        lb.ref = h.root;
        const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), qx, qy, qz, lb.col, [&](uint32_t key, int32_t r) { lb.ws.push(key, r); });
        if (g != GRID_USE_ROOT) lb.ref = (g == GRID_EMPTY) ? lb.ws.pop(lb.col) : g;
        if (s == 0) {   // (c) no seed pass: the lane's OWN descent from its level-5 cell (root if that is empty), then the same grid walk
          LaneState& lc = LC[ln];
          Nn1CountCollector cc{inf_f(), 0x7fffffff, 0, 0};
          for (int k = 0; k < MAX_DEPTH; k++) cc.per_level[k] = 0;
          {   // tree_descend<.., true> with counting
            int32_t r = h.root;
            const int cx = (int)fminf(fmaxf((qx - h.org[0]) * h.key_sc, 0.0f), 1023.0f) >> 5, cy = (int)fminf(fmaxf((qy - h.org[1]) * h.key_sc, 0.0f), 1023.0f) >> 5,
                      cz = (int)fminf(fmaxf((qz - h.org[2]) * h.key_sc, 0.0f), 1023.0f) >> 5;
            const int32_t g5 = tv.grid()[grid_index(5, cx, cy, cz)];
            if (g5 != GRID_EMPTY) r = g5;
            int dn = 0;
            while (r >= 0) {
              const NodeX& nd = tv.nodes[r];
              float dm = INFINITY; int32_t rm = NO_CHILD;
              for (int k = 0; k < 4; k++)
                if (nd.child[k] != NO_CHILD) { float dk = boxd2_q(la.gq, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k], h.scl2); if (dk < dm) { dm = dk; rm = nd.child[k]; } }
              r = rm; dn++;
            }
            Nn1Collector c1{inf_f(), 0x7fffffff};
            scan_leaf(tv, r, qx, qy, qz, c1);
            desc_nodes[ln] = dn;
            lc.col = Nn1CertCollector{c1.bd, c1.bi, inf_f()};
          }
          lc.gq = la.gq; lc.q[0] = qx; lc.q[1] = qy; lc.q[2] = qz; lc.ws.sp = 0; lc.ref = h.root;
          const int32_t g2 = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), qx, qy, qz, lc.col, [&](uint32_t key, int32_t r) { lc.ws.push(key, r); });
          if (g2 != GRID_USE_ROOT) lc.ref = (g2 == GRID_EMPTY) ? lc.ws.pop(lc.col) : g2;
        }
        if (ln == 63 || i == n - 1) {
          wave_lockstep(tv, h, LA, a_node, a_leaf, a_busy);
          wave_lockstep(tv, h, LB, g_node, g_leaf, g_busy, true);
          if (s == 0) {
            int dmax = 0; for (int k = 0; k < 64; k++) dmax = std::max(dmax, desc_nodes[k]);
            c_desc += dmax + 1;
            wave_lockstep(tv, h, LC, c_node, c_leaf, c_busy);
            for (int k = 0; k < 64; k++) if (LC[k].col.bi != LA[k].col.bi) c_mism++;
          }
        }
      }
      Nn1CertCollector ca{cd2, cand, inf_f()};   // (a) from the root
      Walk wa;
      walk_from(tv, h.root, qx, qy, qz, ca, wa);
      Nn1CertCollector cb{cd2, cand, inf_f()};   // (b) from the start grid: the product's grid_start, then the same step-wise walk
      Walk wgk;
      {
        GridQuery gq = grid_query(h, qx, qy, qz);
        std::vector<uint64_t> st(LDS_STACK);
        WalkStack<LDS_STACK> ws(st.data(), 1);
        int pushes = 0;
        int32_t ref = h.root;
        const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), qx, qy, qz, cb, [&](uint32_t key, int32_t r) { ws.push(key, r); pushes++; });
        if (g == GRID_USE_ROOT) use_root++;
        else ref = (g == GRID_EMPTY) ? ws.pop(cb) : g;
        sum_push += pushes;
        for (;;) {
          while (ref >= 0 && ref != NO_CHILD) { ref = node_visit(tv.nodes[ref], gq, h.scl2, cb, ws); wgk.nodes++; }
          if (ref == NO_CHILD) break;
          scan_leaf(tv, ref, qx, qy, qz, cb); wgk.leaves++;
          ref = ws.pop(cb);
        }
      }
      if (ca.bi != cb.bi || ca.bd != cb.bd) mism++;
      if (!(cb.lb >= cb.bd)) lbbad++;   // a bound on every other point is never below the winner's distance ...
      if (i % 37 == 0) {                // ... and never above the true runner-up (exhaustive, on a sample)
        float d1 = INFINITY, d2 = INFINITY;
        for (int j = 0; j < m; j++) {
          float dd = d2f(qx, qy, qz, tp[j].x, tp[j].y, tp[j].z);
          if (dd < d1) { d2 = d1; d1 = dd; } else if (dd < d2) d2 = dd;
        }
        if (d1 != cb.bd || cb.lb > d2 || ca.lb > d2) lbbad++;
      }
      prev[i] = ca.bi;
      int sb = wa.nodes + wa.leaves, sg = wgk.nodes + wgk.leaves;
      sum_b += sb; sum_g += sg;
      bnodes += wa.nodes; bleaves += wa.leaves; gnodes += wgk.nodes; gleaves += wgk.leaves;
      wb = std::max(wb, sb); wg = std::max(wg, sg);
      if ((i & 63) == 63 || i == n - 1) { wm_b += wb; wm_g += wg; waves++; wb = wg = 0; }
    }
    printf("sweep %d: root %.2f steps (%.2f nodes + %.2f leaves), wave-max %.1f | grid %.2f steps (%.2f + %.2f), wave-max %.1f | "
           "%.2f neighbour cells pushed per query, %.2f %% walk from the root | mismatches %ld, bad certificate bounds %ld\n",
           s, sum_b / n, (double)bnodes / n, (double)bleaves / n, wm_b / waves, sum_g / n, (double)gnodes / n, (double)gleaves / n,
           wm_g / waves, sum_push / n, 100.0 * use_root / n, mism, lbbad);
    printf("         lockstep wave: root %.1f node + %.1f leaf iterations (%.1f lanes busy) | grid %.1f + %.1f (%.1f lanes busy)\n",
           (double)a_node / waves, (double)a_leaf / waves, (double)a_busy / (a_node + a_leaf), (double)g_node / waves, (double)g_leaf / waves,
           (double)g_busy / (g_node + g_leaf));
    {
      long tot = 0; for (int k = 0; k <= 64; k++) tot += g_hist[k];
      printf("         grid walk, iterations by busy lanes: <=2 %.1f %%, <=4 %.1f %%, <=8 %.1f %%, <=16 %.1f %%, <=32 %.1f %% of %.1f per wave;", 
             100.0 * (g_hist[1] + g_hist[2]) / tot, 100.0 * (g_hist[1] + g_hist[2] + g_hist[3] + g_hist[4]) / tot,
             [&]{ long a = 0; for (int k = 1; k <= 8; k++) a += g_hist[k]; return 100.0 * a / tot; }(),
             [&]{ long a = 0; for (int k = 1; k <= 16; k++) a += g_hist[k]; return 100.0 * a / tot; }(),
             [&]{ long a = 0; for (int k = 1; k <= 32; k++) a += g_hist[k]; return 100.0 * a / tot; }(), (double)tot / waves);
      printf(" stop at <= 2 / 4 / 8 / 16 busy lanes: %.1f / %.1f / %.1f / %.1f iterations, %.2f / %.2f / %.2f / %.2f lanes handed off per wave\n",
             (double)g_cut_iters[1] / waves, (double)g_cut_iters[2] / waves, (double)g_cut_iters[3] / waves, (double)g_cut_iters[4] / waves,
             (double)g_cut_lanes[1] / waves, (double)g_cut_lanes[2] / waves, (double)g_cut_lanes[3] / waves, (double)g_cut_lanes[4] / waves);
      for (int k = 0; k <= 64; k++) g_hist[k] = 0;
      for (int k = 0; k < 5; k++) g_cut_iters[k] = g_cut_lanes[k] = 0;
    }
    if (s == 0)
      printf("         no seed pass (own descent from the level-5 cell, then the grid walk): %.1f descent + %.1f node + %.1f leaf iterations (%.1f lanes busy in the walk), mismatches %ld\n",
             (double)c_desc / waves, (double)c_node / waves, (double)c_leaf / waves, (double)c_busy / (c_node + c_leaf), c_mism);
  }
  return 0;
}
