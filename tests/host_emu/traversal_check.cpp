// Host-side check of the per-thread device functions in locus_amd/csrc/lh_device.hpp (tree layout + exact
// NN / k-NN traversal, tie rule) against brute force.  Compiled as a HOST program (no kernels, no HIP API
// calls); the tree is built here with plain loops following the same layout the build kernels produce.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#include "../../locus_amd/csrc/lh_device.hpp"

using namespace lh;

struct HostTree {
  std::vector<float4> sorted;
  std::vector<NodeX> buf;          // the device layout: [TreeHeader][start grid: GRID_NODEX slots][nodes]
  NodeX* nodes = nullptr;
  TreeHeader& hdr() { return *reinterpret_cast<TreeHeader*>(buf.data()); }
  const TreeHeader& hdr() const { return *reinterpret_cast<const TreeHeader*>(buf.data()); }
  int32_t* grid() { return reinterpret_cast<int32_t*>(buf.data() + 1); }
  int n = 0, n_leaves = 0, depth = 0;
  std::vector<uint64_t> lkey;      // leaf keys (+ one sentinel), leaf start positions (+ n), binary radix nodes: children / leaf ranges
  std::vector<uint32_t> lstart;
  std::vector<int> ich, irg;
  TreeView view() const { return TreeView{sorted.data(), nodes, &hdr(), n}; }
};

// serial restatement of the build kernels (k_key_b / k_leafcell_b / scan / k_leafrec_b / k_radix_b / k_boxes_b / k_nodex_b)
// with the SAME per-element functions (leafcell_flag, radix_node, leaf_ref) the kernels call
static HostTree build(const std::vector<float4>& pts, uint64_t cloud_id = 3) {
  HostTree t;
  int n = (int)pts.size();
  t.n = n;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (auto& p : pts) {
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  t.buf.assign((size_t)1 + GRID_NODEX + std::max(n, 1), NodeX{});
  t.nodes = t.buf.data() + 1 + GRID_NODEX;
  quant_frame(lo, hi, &t.hdr());  // k_key_b
  for (int k = 0; k < GRID_ENTRIES; k++) t.grid()[k] = GRID_EMPTY;
  std::vector<std::pair<uint64_t, uint32_t>> kv(n);
  for (int i = 0; i < n; i++)
    kv[i] = {(cloud_id << 32) | spatial_key30(pts[i].x, pts[i].y, pts[i].z, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]), (uint32_t)i};
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  std::vector<uint64_t> keys(n);
  t.sorted.resize((size_t)n + LEAF_CAP);
  for (int i = 0; i < n + LEAF_CAP; i++) {
    if (i < n) {
      uint32_t j = kv[i].second;
      keys[i] = kv[i].first;
      t.sorted[i] = make_float4(pts[j].x, pts[j].y, pts[j].z, u2f(j));
    } else
      t.sorted[i] = make_float4(INFINITY, INFINITY, INFINITY, u2f(0x7fffffffu));
  }
  std::vector<uint64_t> lkey;
  std::vector<uint32_t> lstart;
  for (int g = 0; g < n; g++)
    if (leafcell_flag(keys.data(), n, g)) { lkey.push_back(keys[g]); lstart.push_back((uint32_t)g); }
  int L = (int)lkey.size();
  lstart.push_back((uint32_t)n);
  lkey.push_back(~0ull);
  t.n_leaves = L;
  struct B6 { float v[6]; };
  auto leaf_box = [&](int l) {
    B6 b{{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}};
    for (uint32_t g = lstart[l]; g < lstart[l + 1]; g++) {
      float4 p = t.sorted[g];
      b.v[0] = fminf(b.v[0], p.x); b.v[1] = fminf(b.v[1], p.y); b.v[2] = fminf(b.v[2], p.z);
      b.v[3] = fmaxf(b.v[3], p.x); b.v[4] = fmaxf(b.v[4], p.y); b.v[5] = fmaxf(b.v[5], p.z);
    }
    return b;
  };
  for (int l = 0; l < L; l++)
    if ((int)(lstart[l + 1] - lstart[l]) > LEAF_CAP || lstart[l + 1] <= lstart[l]) { printf("bad leaf size\n"); exit(2); }
  if (L == 1) {
    t.hdr().root = leaf_ref(0u, n);
    t.hdr().n_leaves = 1;
    return t;
  }
  std::vector<int> ich(2 * (L - 1)), irg(2 * (L - 1));
  std::vector<int> icom(L - 1);
  for (int i = 0; i < L - 1; i++) {
    int delta;
    radix_node(lkey.data(), L, i, ich[2 * i], ich[2 * i + 1], irg[2 * i], irg[2 * i + 1], &delta);
    icom[i] = key_common(delta);
  }
  // boxes by recursion from the root (the kernels climb from the leaves with arrival counters; same result)
  std::vector<B6> ibox(L - 1);
  std::vector<int> idepth(L - 1, 0);
  std::function<B6(int, int)> boxof = [&](int ref, int dep) -> B6 {
    if (ref < 0) return leaf_box(~ref);
    B6 a = boxof(ich[2 * ref], dep + 1), b = boxof(ich[2 * ref + 1], dep + 1), m;
    for (int k = 0; k < 3; k++) { m.v[k] = fminf(a.v[k], b.v[k]); m.v[3 + k] = fmaxf(a.v[3 + k], b.v[3 + k]); }
    ibox[ref] = m;
    idepth[ref] = dep;
    t.depth = std::max(t.depth, dep);
    return m;
  };
  boxof(0, 0);
  if (irg[0] != 0 || irg[1] != L - 1) { printf("root range wrong\n"); exit(2); }
  for (int i = 0; i < L - 1; i++) {
    NodeX nd;
    for (int k = 0; k < 4; k++) { nd.lo_xy[k] = 0xffffffffu; nd.hi_xy[k] = 0u; nd.z_lohi[k] = 0xffffffffu; nd.child[k] = NO_CHILD; }
    int cnt = 0;
    auto emit = [&](int ref) {
      B6 b = ref < 0 ? leaf_box(~ref) : ibox[ref];
      quant_box(t.hdr(), b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], nd.lo_xy[cnt], nd.hi_xy[cnt], nd.z_lohi[cnt]);
      // the decoded box must enclose the float box with at least half a step to spare (double arithmetic = ground truth)
      const uint32_t qlo[3] = {nd.lo_xy[cnt] & 0xffffu, nd.lo_xy[cnt] >> 16, nd.z_lohi[cnt] & 0xffffu};
      const uint32_t qhi[3] = {nd.hi_xy[cnt] & 0xffffu, nd.hi_xy[cnt] >> 16, 65535u - (nd.z_lohi[cnt] >> 16)};
      for (int a = 0; a < 3; a++) {
        double dlo = (double)t.hdr().org[a] + (double)qlo[a] * (double)t.hdr().scl;
        double dhi = (double)t.hdr().org[a] + (double)qhi[a] * (double)t.hdr().scl;
        double half = 0.5 * (double)t.hdr().scl;
        bool lo_ok = qlo[a] == 0 ? dlo <= (double)b.v[a] : dlo + half <= (double)b.v[a];
        if (!lo_ok || dhi - half < (double)b.v[3 + a]) { printf("quantised box does not enclose: axis %d\n", a); exit(2); }
      }
      nd.child[cnt] = ref < 0 ? leaf_ref(lstart[~ref], (int)(lstart[~ref + 1] - lstart[~ref])) : ref;
      cnt++;
    };
    for (int side = 0; side < 2; side++) {
      int c = ich[2 * i + side];
      if (c < 0) emit(c);
      else { emit(ich[2 * c]); emit(ich[2 * c + 1]); }
    }
    t.nodes[i] = nd;
  }
  t.hdr().root = 0;  // Karras: node 0 covers every leaf
  t.hdr().n_leaves = L;
  // the start grid (k_nodex_b): every binary node offers its two children, the root itself where the whole cloud is one cell.  (The host
  // builds a 4-ary node for EVERY binary node with plain grandchild adoption, so every cell root has one; the device builds only the
  // reachable ones -- its rule is exercised by the GPU tests against the exhaustive search.)
  auto ref_of = [&](int c) -> int32_t { return c < 0 ? leaf_ref(lstart[~c], (int)(lstart[~c + 1] - lstart[~c])) : c; };
  for (int i = 0; i < L - 1; i++)
    for (int side = 0; side < 2; side++) {
      const int c = ich[2 * i + side];
      grid_fill_child(t.grid(), icom[i], c < 0, c < 0 ? 30 : icom[c], (uint32_t)lkey[c < 0 ? ~c : irg[2 * c]] & 0x3fffffffu, ref_of(c));
    }
  grid_fill_root(t.grid(), icom[0], (uint32_t)lkey[0] & 0x3fffffffu, 0);
  t.hdr().grid_on = 1;
  t.lkey = lkey; t.lstart = lstart; t.ich = ich; t.irg = irg;
  return t;
}

static int run_case(int n, int nq, int k, unsigned seed, int dup) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> g(0.f, 1.f);
  std::vector<float4> pts(n), qs(nq);
  for (int i = 0; i < n; i++) {
    if (dup == 1 && i > 0 && (i % 3) == 0) { pts[i] = pts[i - 1]; continue; }  // exact duplicates exercise the tie rule
    if (dup == 2 && i > 0 && (i % 50) != 0) { pts[i] = pts[i - 1]; continue; }  // runs of 50 identical points: chunked leaves, index tie-breaks in the radix tree
    float cx = (float)((i % 7) * 3), cy = (float)((i % 5) * 2);
    pts[i] = make_float4(cx + g(rng), cy + g(rng), 0.2f * g(rng), 1.f);
  }
  for (int i = 0; i < nq; i++) {
    if (i % 4 == 0) qs[i] = pts[rng() % n];  // exact hits
    else if (i % 16 == 1) qs[i] = make_float4(-3.f - 0.5f * fabsf(g(rng)), 4.f + 5.f * g(rng), g(rng), 1.f);  // just outside the cloud's box (the grid's e2 term)
    else if (i % 16 == 5) qs[i] = make_float4(3.0e4f * g(rng), 2.0e4f * g(rng), 1.0e3f * g(rng), 1.f);      // hundreds of extents away
    else if (i % 16 == 9) qs[i] = make_float4(1.0e12f, -1.0e12f, 1.0e12f, 1.f);                             // absurdly far: e2 stays finite
    else qs[i] = make_float4(10.f + 8.f * g(rng), 4.f + 5.f * g(rng), g(rng), 1.f);
  }
  HostTree t = build(pts);
  TreeView tv = t.view();
  int bad = 0;
  // the quantised box bound must never exceed the float distance to any point of the leaf it bounds (what pruning relies on)
  if (t.n_leaves > 1) {
    for (int i = 0; i < std::min(nq, 64); i++) {
      GridQuery gq = grid_query(t.hdr(), qs[i].x, qs[i].y, qs[i].z);
      for (int nd = 0; nd < t.n_leaves - 1; nd++)
        for (int c = 0; c < 4; c++) {
          int32_t ref = t.nodes[nd].child[c];
          if (ref == NO_CHILD || ref >= 0) continue;
          float bnd = boxd2_q(gq, t.nodes[nd].lo_xy[c], t.nodes[nd].hi_xy[c], t.nodes[nd].z_lohi[c], t.hdr().scl2);
          uint32_t u = (uint32_t)~ref;
          for (uint32_t e = 0; e <= (u & 15u); e++) {
            float4 p = t.sorted[(u >> 4) + e];
            if (!(bnd <= d2f(qs[i].x, qs[i].y, qs[i].z, p.x, p.y, p.z))) { printf("box bound above a point distance\n"); bad++; }
          }
        }
    }
  }
  std::vector<uint64_t> stk(LDS_STACK);
  std::vector<float> kd(k), bd(n);
  std::vector<int> ki(k), ord(n);
  for (int i = 0; i < nq; i++) {
    float4 q = qs[i];
    // brute force, lexicographic (d2, idx)
    for (int j = 0; j < n; j++) { bd[j] = d2f(q.x, q.y, q.z, pts[j].x, pts[j].y, pts[j].z); ord[j] = j; }
    int kk = std::min(k, n);
    std::partial_sort(ord.begin(), ord.begin() + kk, ord.end(), [&](int a, int b) { return bd[a] < bd[b] || (bd[a] == bd[b] && a < b); });
    Nn1Collector c1{inf_f(), 0x7fffffff};
    tree_search(tv, q.x, q.y, q.z, c1, stk.data(), 1);
    if (c1.bi != ord[0] || c1.bd != bd[ord[0]]) bad++;
    // warm start from an arbitrary valid candidate must give the same answer
    int w = (int)(rng() % n);
    Nn1Collector c2{bd[w], w};
    tree_search(tv, q.x, q.y, q.z, c2, stk.data(), 1);
    if (c2.bi != ord[0] || c2.bd != bd[ord[0]]) bad++;
    {  // the seed pass's descent (tree_descend): ONE leaf, no backtracking -- a valid candidate (a real point at its real distance),
       // never closer than the true neighbour, and close to it for most queries
      Nn1Collector cd{inf_f(), 0x7fffffff};
      tree_descend(tv, q.x, q.y, q.z, cd);
      if (cd.bi < 0 || cd.bi >= n || cd.bd != bd[cd.bi] || cd.bd < bd[ord[0]]) bad++;
    }
    {  // the step-wise form of the same traversal (WalkStack + node_visit: what k_walk's persistent lanes run) must leave the
       // certificate collector in the same state as tree_search, warm (any valid candidate) or cold
      for (int warm = 0; warm < 2; warm++) {
        Nn1CertCollector ca{warm ? bd[w] : inf_f(), warm ? w : 0x7fffffff, inf_f()}, cb = ca;
        tree_search(tv, q.x, q.y, q.z, ca, stk.data(), 1);
        TreeHeader h = *tv.hdr;
        GridQuery gq = grid_query(h, q.x, q.y, q.z);
        std::vector<uint64_t> st2(4);
        WalkStack<4> ws(st2.data(), 1);   // a short in-"LDS" part so that the spill path is exercised too
        int32_t ref = h.root;
        for (;;) {
          while (ref >= 0 && ref != NO_CHILD) ref = node_visit(tv.nodes[ref], gq, h.scl2, cb, ws);
          if (ref == NO_CHILD) break;
          scan_leaf(tv, ref, q.x, q.y, q.z, cb);
          ref = ws.pop(cb);
        }
        if (ca.bi != cb.bi || ca.bd != cb.bd || ca.bi != ord[0]) bad++;
        // the certificate bound: tree_search's cold start makes a greedy descent first, which may examine (and so bound) other
        // points than the plain loop; warm walks are the same sequence of visits => the same bound, bit for bit
        if (warm && ca.lb != cb.lb) bad++;
        if (n > 1 && !(cb.lb >= bd[ord[0]]) ) bad++;   // a valid lower bound on every other point is never below the winner's distance
        if (n > 1 && cb.lb > bd[ord[1]]) bad++;        // ... and never above the true runner-up
      }
    }
    if (t.n_leaves > 1) {  // the sweeps' warm walks from the start grid (tree_search<.., true>): candidates near (the ball stays inside a few cells),
                           // far (coarser tables) and absurd (falls back to the root); neighbour AND certificate bound against brute force
      const int cands[4] = {ord[std::min(n - 1, 2)], ord[std::min(n - 1, 40)], w, ord[0]};
      for (int cidx = 0; cidx < 4; cidx++) {
        const int cw = cands[cidx];
        Nn1CertCollector cg{bd[cw], cw, inf_f()};
        tree_search<Nn1CertCollector, true>(tv, q.x, q.y, q.z, cg, stk.data(), 1);
        if (cg.bi != ord[0] || cg.bd != bd[ord[0]]) bad++;
        if (n > 1 && !(cg.lb >= bd[ord[0]])) bad++;
        if (n > 1 && cg.lb > bd[ord[1]]) bad++;
      }
      Nn1CertCollector cc{inf_f(), 0x7fffffff, inf_f()};   // cold: a descent below the query's own cell for a bound, then the grid walk (lh_nn1)
      tree_search<Nn1CertCollector, true>(tv, q.x, q.y, q.z, cc, stk.data(), 1);
      if (cc.bi != ord[0] || cc.bd != bd[ord[0]] || (n > 1 && (!(cc.lb >= bd[ord[0]]) || cc.lb > bd[ord[1]]))) bad++;
      Nn1Collector cs{inf_f(), 0x7fffffff};   // the seed descent from the query's own cell
      tree_descend<Nn1Collector, true>(tv, q.x, q.y, q.z, cs);
      if (cs.bi < 0 || cs.bi >= n || cs.bd != bd[cs.bi]) bad++;
    }
    KnnCollector ck{kd.data(), ki.data(), kk, 1, 0};
    tree_search(tv, q.x, q.y, q.z, ck, stk.data(), 1);
    if (ck.cnt != kk) bad++;
    for (int e = 0; e < kk; e++)
      if (ki[e] != ord[e] || kd[e] != bd[ord[e]]) { bad++; break; }
    {  // radius collector: the exact set of points with d2 < r2 (count) and their coordinate sum
      float r2 = (i % 3 == 0) ? 0.25f : (i % 3 == 1 ? 4.0f : bd[ord[std::min(n - 1, 5)]]);  // third flavour: r2 equal to a point distance
      RadiusMomentCollector rc;
      rc.r2 = r2; rc.cnt = 0;
      for (int e = 0; e < 9; e++) rc.a[e] = 0.f;
      tree_search(tv, q.x, q.y, q.z, rc, stk.data(), 1);
      int cnt = 0;
      double sx = 0;
      for (int j = 0; j < n; j++)
        if (bd[j] < r2) { cnt++; sx += pts[j].x; }
      if (rc.cnt != cnt || fabs(rc.a[6] - sx) > 1e-3 * (1.0 + fabs(sx))) bad++;
    }
    if (kk <= 20) {  // register-resident list used by the device k-NN kernels
      KnnRegCollector<20> cr;
      cr.init(kk);
      tree_search(tv, q.x, q.y, q.z, cr, stk.data(), 1);
      std::vector<float> rd(kk);
      std::vector<int> ri(kk);
      if (cr.dump(rd.data(), ri.data(), 1) != kk) bad++;
      for (int e = 0; e < kk; e++)
        if (ri[e] != ord[e] || rd[e] != bd[ord[e]]) { bad++; break; }
    }
  }
  printf("n=%d nq=%d k=%d dup=%d leaves=%d binary depth=%d -> %s (%d mismatches)\n", n, nq, k, (int)dup, t.n_leaves, t.depth, bad ? "FAIL" : "ok", bad);
  return bad;
}

#ifndef TRAVERSAL_CHECK_NO_MAIN
int main() {
  int bad = 0;
  bad += run_case(1, 16, 1, 1, false);
  bad += run_case(7, 64, 5, 2, false);
  bad += run_case(8, 64, 8, 3, true);
  bad += run_case(9, 64, 9, 4, false);
  bad += run_case(33, 128, 20, 5, true);
  bad += run_case(1000, 500, 20, 6, true);
  bad += run_case(20000, 800, 20, 7, false);
  bad += run_case(20000, 400, 20, 8, true);
  bad += run_case(3000, 300, 20, 9, 2);
  bad += run_case(70, 100, 8, 10, 2);
  printf(bad ? "TRAVERSAL_CHECK_FAILED\n" : "TRAVERSAL_CHECK_OK\n");
  return bad ? 1 : 0;
}
#endif
