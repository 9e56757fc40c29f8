// Host-side check of the per-thread device functions in locus_amd/csrc/lh_device.hpp (tree layout + exact
// NN / k-NN traversal, tie rule) against brute force.  Compiled as a HOST program (no kernels, no HIP API
// calls); the tree is built here with plain loops following the same layout the build kernels produce.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../locus_amd/csrc/lh_device.hpp"

using namespace lh;

struct HostTree {
  std::vector<float4> sorted;
  std::vector<Node4> nodes;
  int depth = 0, first_leaf = 0, n = 0;
  TreeView view() const { return TreeView{sorted.data(), nodes.data(), first_leaf, n}; }
};

static int level_offset(int l) { return (int)(((1ll << (2 * l)) - 1) / 3); }

static HostTree build(const std::vector<float4>& pts) {
  HostTree t;
  int n = (int)pts.size();
  t.n = n;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (auto& p : pts) {
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  std::vector<std::pair<uint32_t, uint32_t>> kv(n);
  for (int i = 0; i < n; i++) kv[i] = {spatial_key30(pts[i].x, pts[i].y, pts[i].z, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]), (uint32_t)i};
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  int n_leaves = (n + LEAF - 1) / LEAF;
  int depth = 0;
  while ((1ll << (2 * depth)) < n_leaves) depth++;
  t.depth = depth;
  t.first_leaf = level_offset(depth);
  t.sorted.resize((size_t)n_leaves * LEAF);
  for (int i = 0; i < n_leaves * LEAF; i++) {
    if (i < n) {
      uint32_t j = kv[i].second;
      t.sorted[i] = make_float4(pts[j].x, pts[j].y, pts[j].z, u2f(j));
    } else
      t.sorted[i] = make_float4(INFINITY, INFINITY, INFINITY, u2f(0x7fffffffu));
  }
  t.nodes.resize(std::max(1, t.first_leaf));
  if (depth > 0) {
    int slots = 1 << (2 * depth);
    for (int L = 0; L < slots; L++) {
      float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int e = 0; e < LEAF; e++)
        if (L * LEAF + e < n) {
          float4 p = t.sorted[L * LEAF + e];
          l3[0] = fminf(l3[0], p.x); h3[0] = fmaxf(h3[0], p.x);
          l3[1] = fminf(l3[1], p.y); h3[1] = fmaxf(h3[1], p.y);
          l3[2] = fminf(l3[2], p.z); h3[2] = fmaxf(h3[2], p.z);
        }
      Node4& nd = t.nodes[level_offset(depth - 1) + (L >> 2)];
      int c = L & 3;
      nd.lox[c] = l3[0]; nd.loy[c] = l3[1]; nd.loz[c] = l3[2];
      nd.hix[c] = h3[0]; nd.hiy[c] = h3[1]; nd.hiz[c] = h3[2];
    }
    for (int l = depth - 2; l >= 0; l--)
      for (int tt = 0; tt < (1 << (2 * l + 2)); tt++) {
        int j = tt >> 2, c = tt & 3;
        const Node4& ch = t.nodes[level_offset(l + 1) + 4 * j + c];
        Node4& nd = t.nodes[level_offset(l) + j];
        nd.lox[c] = fminf(fminf(ch.lox[0], ch.lox[1]), fminf(ch.lox[2], ch.lox[3]));
        nd.loy[c] = fminf(fminf(ch.loy[0], ch.loy[1]), fminf(ch.loy[2], ch.loy[3]));
        nd.loz[c] = fminf(fminf(ch.loz[0], ch.loz[1]), fminf(ch.loz[2], ch.loz[3]));
        nd.hix[c] = fmaxf(fmaxf(ch.hix[0], ch.hix[1]), fmaxf(ch.hix[2], ch.hix[3]));
        nd.hiy[c] = fmaxf(fmaxf(ch.hiy[0], ch.hiy[1]), fmaxf(ch.hiy[2], ch.hiy[3]));
        nd.hiz[c] = fmaxf(fmaxf(ch.hiz[0], ch.hiz[1]), fmaxf(ch.hiz[2], ch.hiz[3]));
      }
  }
  return t;
}

static int run_case(int n, int nq, int k, unsigned seed, bool dup) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> g(0.f, 1.f);
  std::vector<float4> pts(n), qs(nq);
  for (int i = 0; i < n; i++) {
    if (dup && i > 0 && (i % 3) == 0) { pts[i] = pts[i - 1]; continue; }  // exact duplicates exercise the tie rule
    float cx = (float)((i % 7) * 3), cy = (float)((i % 5) * 2);
    pts[i] = make_float4(cx + g(rng), cy + g(rng), 0.2f * g(rng), 1.f);
  }
  for (int i = 0; i < nq; i++) {
    if (i % 4 == 0) qs[i] = pts[rng() % n];  // exact hits
    else qs[i] = make_float4(10.f + 8.f * g(rng), 4.f + 5.f * g(rng), g(rng), 1.f);
  }
  HostTree t = build(pts);
  TreeView tv = t.view();
  int bad = 0;
  std::vector<uint32_t> stk(STACK_MAX);
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
  printf("n=%d nq=%d k=%d dup=%d depth=%d -> %s (%d mismatches)\n", n, nq, k, (int)dup, t.depth, bad ? "FAIL" : "ok", bad);
  return bad;
}

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
  printf(bad ? "TRAVERSAL_CHECK_FAILED\n" : "TRAVERSAL_CHECK_OK\n");
  return bad ? 1 : 0;
}
