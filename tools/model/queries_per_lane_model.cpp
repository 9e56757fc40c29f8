// each lane carries Q queries at once (no refill): a wave = 64 lanes x Q queries; every wave iteration is a node step or a leaf step (vote over all
// live queries); in it a lane advances ONE of its queries that is in that state.  Wave iterations per 64 queries.
#define TRAVERSAL_CHECK_NO_MAIN
#include "../../tests/host_emu/traversal_check.cpp"
#include <cstring>
#include <string>
static std::vector<float> read_f32(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb"); if (!f) exit(1);
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<float> v(sz / 4); if (fread(v.data(), 4, v.size(), f) != v.size()) exit(1); fclose(f); return v;
}
struct Q1 { Nn1CertCollector col{inf_f(), 0x7fffffff, inf_f()}; std::vector<uint64_t> mem; WalkStack<LDS_STACK> ws; GridQuery gq{0,0,0,0.f}; float q[3]; int32_t ref = NO_CHILD; int qi = -1;
  Q1() : mem(LDS_STACK), ws(mem.data(), 1) {} };
int main(int argc, char** argv) {
  std::string dir = argv[1];
  int NQ = atoi(argv[2]), policy = argc > 3 ? atoi(argv[3]) : 1;
  auto tg = read_f32(dir + "/tgt.f32"), sr = read_f32(dir + "/src.f32"), po = read_f32(dir + "/poses.f32");
  int m = tg.size() / 3, n = sr.size() / 3, np = po.size() / 12;
  std::vector<float4> tp(m);
  for (int i = 0; i < m; i++) tp[i] = make_float4(tg[3*i], tg[3*i+1], tg[3*i+2], 1.f);
  HostTree t = build(tp); TreeView tv = t.view(); const TreeHeader h = t.hdr();
  std::vector<int> prev(n, -1);
  for (int s = 0; s < np && s < 3; s++) {
    const float* T = &po[12 * s];
    if (s == 0) for (int i = 0; i < n; i += 4) { float qx,qy,qz; xform_pt(T, sr[3*i], sr[3*i+1], sr[3*i+2], qx,qy,qz); Nn1Collector cd{inf_f(), 0x7fffffff}; tree_descend<Nn1Collector, true>(tv, qx,qy,qz, cd); for (int e=0;e<4&&i+e<n;e++) prev[i+e]=cd.bi; }
    long iters = 0, busy = 0, nodeit = 0, leafit = 0;
    std::vector<int> nxt(n);
    const int W = 64 * NQ;
    for (int b0 = 0; b0 < n; b0 += W) {
      std::vector<Q1> L(W);   // query j of lane l = L[j * 64 + l]  (consecutive points share a slot index: coalesced loads)
      for (int k = 0; k < W && b0 + k < n; k++) {
        Q1& l = L[k]; int i = b0 + k; l.qi = i;
        float qx,qy,qz; xform_pt(T, sr[3*i], sr[3*i+1], sr[3*i+2], qx,qy,qz);
        int cand = prev[i];
        l.col = Nn1CertCollector{d2f(qx,qy,qz,tp[cand].x,tp[cand].y,tp[cand].z), cand, inf_f()};
        l.gq = grid_query(h, qx,qy,qz); l.q[0]=qx; l.q[1]=qy; l.q[2]=qz; l.ws.sp = 0; l.ref = h.root;
        const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), qx,qy,qz, l.col, [&](uint32_t key, int32_t r) { l.ws.push(key, r); });
        if (g != GRID_USE_ROOT) l.ref = (g == GRID_EMPTY) ? l.ws.pop(l.col) : g;
      }
      for (;;) {
        // per lane: does it hold a node-state query / a leaf-state query?
        int ln = 0, ll = 0;
        for (int l = 0; l < 64; l++) {
          bool hn = false, hl = false;
          for (int j = 0; j < NQ; j++) { int32_t r = L[j * 64 + l].ref; if (r >= 0 && r != NO_CHILD) hn = true; else if (r < 0) hl = true; }
          ln += hn; ll += hl;
        }
        if (!ln && !ll) break;
        bool do_node = policy == 0 ? ln > 0 : ln >= ll;
        iters++;
        if (do_node) { nodeit++; busy += ln; } else { leafit++; busy += ll; }
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < NQ; j++) {
            Q1& q = L[j * 64 + l];
            if (do_node && q.ref >= 0 && q.ref != NO_CHILD) { q.ref = node_visit(tv.nodes[q.ref], q.gq, h.scl2, q.col, q.ws); break; }
            if (!do_node && q.ref < 0) { scan_leaf(tv, q.ref, q.q[0], q.q[1], q.q[2], q.col); q.ref = q.ws.pop(q.col); break; }
          }
      }
      for (auto& l : L) if (l.qi >= 0) nxt[l.qi] = l.col.bi;
    }
    double w = n / 64.0;
    printf("sweep %d, %d queries per lane, policy %d: %.2f wave iterations per 64 queries (%.2f node + %.2f leaf), %.1f lanes busy\n", s, NQ, policy, iters / w, nodeit / w, leafit / w, (double)busy / iters);
    prev = nxt;
  }
}
