// persistent-lane model: one wave serves SPAN consecutive queries; a lane whose walk has ended takes the next query once >= R lanes are idle;
// every wave iteration is a node step or a leaf step (majority vote).  Counts wave iterations per 64 queries.
#define TRAVERSAL_CHECK_NO_MAIN
#include "../../tests/host_emu/traversal_check.cpp"
#include <cstring>
#include <string>
static std::vector<float> read_f32(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb"); if (!f) exit(1);
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<float> v(sz / 4); if (fread(v.data(), 4, v.size(), f) != v.size()) exit(1); fclose(f); return v;
}
struct Lane { Nn1CertCollector col{inf_f(), 0x7fffffff, inf_f()}; std::vector<uint64_t> mem; WalkStack<LDS_STACK> ws; GridQuery gq{0,0,0,0.f}; float q[3]; int32_t ref = NO_CHILD; int qi = -1;
  Lane() : mem(LDS_STACK), ws(mem.data(), 1) {} };
int main(int argc, char** argv) {
  std::string dir = argv[1];
  int SPAN = atoi(argv[2]), R = atoi(argv[3]), policy = argc > 4 ? atoi(argv[4]) : 1;
  auto tg = read_f32(dir + "/tgt.f32"), sr = read_f32(dir + "/src.f32"), po = read_f32(dir + "/poses.f32");
  int m = tg.size() / 3, n = sr.size() / 3, np = po.size() / 12;
  std::vector<float4> tp(m);
  for (int i = 0; i < m; i++) tp[i] = make_float4(tg[3*i], tg[3*i+1], tg[3*i+2], 1.f);
  HostTree t = build(tp); TreeView tv = t.view(); const TreeHeader h = t.hdr();
  std::vector<int> prev(n, -1);
  for (int s = 0; s < np && s < 3; s++) {
    const float* T = &po[12 * s];
    if (s == 0) for (int i = 0; i < n; i += 4) { float qx,qy,qz; xform_pt(T, sr[3*i], sr[3*i+1], sr[3*i+2], qx,qy,qz); Nn1Collector cd{inf_f(), 0x7fffffff}; tree_descend<Nn1Collector, true>(tv, qx,qy,qz, cd); for (int e=0;e<4&&i+e<n;e++) prev[i+e]=cd.bi; }
    long iters = 0, refills = 0, refill_lanes = 0, busy = 0, nodeit = 0, leafit = 0;
    std::vector<int> nxt(n);
    for (int b0 = 0; b0 < n; b0 += SPAN) {
      int total = std::min(SPAN, n - b0), head = 0;
      std::vector<Lane> L(64);
      for (;;) {
        int nidle = 0; for (auto& l : L) nidle += l.ref == NO_CHILD;
        if (nidle == 64 || (head < total && nidle >= R)) {
          refills++;
          for (auto& l : L) if (l.ref == NO_CHILD) {
            if (l.qi >= 0) { nxt[b0 + l.qi] = l.col.bi; l.qi = -1; }
            if (head < total) {
              l.qi = head++; refill_lanes++;
              int i = b0 + l.qi; float qx,qy,qz; xform_pt(T, sr[3*i], sr[3*i+1], sr[3*i+2], qx,qy,qz);
              int cand = prev[i];
              l.col = Nn1CertCollector{d2f(qx,qy,qz,tp[cand].x,tp[cand].y,tp[cand].z), cand, inf_f()};
              l.gq = grid_query(h, qx,qy,qz); l.q[0]=qx; l.q[1]=qy; l.q[2]=qz; l.ws.sp = 0; l.ref = h.root;
              const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), qx,qy,qz, l.col, [&](uint32_t key, int32_t r) { l.ws.push(key, r); });
              if (g != GRID_USE_ROOT) l.ref = (g == GRID_EMPTY) ? l.ws.pop(l.col) : g;
            }
          }
          bool any = false; for (auto& l : L) any |= l.ref != NO_CHILD;
          if (!any) break;
        }
        int nn = 0, nl = 0; for (auto& l : L) { if (l.ref >= 0 && l.ref != NO_CHILD) nn++; else if (l.ref < 0) nl++; }
        bool do_node = policy == 0 ? nn > 0 : nn >= nl;
        if (do_node) { iters++; nodeit++; busy += nn; for (auto& l : L) if (l.ref >= 0 && l.ref != NO_CHILD) l.ref = node_visit(tv.nodes[l.ref], l.gq, h.scl2, l.col, l.ws); }
        else { iters++; leafit++; busy += nl; for (auto& l : L) if (l.ref < 0) { scan_leaf(tv, l.ref, l.q[0], l.q[1], l.q[2], l.col); l.ref = l.ws.pop(l.col); } }
      }
    }
    double w = n / 64.0;
    printf("sweep %d span %d refill>=%d policy %d: %.2f iterations per 64 queries (%.2f node + %.2f leaf), %.1f lanes busy, %.2f refill rounds per 64 queries (%.1f lanes each)\n", s, SPAN, R, policy, iters / w, nodeit / w, leafit / w, (double)busy / iters, refills / w, (double)refill_lanes / refills);
    prev = nxt;
  }
}
