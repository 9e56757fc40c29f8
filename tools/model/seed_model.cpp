// cold sweep: candidate quality. mode 0: group shares the seed's neighbour j; mode 1: point e of the group takes j + e (the target is a ring-ordered scan too)
#define TRAVERSAL_CHECK_NO_MAIN
#include "../../tests/host_emu/traversal_check.cpp"
#include <cstring>
#include <string>
static std::vector<float> read_f32(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb"); if (!f) exit(1);
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<float> v(sz / 4); if (fread(v.data(), 4, v.size(), f) != v.size()) exit(1); fclose(f); return v;
}
struct Q1 { Nn1CertCollector col{inf_f(), 0x7fffffff, inf_f()}; std::vector<uint64_t> mem; WalkStack<LDS_STACK> ws; GridQuery gq{0,0,0,0.f}; float q[3]; int32_t ref = NO_CHILD;
  Q1() : mem(LDS_STACK), ws(mem.data(), 1) {} };
int main(int argc, char** argv) {
  std::string dir = argv[1];
  int group = atoi(argv[2]), mode = atoi(argv[3]);
  auto tg = read_f32(dir + "/tgt.f32"), sr = read_f32(dir + "/src.f32"), po = read_f32(dir + "/poses.f32");
  int m = tg.size() / 3, n = sr.size() / 3;
  std::vector<float4> tp(m);
  for (int i = 0; i < m; i++) tp[i] = make_float4(tg[3*i], tg[3*i+1], tg[3*i+2], 1.f);
  HostTree t = build(tp); TreeView tv = t.view(); const TreeHeader h = t.hdr();
  const float* T = &po[0];
  std::vector<int> cand(n);
  for (int i = 0; i < n; i += group) {
    float qx,qy,qz; xform_pt(T, sr[3*i], sr[3*i+1], sr[3*i+2], qx,qy,qz);
    Nn1Collector cd{inf_f(), 0x7fffffff}; tree_descend<Nn1Collector, true>(tv, qx,qy,qz, cd);
    for (int e = 0; e < group && i + e < n; e++) cand[i+e] = mode == 0 ? cd.bi : std::min(m - 1, std::max(0, cd.bi + (mode == 1 ? e : -e)));
  }
  long iters = 0, busy = 0; double sumd = 0, steps = 0;
  for (int b0 = 0; b0 < n; b0 += 64) {
    std::vector<Q1> L(64);
    for (int k = 0; k < 64 && b0 + k < n; k++) {
      Q1& l = L[k]; int i = b0 + k;
      float qx,qy,qz; xform_pt(T, sr[3*i], sr[3*i+1], sr[3*i+2], qx,qy,qz);
      int c = cand[i];
      float d0 = d2f(qx,qy,qz,tp[c].x,tp[c].y,tp[c].z); sumd += sqrt(d0);
      l.col = Nn1CertCollector{d0, c, inf_f()};
      l.gq = grid_query(h, qx,qy,qz); l.q[0]=qx; l.q[1]=qy; l.q[2]=qz; l.ws.sp = 0; l.ref = h.root;
      const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), qx,qy,qz, l.col, [&](uint32_t key, int32_t r) { l.ws.push(key, r); });
      if (g != GRID_USE_ROOT) l.ref = (g == GRID_EMPTY) ? l.ws.pop(l.col) : g;
    }
    for (;;) {
      int nn = 0, nl = 0; for (auto& l : L) { if (l.ref >= 0 && l.ref != NO_CHILD) nn++; else if (l.ref < 0) nl++; }
      if (!nn && !nl) break;
      iters++;
      if (nn >= nl) { busy += nn; steps += nn; for (auto& l : L) if (l.ref >= 0 && l.ref != NO_CHILD) l.ref = node_visit(tv.nodes[l.ref], l.gq, h.scl2, l.col, l.ws); }
      else { busy += nl; steps += nl; for (auto& l : L) if (l.ref < 0) { scan_leaf(tv, l.ref, l.q[0], l.q[1], l.q[2], l.col); l.ref = l.ws.pop(l.col); } }
    }
  }
  printf("seed group %d, mode %d: mean candidate distance %.3f m, %.2f steps per query, %.2f wave iterations per 64 queries (%.1f lanes busy)\n", group, mode, sumd / n, steps / n, iters / (n / 64.0), (double)busy / iters);
}
