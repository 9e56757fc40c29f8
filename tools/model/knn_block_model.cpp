// Host model of the block k-NN search (locus_amd/csrc/lh_knn_block.hpp): one "wave" = 64 lanes stepped in lockstep by plain
// loops, the product's own per-lane functions and networks, the tree built by tests/host_emu/traversal_check.cpp's serial restatement of
// the build kernels.  Checks every query's k-NN set (d2, index) against an exhaustive search and counts what a wave executes: chunks
// merged, nodes visited, remembered chunks, redo lanes -- the numbers the kernel's design (window size, child order) was chosen by.
//   g++ -O2 -std=c++17 -x c++ -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/model/knn_block_model.cpp -o /tmp/knn_block_model
//   /tmp/knn_block_model scan.bin [k] [win_side]      (scan.bin: float32 x,y,z,1 records; without arguments: built-in cases)
#define TRAVERSAL_CHECK_NO_MAIN
#include "../../tests/host_emu/traversal_check.cpp"
#include "../../locus_amd/csrc/lh_knn_block.hpp"

#include <cstring>

struct Stats {
  long blocks = 0, win_chunks = 0, win_merged = 0, nodes = 0, pops = 0, pop_pruned = 0, leaves_proc = 0, leaves_merged = 0, acc = 0, pass2_chunks = 0;
  long redo = 0, bad = 0, max_acc = 0, max_sp = 0, queries = 0;
  long fail_ties = 0, fail_chunks = 0, fail_stack = 0, fail_inf = 0;
};

template <int K>
static void run_block(const HostTree& t, const std::vector<float4>& pts, int b0, int k, int win_side, Stats& st, std::vector<int>& out_idx,
                      std::vector<float>& out_d2, std::vector<uint8_t>& redo) {
  const int n = t.n, W = KNN_BLOCK_Q;
  const TreeHeader& h = t.hdr();
  struct Lane { float qx, qy, qz; GridQuery gq; uint32_t L[K]; int pos; };
  Lane ln[KNN_BLOCK_Q];
  for (int l = 0; l < W; l++) {
    int pos = std::min(b0 + l, n - 1);
    float4 p = t.sorted[pos];
    ln[l].qx = p.x; ln[l].qy = p.y; ln[l].qz = p.z; ln[l].pos = pos;
    ln[l].gq = grid_query(h, p.x, p.y, p.z);
    knn_list_init<K>(ln[l].L, k);
  }
  uint32_t acc[KNN_ACC_CAP];
  int n_acc = 0, fail = 0;
  auto process = [&](int first, int cnt, bool is_win) {
    bool any_le = false, any_lt = false;
    uint32_t B[KNN_BLOCK_Q][8];
    for (int l = 0; l < W; l++) {
      knn_chunk_keys(ln[l].qx, ln[l].qy, ln[l].qz, t.sorted.data() + first, cnt, B[l]);
      uint32_t tau = ln[l].L[K - 1], m = knn_min8(B[l]);
      any_le |= m <= tau;
      any_lt |= m < tau;
    }
    if (any_le) {
      if (n_acc < KNN_ACC_CAP) acc[n_acc++] = knn_chunk_ref((uint32_t)first, cnt);
      else fail |= KNN_FAIL_CHUNKS;
    }
    if (any_lt) {
      for (int l = 0; l < W; l++) { knn_sort8(B[l]); KnnNet<K>::merge(ln[l].L, B[l]); }
      (is_win ? st.win_merged : st.leaves_merged)++;
    }
    (is_win ? st.win_chunks : st.leaves_proc)++;
  };
  const int w0 = std::max(0, b0 - win_side), w1 = std::min(n, b0 + W + win_side);
  // window: the block's own chunks first, then outwards
  {
    std::vector<int> starts;
    for (int c = w0; c < w1; c += 8) starts.push_back(c);
    std::stable_sort(starts.begin(), starts.end(), [&](int a, int b) {
      auto dist = [&](int c) { int mid = c + 4; return mid < b0 ? b0 - mid : (mid > b0 + W ? mid - (b0 + W) : 0); };
      return dist(a) < dist(b);
    });
    for (int c : starts) process(c, std::min(8, w1 - c), true);
  }
  // the wave's walk
  struct Ent { int32_t ref; uint32_t lo_xy, hi_xy, z_lohi; };
  Ent stack[KNN_STACK_CAP];
  int sp = 0;
  auto wanted = [&](uint32_t lo_xy, uint32_t hi_xy, uint32_t z_lohi) {
    int cnt = 0;
    for (int l = 0; l < W; l++) {
      float bd = boxd2_q(ln[l].gq, lo_xy, hi_xy, z_lohi, h.scl2);
      if (f2u(bd) <= ln[l].L[K - 1]) cnt++;   // bd >= 0: unsigned order of the bits = float order
    }
    return cnt;
  };
  auto visit_node = [&](int32_t ref) {
    st.nodes++;
    const NodeX& nd = t.nodes[ref];
    int w[4], top = 0;
    for (int c = 0; c < 4; c++) w[c] = nd.child[c] == NO_CHILD ? 0 : wanted(nd.lo_xy[c], nd.hi_xy[c], nd.z_lohi[c]);
    for (int c = 1; c < 4; c++)
      if (w[c] > w[top]) top = c;
    if (sp + 4 > KNN_STACK_CAP) { fail |= KNN_FAIL_STACK; return; }
    for (int c = 0; c < 4; c++)   // the child most lanes want goes on top, the others in child order
      if (w[c] && c != top) stack[sp++] = Ent{nd.child[c], nd.lo_xy[c], nd.hi_xy[c], nd.z_lohi[c]};
    if (w[top]) stack[sp++] = Ent{nd.child[top], nd.lo_xy[top], nd.hi_xy[top], nd.z_lohi[top]};
    st.max_sp = std::max<long>(st.max_sp, sp);
  };
  auto visit_leaf = [&](int32_t ref) {
    uint32_t u = (uint32_t)~ref;
    int first = (int)(u >> 4), cnt = (int)(u & 15u) + 1, f2, c2;
    if (!knn_clip_chunk(first, cnt, w0, w1, f2, c2)) return;
    process(f2, c2, false);
  };
  if (h.root < 0) visit_leaf(h.root);
  else {
    visit_node(h.root);
    while (sp > 0 && !fail) {
      Ent e = stack[--sp];
      st.pops++;
      if (!wanted(e.lo_xy, e.hi_xy, e.z_lohi)) { st.pop_pruned++; continue; }
      if (e.ref < 0) visit_leaf(e.ref);
      else visit_node(e.ref);
    }
  }
  st.acc += n_acc;
  st.max_acc = std::max<long>(st.max_acc, n_acc);
  // pass 2 + final sort
  for (int l = 0; l < W; l++) {
    if (b0 + l >= n) continue;
    st.queries++;
    int lane_fail = fail;
    const uint32_t tau = ln[l].L[K - 1];
    if (tau == KNN_KEY_INF && n >= k) lane_fail |= KNN_FAIL_INF;
    uint32_t table[K];
    int cnt = 0;
    if (!lane_fail) {
      for (int a = 0; a < n_acc; a++) {
        int first = (int)(acc[a] >> 4), c = (int)(acc[a] & 15u) + 1;
        uint32_t B[8];
        knn_chunk_keys(ln[l].qx, ln[l].qy, ln[l].qz, t.sorted.data() + first, c, B);
        for (int e = 0; e < 8; e++)
          if (B[e] <= tau && B[e] != KNN_KEY_INF) {
            if (cnt < K) table[cnt] = (uint32_t)(first + e);
            cnt++;
          }
      }
      if (cnt > std::min(k, K) || (n >= k && cnt != k)) lane_fail |= KNN_FAIL_TIES;
    }
    const int qid = (int)f2u(t.sorted[ln[l].pos].w);
    if (lane_fail == KNN_FAIL_TIES) {   // the wave settles ties itself: a (d2, index) insertion list over the remembered chunks
      st.fail_ties++;
      KnnRegCollector<K> col;
      col.init(std::min(k, K));
      for (int a = 0; a < n_acc; a++) {
        int first = (int)(acc[a] >> 4), c = (int)(acc[a] & 15u) + 1;
        for (int e = 0; e < c; e++) {
          float4 p = t.sorted[first + e];
          col.offer(d2f(ln[l].qx, ln[l].qy, ln[l].qz, p.x, p.y, p.z), (int)f2u(p.w));
        }
      }
      for (int j = 0; j < k; j++) {
        bool ok = j < K && col.id[j] != 0x7fffffff;
        out_idx[(size_t)qid * k + j] = ok ? col.id[j] : -1;
        out_d2[(size_t)qid * k + j] = ok ? col.d[j] : INFINITY;
      }
      continue;
    }
    if (lane_fail) {
      st.redo++;
      redo[qid] = 1;
      if (lane_fail & KNN_FAIL_CHUNKS) st.fail_chunks++;
      if (lane_fail & KNN_FAIL_STACK) st.fail_stack++;
      if (lane_fail & KNN_FAIL_INF) st.fail_inf++;
      continue;
    }
    uint64_t keys[K];
    for (int j = 0; j < K; j++) {
      if (j < cnt) {
        float4 p = t.sorted[table[j]];
        keys[j] = ((uint64_t)f2u(d2f(ln[l].qx, ln[l].qy, ln[l].qz, p.x, p.y, p.z)) << 32) | f2u(p.w);
      } else
        keys[j] = ~0ull;
    }
    KnnNet<K>::sort_pairs(keys);
    for (int j = 0; j < k; j++) {
      out_idx[(size_t)qid * k + j] = j < cnt ? (int)(uint32_t)keys[j] : -1;
      out_d2[(size_t)qid * k + j] = j < cnt ? u2f((uint32_t)(keys[j] >> 32)) : INFINITY;
    }
  }
  st.pass2_chunks += n_acc;
  st.blocks++;
}

template <int K>
static long run_cloud(const std::vector<float4>& pts, int k, int win_side, bool verify_all, const char* tag) {
  HostTree t = build(pts);
  const int n = (int)pts.size();
  Stats st;
  std::vector<int> idx((size_t)n * k, -2);
  std::vector<float> d2((size_t)n * k, -1.f);
  std::vector<uint8_t> redo(n, 0);
  for (int b0 = 0; b0 < n; b0 += KNN_BLOCK_Q) run_block<K>(t, pts, b0, k, win_side, st, idx, d2, redo);
  // redo lanes: the one-query-per-lane search
  std::vector<uint64_t> stk(LDS_STACK);
  for (int i = 0; i < n; i++)
    if (redo[i]) {
      KnnRegCollector<K> cr;
      cr.init(std::min(k, K));
      tree_search(t.view(), pts[i].x, pts[i].y, pts[i].z, cr, stk.data(), 1);
      for (int j = 0; j < k; j++) {
        bool ok = cr.id[j] != 0x7fffffff;
        idx[(size_t)i * k + j] = ok ? cr.id[j] : -1;
        d2[(size_t)i * k + j] = ok ? cr.d[j] : INFINITY;
      }
    }
  // exhaustive check (every query for small clouds, a sample for large ones)
  std::vector<float> bd(n);
  std::vector<int> ord(n);
  int step = verify_all ? 1 : std::max(1, n / 3000);
  for (int i = 0; i < n; i += step) {
    for (int j = 0; j < n; j++) { bd[j] = d2f(pts[i].x, pts[i].y, pts[i].z, pts[j].x, pts[j].y, pts[j].z); ord[j] = j; }
    int kk = std::min(k, n);
    std::partial_sort(ord.begin(), ord.begin() + kk, ord.end(), [&](int a, int b) { return bd[a] < bd[b] || (bd[a] == bd[b] && a < b); });
    for (int j = 0; j < k; j++) {
      int ei = j < kk ? ord[j] : -1;
      float ed = j < kk ? bd[ord[j]] : INFINITY;
      if (idx[(size_t)i * k + j] != ei || d2[(size_t)i * k + j] != ed) { st.bad++; break; }
    }
  }
  double B = (double)std::max<long>(1, st.blocks);
  printf("%s n=%d k=%d K=%d win=%d: %s  blocks=%ld  per block: window chunks %.1f (merged %.1f)  nodes %.1f  pops %.1f (pruned %.1f)  leaf chunks %.1f (merged %.1f)  "
         "remembered %.1f (max %ld)  max stack %ld | ties settled in the wave %ld, redo %ld of %ld (chunks %ld stack %ld inf %ld)\n",
         tag, n, k, K, win_side, st.bad ? "FAIL" : "ok", st.blocks, st.win_chunks / B, st.win_merged / B, st.nodes / B, st.pops / B, st.pop_pruned / B,
         st.leaves_proc / B, st.leaves_merged / B, st.acc / B, st.max_acc, st.max_sp, st.fail_ties, st.redo, st.queries, st.fail_chunks, st.fail_stack, st.fail_inf);
  // what a wave executes, in vector instructions (model: chunk = 64 distance + 12 bound/min ops; merge = 38 + merge network; node = 4 x 11 + 8; pop test 12;
  // pass-2 chunk = 64 + 8 x 5)
  const double merge_ops = K == 20 ? 96 : (K == 8 ? 32 : 152);
  double ops = (st.win_chunks + st.leaves_proc) * 76.0 + (st.win_merged + st.leaves_merged) * (38.0 + merge_ops) + st.nodes * 52.0 + st.pops * 12.0 + st.pass2_chunks * 104.0;
  printf("   modelled vector instructions per block: %.0f (+ ~1500 for the final sort, moments and eigen-solve)\n", ops / B);
  return st.bad;
}

static std::vector<float4> synth_cloud(int n, unsigned seed, int dup) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> g(0.f, 1.f);
  std::vector<float4> pts(n);
  for (int i = 0; i < n; i++) {
    if (dup == 1 && i > 0 && (i % 3) == 0) { pts[i] = pts[i - 1]; continue; }
    if (dup == 2 && i > 0 && (i % 50) != 0) { pts[i] = pts[i - 1]; continue; }
    if (dup == 3) { pts[i] = make_float4(0.1f * (float)(i % 10), 0.1f * (float)((i / 10) % 10), 0.1f * (float)(i / 100), 1.f); continue; }  // lattice: ties everywhere
    float cx = (float)((i % 7) * 3), cy = (float)((i % 5) * 2);
    pts[i] = make_float4(cx + g(rng), cy + g(rng), 0.2f * g(rng), 1.f);
  }
  return pts;
}

int main(int argc, char** argv) {
  long bad = 0;
  if (argc > 1) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<float4> pts(bytes / 16);
    if (fread(pts.data(), 16, pts.size(), f) != pts.size()) return 2;
    fclose(f);
    int k = argc > 2 ? atoi(argv[2]) : 20, win = argc > 3 ? atoi(argv[3]) : KNN_WIN_SIDE;
    if (k <= 8) bad += run_cloud<8>(pts, k, win, false, argv[1]);
    else if (k <= 20) bad += run_cloud<20>(pts, k, win, false, argv[1]);
    else bad += run_cloud<32>(pts, k, win, false, argv[1]);
  } else {
    const int W = KNN_WIN_SIDE;
    bad += run_cloud<20>(synth_cloud(1, 1, 0), 20, W, true, "one point");
    bad += run_cloud<20>(synth_cloud(7, 2, 0), 5, W, true, "seven");
    bad += run_cloud<8>(synth_cloud(8, 3, 1), 8, W, true, "eight dup");
    bad += run_cloud<20>(synth_cloud(19, 4, 0), 20, W, true, "fewer than k");
    bad += run_cloud<20>(synth_cloud(33, 5, 1), 20, W, true, "33 dup");
    bad += run_cloud<20>(synth_cloud(1000, 6, 1), 20, W, true, "1000 dup");
    bad += run_cloud<20>(synth_cloud(1000, 6, 0), 10, W, true, "1000 k=10 in K=20");
    bad += run_cloud<32>(synth_cloud(3000, 7, 0), 32, W, true, "3000 k=32");
    bad += run_cloud<32>(synth_cloud(3000, 7, 0), 25, W, true, "3000 k=25");
    bad += run_cloud<8>(synth_cloud(3000, 8, 0), 3, W, true, "3000 k=3");
    bad += run_cloud<20>(synth_cloud(3000, 9, 2), 20, W, true, "runs of 50");
    bad += run_cloud<20>(synth_cloud(1000, 10, 3), 20, W, true, "lattice");
    bad += run_cloud<20>(synth_cloud(20000, 11, 0), 20, W, false, "20000");
  }
  printf(bad ? "KNN_BLOCK_MODEL_FAILED\n" : "KNN_BLOCK_MODEL_OK\n");
  return bad ? 1 : 0;
}
