// Host model (no GPU) of two restructurings of the all-walk sweeps' exact 1-NN search (round 6), priced in WAVE STEPS per 64 queries
// against the product's lockstep per-lane walk (grid_start_model.cpp: 17.6 / 12.3 / 10.7 steps in sweeps 0 / 1 / 2):
//
//  (A) WAVE-SHARED WALK (the k-NN block search of lh_knn.hip applied to 1-NN of T p): 64 consecutive queries, ONE traversal; a child is
//      visited if ANY lane's ball reaches its box, a leaf's eight points are offered to ALL lanes (candidates at wave-uniform addresses).
//      Counted: node visits and leaf visits of the wave (each one a full-wave step), from the root and from the deepest node that
//      covers every lane's start cell.
//  (B) ITEM STACKS: the unit of work is (query, subtree) instead of a lane's whole walk.  A wave owns Q consecutive queries and two
//      LIFO stacks in LDS -- node items and leaf items -- a step pops up to 64 items of ONE kind, lane l works on item l: a node item
//      tests the four child boxes against the query's CURRENT bound (a per-query 64-bit (d2, id) slot updated with an atomic minimum)
//      and pushes the survivors, a leaf item scans its eight points.  Exactness does not depend on the order: a subtree is dropped
//      only if its box is farther than a real candidate, the minimum over (d2 << 32 | id) is the nearest point with the lowest index.
//      Counted: steps per 64 queries, lanes busy, items, the stack's high-water mark.
//   hipcc -O2 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -x hip tools/model/item_stack_model.cpp -o /tmp/ism && /tmp/ism /tmp/wm 256
#define TRAVERSAL_CHECK_NO_MAIN
#include "../../tests/host_emu/traversal_check.cpp"
#include <cstring>
#include <string>
#include <algorithm>

static std::vector<float> read_f32(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { printf("cannot open %s\n", path.c_str()); exit(1); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<float> v(sz / 4);
  if (fread(v.data(), 4, v.size(), f) != v.size()) exit(1);
  fclose(f);
  return v;
}

struct Item { int q; int32_t ref; float key; };
struct Query { float x, y, z; GridQuery gq; float bd; int bi; };

static inline void offer(Query& q, float d, int id) {
  if (d < q.bd || (d == q.bd && id < q.bi)) { q.bd = d; q.bi = id; }
}

int main(int argc, char** argv) {
  std::string dir = argc > 1 ? argv[1] : "/tmp/wm";
  const int Q = argc > 2 ? atoi(argv[2]) : 256;        // queries per wave (B)
  const int policy = argc > 3 ? atoi(argv[3]) : 0;     // (B) 0: leaves first when >= 64 or no nodes; 1: larger stack; 2: one mixed stack (a step with both kinds costs two)
  const int seed_group = 16;
  auto tg = read_f32(dir + "/tgt.f32"), sr = read_f32(dir + "/src.f32"), po = read_f32(dir + "/poses.f32");
  int m = (int)tg.size() / 3, n = (int)sr.size() / 3, np = (int)po.size() / 12;
  std::vector<float4> tp(m);
  for (int i = 0; i < m; i++) tp[i] = make_float4(tg[3 * i], tg[3 * i + 1], tg[3 * i + 2], 1.f);
  HostTree t = build(tp);
  TreeView tv = t.view();
  const TreeHeader h = t.hdr();
  std::vector<int> prev(n, -1);
  for (int s = 0; s < np && s < 3; s++) {
    const float* T = &po[12 * s];
    std::vector<Query> qs(n);
    std::vector<int> truth(n);
    for (int i = 0; i < n; i++) {
      Query& q = qs[i];
      xform_pt(T, sr[3 * i], sr[3 * i + 1], sr[3 * i + 2], q.x, q.y, q.z);
      q.gq = grid_query(h, q.x, q.y, q.z);
      int cand = prev[i];
      if (s == 0) {
        int g0 = i - i % seed_group;
        if (i == g0) {
          Nn1Collector cd{inf_f(), 0x7fffffff};
          tree_descend<Nn1Collector, true>(tv, q.x, q.y, q.z, cd);
          prev[g0] = cd.bi;
        }
        cand = prev[g0];
        int alt = std::min(m - 1, cand + (i - g0));
        if (d2f(q.x, q.y, q.z, tp[alt].x, tp[alt].y, tp[alt].z) < d2f(q.x, q.y, q.z, tp[cand].x, tp[cand].y, tp[cand].z)) cand = alt;
      }
      q.bd = d2f(q.x, q.y, q.z, tp[cand].x, tp[cand].y, tp[cand].z);
      q.bi = cand;
      // the truth: the product's walk
      Nn1CertCollector c{q.bd, q.bi, inf_f()};
      std::vector<uint64_t> st(LDS_STACK);
      WalkStack<LDS_STACK> ws(st.data(), 1);
      int32_t ref = h.root;
      const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), q.x, q.y, q.z, c, [&](uint32_t key, int32_t r) { ws.push(key, r); });
      if (g != GRID_USE_ROOT) ref = (g == GRID_EMPTY) ? ws.pop(c) : g;
      for (;;) {
        while (ref >= 0 && ref != NO_CHILD) ref = node_visit(tv.nodes[ref], q.gq, h.scl2, c, ws);
        if (ref == NO_CHILD) break;
        scan_leaf(tv, ref, q.x, q.y, q.z, c);
        ref = ws.pop(c);
      }
      truth[i] = c.bi;
    }
    const double waves = n / 64.0;

    // ---- (A) wave-shared walk from the root -------------------------------------------------------------------------------
    {
      long nodes = 0, leaves = 0, mism = 0, top = 0;
      for (int b0 = 0; b0 < n; b0 += 64) {
        const int cnt = std::min(64, n - b0);
        std::vector<Query> L(qs.begin() + b0, qs.begin() + b0 + cnt);
        std::vector<int32_t> stack{h.root};
        // the part of the path above the deepest node that still holds every lane's ball is the same for the whole wave and could be
        // skipped with a start table: counted separately (`top`)
        bool branched = false;
        while (!stack.empty()) {
          int32_t ref = stack.back(); stack.pop_back();
          if (ref < 0) {
            const uint32_t u = (uint32_t)~ref;
            const float4* p = tv.pts + (u >> 4);
            // visited only if some lane still wants it (the bound may have tightened since the push)
            // (the product would re-test with the leaf's box; the model keeps the parent's vote)
            leaves++;
            for (auto& q : L)
              for (int e = 0; e < LEAF_CAP; e++) offer(q, d2f(q.x, q.y, q.z, p[e].x, p[e].y, p[e].z), (int)f2u(p[e].w));
            continue;
          }
          const NodeX& nd = tv.nodes[ref];
          nodes++;
          int votes[4] = {0, 0, 0, 0};
          for (auto& q : L)
            for (int k = 0; k < 4; k++)
              if (nd.child[k] != NO_CHILD && boxd2_q(q.gq, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k], h.scl2) <= q.bd) votes[k]++;
          int order[4] = {0, 1, 2, 3};
          std::sort(order, order + 4, [&](int a, int b) { return votes[a] < votes[b]; });   // most wanted child on top of the stack
          int nv = 0;
          for (int k = 0; k < 4; k++)
            if (votes[order[k]] > 0) { stack.push_back(nd.child[order[k]]); nv++; }
          if (!branched) { if (nv > 1) branched = true; else top++; }
        }
        for (int k = 0; k < cnt; k++) if (L[k].bi != truth[b0 + k]) mism++;
      }
      printf("sweep %d (A) wave-shared walk: %.1f node + %.1f leaf visits per 64 queries (%.1f of the node visits on the common path above the first branch), mismatches %ld\n",
             s, nodes / waves, leaves / waves, top / waves, mism);
    }

    // ---- (B) item stacks -------------------------------------------------------------------------------------------------
    {
      long steps_n = 0, steps_l = 0, busy = 0, items_n = 0, items_l = 0, dead = 0, mism = 0, hw = 0, init_items = 0;
      for (int b0 = 0; b0 < n; b0 += Q) {
        const int cnt = std::min(Q, n - b0);
        std::vector<Query> L(qs.begin() + b0, qs.begin() + b0 + cnt);
        std::vector<Item> SN, SL;   // node items, leaf items (policy 2: everything in SN)
        auto put = [&](int q, int32_t ref, float key) {
          if (policy == 2 || ref >= 0) SN.push_back(Item{q, ref, key}); else SL.push_back(Item{q, ref, key});
        };
        for (int k = 0; k < cnt; k++) {
          Query& q = L[k];
          Nn1CertCollector c{q.bd, q.bi, inf_f()};
          int32_t ref = h.root;
          const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), q.x, q.y, q.z, c, [&](uint32_t key, int32_t r) { put(k, r, u2f(key)); init_items++; });
          if (g != GRID_USE_ROOT) ref = g;
          if (ref != GRID_EMPTY) { put(k, ref, 0.0f); init_items++; }
        }
        for (;;) {
          if (SN.empty() && SL.empty()) break;
          hw = std::max<long>(hw, (long)(SN.size() + SL.size()));
          bool leaf_step;
          if (policy == 0) leaf_step = SL.size() >= 64 || SN.empty();
          else leaf_step = SL.size() >= SN.size();
          if (policy == 2) leaf_step = false;
          std::vector<Item>& S = leaf_step ? SL : SN;
          const int take = (int)std::min<size_t>(64, S.size());
          std::vector<Item> cur(S.end() - take, S.end());
          S.resize(S.size() - take);
          bool has_n = false, has_l = false;
          // all lanes read the bounds of the moment, the updates land afterwards (atomic minimum)
          std::vector<std::pair<int, std::pair<float, int>>> upd;
          std::vector<Item> pushes;
          for (auto& it : cur) {
            Query& q = L[it.q];
            if (it.key > q.bd) { dead++; continue; }   // the bound has tightened since the push: an idle lane in this step
            busy++;
            if (it.ref < 0) {
              has_l = true; items_l++;
              const uint32_t u = (uint32_t)~it.ref;
              const float4* p = tv.pts + (u >> 4);
              float bd = q.bd; int bi = q.bi;
              for (int e = 0; e < LEAF_CAP; e++) {
                const float d = d2f(q.x, q.y, q.z, p[e].x, p[e].y, p[e].z); const int id = (int)f2u(p[e].w);
                if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
              }
              upd.push_back({it.q, {bd, bi}});
            } else {
              has_n = true; items_n++;
              const NodeX& nd = tv.nodes[it.ref];
              for (int k = 0; k < 4; k++) {
                if (nd.child[k] == NO_CHILD) continue;
                const float dk = boxd2_q(q.gq, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k], h.scl2);
                if (dk <= q.bd) pushes.push_back(Item{it.q, nd.child[k], dk});
              }
            }
          }
          for (auto& u : upd) offer(L[u.first], u.second.first, u.second.second);
          for (auto& p : pushes) put(p.q, p.ref, p.key);
          if (policy == 2) { steps_n += has_n; steps_l += has_l; if (!has_n && !has_l) steps_n++; }
          else (leaf_step ? steps_l : steps_n)++;
        }
        for (int k = 0; k < cnt; k++) if (L[k].bi != truth[b0 + k]) mism++;
      }
      printf("sweep %d (B) item stacks, %d queries per wave, policy %d: %.2f node + %.2f leaf steps per 64 queries = %.2f (%.1f lanes busy), items per query %.2f node + %.2f leaf (+ %.2f dropped at the pop), "
             "%.2f start items per query, stack high-water %ld, mismatches %ld\n",
             s, Q, policy, steps_n / waves, steps_l / waves, (steps_n + steps_l) / waves, (double)busy / (steps_n + steps_l), (double)items_n / n, (double)items_l / n,
             (double)dead / n, (double)init_items / n, hw, mism);
    }

    // ---- (C) per-lane depth-first walks over a SHARED stack ----------------------------------------------------------------
    // A lane keeps descending into the nearest surviving child like the product's walk (the item stays in its registers), the
    // other survivors go on the WAVE's stack (farthest first); a lane whose walk has nothing left takes the next FRESH query
    // (its start cell, from the start phase) and, when there is none (or the stack is more than `drain` full), the top of the
    // stack -- whoever's query that belongs to.  The wave takes a node step or a leaf step by majority of the held items.
    for (int two = 0; two < 2; two++) {
      long steps_n = 0, steps_l = 0, busy = 0, items_n = 0, items_l = 0, dead = 0, mism = 0, hw = 0, pops = 0, fresh_taken = 0, pop_rounds = 0;
      const size_t drain = 384;
      const int max_pop_rounds = argc > 4 ? atoi(argv[4]) : 4;
      for (int b0 = 0; b0 < n; b0 += Q) {
        const int cnt = std::min(Q, n - b0);
        std::vector<Query> L(qs.begin() + b0, qs.begin() + b0 + cnt);
        std::vector<Item> SN, SL;            // deferred items by kind (two == 0: all in SN)
        std::vector<int32_t> home(cnt);
        auto put = [&](int q, int32_t ref, float key) {
          if (!two || ref >= 0) SN.push_back(Item{q, ref, key}); else SL.push_back(Item{q, ref, key});
        };
        for (int k = 0; k < cnt; k++) {
          Query& q = L[k];
          Nn1CertCollector c{q.bd, q.bi, inf_f()};
          int32_t ref = h.root;
          const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), q.x, q.y, q.z, c, [&](uint32_t key, int32_t r) { put(k, r, u2f(key)); });
          if (g != GRID_USE_ROOT) ref = g;
          home[k] = ref;   // GRID_EMPTY: nothing in the query's own cell
        }
        int next_fresh = 0;
        struct Lane { int q = -1; int32_t ref = NO_CHILD; };
        std::vector<Lane> W(64);
        for (;;) {
          // refill: idle lanes take fresh queries, then stack items
          hw = std::max<long>(hw, (long)(SN.size() + SL.size()));
          int nn = 0, nl = 0;
          for (auto& l : W) { if (l.ref == NO_CHILD) continue; (l.ref >= 0 ? nn : nl)++; }
          const bool drain_now = SN.size() + SL.size() > drain;
          for (int round = 0; round < max_pop_rounds; round++) {
            bool any_dead = false, any_pop = false;
            for (auto& l : W) {
              if (l.ref != NO_CHILD) continue;
              if (!drain_now && round == 0) {
                while (next_fresh < cnt && home[next_fresh] == GRID_EMPTY) next_fresh++;
                if (next_fresh < cnt) { l.q = next_fresh; l.ref = home[next_fresh]; next_fresh++; fresh_taken++; (l.ref >= 0 ? nn : nl)++; continue; }
              }
              // a stack item: of the kind the wave is about to step if there is a choice
              std::vector<Item>* S = &SN;
              if (two) {
                const bool want_node = nn >= nl;
                S = want_node ? (SN.empty() ? &SL : &SN) : (SL.empty() ? &SN : &SL);
              }
              if (S->empty()) continue;
              Item it = S->back(); S->pop_back(); pops++; any_pop = true;
              if (it.key > L[it.q].bd) { dead++; any_dead = true; continue; }   // dropped at the pop
              l.q = it.q; l.ref = it.ref; (l.ref >= 0 ? nn : nl)++;
            }
            if (any_pop) pop_rounds++;
            if (!any_dead) break;
          }
          if (nn == 0 && nl == 0) {
            if (next_fresh >= cnt && SN.empty() && SL.empty()) break;
            continue;   // (everything popped was dead: another refill round)
          }
          const bool node_step = nn >= nl;
          (node_step ? steps_n : steps_l)++;
          std::vector<std::pair<int, std::pair<float, int>>> upd;
          std::vector<Item> pushes[3];
          for (auto& l : W) {
            if (l.ref == NO_CHILD) continue;
            Query& q = L[l.q];
            if (node_step && l.ref >= 0) {
              busy++; items_n++;
              const NodeX& nd = tv.nodes[l.ref];
              std::pair<float, int32_t> ch[4]; int nc = 0;
              for (int k = 0; k < 4; k++) {
                if (nd.child[k] == NO_CHILD) continue;
                const float dk = boxd2_q(q.gq, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k], h.scl2);
                if (dk <= q.bd) ch[nc++] = {dk, nd.child[k]};
              }
              std::sort(ch, ch + nc, [](auto& a, auto& b) { return a.first < b.first; });
              for (int k = 1; k < nc; k++) pushes[k - 1].push_back(Item{l.q, ch[k].second, ch[k].first});
              if (nc) l.ref = ch[0].second; else l.ref = NO_CHILD;
            } else if (!node_step && l.ref < 0) {
              busy++; items_l++;
              const uint32_t u = (uint32_t)~l.ref;
              const float4* p = tv.pts + (u >> 4);
              float bd = q.bd; int bi = q.bi;
              for (int e = 0; e < LEAF_CAP; e++) {
                const float d = d2f(q.x, q.y, q.z, p[e].x, p[e].y, p[e].z); const int id = (int)f2u(p[e].w);
                if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
              }
              upd.push_back({l.q, {bd, bi}});
              l.ref = NO_CHILD;
            }
          }
          for (auto& u : upd) offer(L[u.first], u.second.first, u.second.second);
          for (int r = 2; r >= 0; r--) for (auto& p : pushes[r]) put(p.q, p.ref, p.key);   // farthest first: the nearer ones end on top
        }
        for (int k = 0; k < cnt; k++) if (L[k].bi != truth[b0 + k]) mism++;
      }
      printf("sweep %d (C) shared stack (%s), %d queries per wave: %.2f node + %.2f leaf steps per 64 queries = %.2f (%.1f lanes busy), items per query %.2f node + %.2f leaf, "
             "%.2f pops per query of which %.2f dead, %.2f pop rounds per 64 queries, stack high-water %ld, mismatches %ld\n",
             s, two ? "two kinds" : "one", Q, steps_n / waves, steps_l / waves, (steps_n + steps_l) / waves, (double)busy / (steps_n + steps_l), (double)items_n / n, (double)items_l / n,
             (double)pops / n, (double)dead / n, hw, mism);
    }

    // ---- (D) first descent in lockstep, the rest as items --------------------------------------------------------------
    // Batches of 64 fresh queries walk their FIRST path (start cell -> nearest leaf) in lockstep like the product, every lane on its own
    // query, the surviving siblings go to the wave's shared stacks (by kind); after the batch's leaf scan the lanes do not pop their own
    // stacks: once all Q queries of the wave have made their first descent, the stacked items are worked off 64 at a time in steps of ONE
    // kind, whoever's query they belong to, dead ones dropped in a pop loop that costs a fraction of a step.
    for (int cold_noseed = 0; cold_noseed < (s == 0 ? 2 : 1); cold_noseed++) {
      long d_node = 0, d_leaf = 0, p_node = 0, p_leaf = 0, busy1 = 0, busy2 = 0, items2 = 0, dead = 0, mism = 0, hw = 0, pop_rounds = 0;
      for (int b0 = 0; b0 < n; b0 += Q) {
        const int cnt = std::min(Q, n - b0);
        std::vector<Query> L(qs.begin() + b0, qs.begin() + b0 + cnt);
        if (cold_noseed) for (auto& q : L) { q.bd = inf_f(); q.bi = 0x7fffffff; }
        std::vector<Item> SN, SL;
        auto put = [&](int q, int32_t ref, float key) { (ref >= 0 ? SN : SL).push_back(Item{q, ref, key}); };
        for (int w0 = 0; w0 < cnt; w0 += 64) {
          const int wc = std::min(64, cnt - w0);
          std::vector<int32_t> ref(wc);
          for (int k = 0; k < wc; k++) {
            Query& q = L[w0 + k];
            int32_t r = h.root;
            if (q.bd < inf_f()) {
              Nn1CertCollector c{q.bd, q.bi, inf_f()};
              const int32_t g = grid_start(h.org, h.key_sc, h.key_inv, tv.grid(), q.x, q.y, q.z, c, [&](uint32_t key, int32_t rr) { put(w0 + k, rr, u2f(key)); });
              if (g != GRID_USE_ROOT) r = g;
            } else {   // cold without a seed: the query's own level-5 cell if it holds anything (tree_descend<.., true>), else the root; the rest of the
                       // cloud is owed a visit: the root goes on the stack with key 0 (popped dead once the bound is tight? no: it must be walked) -- so
                       // the model walks from the ROOT for these
              r = h.root;
            }
            ref[k] = r;   // GRID_EMPTY: nothing in the own cell -- the lane idles through phase 1
          }
          for (;;) {   // node steps until every lane holds a leaf (or nothing)
            int act = 0;
            for (int k = 0; k < wc; k++) if (ref[k] >= 0 && ref[k] != NO_CHILD) act++;
            if (!act) break;
            d_node++; busy1 += act;
            for (int k = 0; k < wc; k++) {
              if (!(ref[k] >= 0 && ref[k] != NO_CHILD)) continue;
              Query& q = L[w0 + k];
              const NodeX& nd = tv.nodes[ref[k]];
              std::pair<float, int32_t> ch[4]; int nc = 0;
              for (int c = 0; c < 4; c++) {
                if (nd.child[c] == NO_CHILD) continue;
                const float dk = boxd2_q(q.gq, nd.lo_xy[c], nd.hi_xy[c], nd.z_lohi[c], h.scl2);
                if (dk <= q.bd) ch[nc++] = {dk, nd.child[c]};
              }
              std::sort(ch, ch + nc, [](auto& a, auto& b) { return a.first < b.first; });
              for (int c = nc - 1; c >= 1; c--) put(w0 + k, ch[c].second, ch[c].first);
              ref[k] = nc ? ch[0].second : NO_CHILD;
            }
          }
          int act = 0;
          for (int k = 0; k < wc; k++) if (ref[k] < 0 && ref[k] != GRID_EMPTY) act++;
          if (act) {
            d_leaf++; busy1 += act;
            for (int k = 0; k < wc; k++) {
              if (!(ref[k] < 0 && ref[k] != GRID_EMPTY)) continue;
              Query& q = L[w0 + k];
              const uint32_t u = (uint32_t)~ref[k];
              const float4* p = tv.pts + (u >> 4);
              for (int e = 0; e < LEAF_CAP; e++) offer(q, d2f(q.x, q.y, q.z, p[e].x, p[e].y, p[e].z), (int)f2u(p[e].w));
            }
          }
          hw = std::max<long>(hw, (long)(SN.size() + SL.size()));
        }
        // phase 2: the stacked items, one kind per step; a lane pops until it holds a live item (each round of pops is counted)
        for (;;) {
          if (SN.empty() && SL.empty()) break;
          const bool leaf_step = SL.size() >= 64 || SN.empty() || (SL.size() >= SN.size() && SN.size() < 64);
          std::vector<Item>& S = leaf_step ? SL : SN;
          std::vector<Item> cur;
          while ((int)cur.size() < 64 && !S.empty()) {
            pop_rounds++;
            const int want = 64 - (int)cur.size();
            const int take = (int)std::min<size_t>(want, S.size());
            for (int k = 0; k < take; k++) {
              Item it = S.back(); S.pop_back();
              if (it.key > L[it.q].bd) { dead++; continue; }
              cur.push_back(it);
            }
          }
          if (cur.empty()) continue;
          (leaf_step ? p_leaf : p_node)++;
          busy2 += (long)cur.size(); items2 += (long)cur.size();
          std::vector<std::pair<int, std::pair<float, int>>> upd;
          std::vector<Item> pushes;
          for (auto& it : cur) {
            Query& q = L[it.q];
            if (it.ref < 0) {
              const uint32_t u = (uint32_t)~it.ref;
              const float4* p = tv.pts + (u >> 4);
              float bd = q.bd; int bi = q.bi;
              for (int e = 0; e < LEAF_CAP; e++) {
                const float d = d2f(q.x, q.y, q.z, p[e].x, p[e].y, p[e].z); const int id = (int)f2u(p[e].w);
                if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
              }
              upd.push_back({it.q, {bd, bi}});
            } else {
              const NodeX& nd = tv.nodes[it.ref];
              for (int k = 0; k < 4; k++) {
                if (nd.child[k] == NO_CHILD) continue;
                const float dk = boxd2_q(q.gq, nd.lo_xy[k], nd.hi_xy[k], nd.z_lohi[k], h.scl2);
                if (dk <= q.bd) pushes.push_back(Item{it.q, nd.child[k], dk});
              }
            }
          }
          for (auto& u : upd) offer(L[u.first], u.second.first, u.second.second);
          for (auto& p : pushes) put(p.q, p.ref, p.key);
          hw = std::max<long>(hw, (long)(SN.size() + SL.size()));
        }
        for (int k = 0; k < cnt; k++) if (L[k].bi != truth[b0 + k]) mism++;
      }
      printf("sweep %d (D) lockstep first descent + items%s, %d queries per wave: phase 1 %.2f node + %.2f leaf steps per 64 queries (%.1f lanes busy), phase 2 %.2f node + %.2f leaf (%.1f lanes busy, %.2f live items "
             "per query, %.2f dead, %.2f pop rounds per 64 queries) = %.2f steps, stack high-water %ld, mismatches %ld\n",
             s, cold_noseed ? " (cold, NO seeds: from the root)" : "", Q, d_node / waves, d_leaf / waves, (double)busy1 / std::max(1L, d_node + d_leaf), p_node / waves, p_leaf / waves,
             (double)busy2 / std::max(1L, p_node + p_leaf), (double)items2 / n, (double)dead / n, pop_rounds / waves, (d_node + d_leaf + p_node + p_leaf) / waves, hw, mism);
    }
    for (int i = 0; i < n; i++) prev[i] = truth[i];
  }
  return 0;
}
