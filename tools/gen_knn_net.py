#!/usr/bin/env python3
"""Generates locus_amd/csrc/lh_knn_net.hpp: the straight-line compare-exchange networks of the block k-NN kernel (k_knn_block).

  knn_sort8(B)            the 19-comparator network for 8 keys
  knn_merge<K>(L, B)      L (K keys ascending) := the K smallest of L u B (B: 8 keys ascending), ascending.  A bitonic merge of
                          L ++ INF... ++ reverse(B) padded to a power of two, with three prunings done HERE, symbolically:
                          a comparator against a known +INF pad is a rename, a comparator between two untouched entries of the sorted
                          L is a no-op, and a comparator whose outputs never reach positions 0..K-1 is dropped (one whose larger / smaller
                          output is dead becomes a single min / max).
  knn_sort_pairs<K>(a)    K 64-bit keys ascending: Batcher's odd-even merge sort for the next power of two with every comparator that
                          touches a pad position (>= K) dropped (pads = +INF at the top never move).

Every network is checked here before it is written: sort8 and the pair sorts by the 0-1 principle / random permutations, the merges
against sorted(L + B)[:K] on random inputs with ties.  Keys are unsigned: the float bits of non-negative squared distances.
Run: python tools/gen_knn_net.py   (rewrites the header; the header is committed)"""
import itertools
import os
import random

SORT8 = [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (1, 2), (5, 6), (0, 4), (3, 7), (1, 5), (2, 6), (1, 4), (3, 6),
         (2, 4), (3, 5), (3, 4)]


def check_sort8():
    for bits in itertools.product((0, 1), repeat=8):
        v = list(bits)
        for i, j in SORT8:
            if v[i] > v[j]:
                v[i], v[j] = v[j], v[i]
        assert v == sorted(v), bits


def merge_ops(K, NB=8):
    """-> (ops, out): ops = list of (kind, dst..., a, b) over SSA names; out[i] = name holding result position i"""
    P = 1
    while P < K + NB:
        P *= 2
    X = [("L", i) for i in range(K)] + ["INF"] * (P - K - NB) + [("B", NB - 1 - j) for j in range(NB)]
    ops = []   # (lo_name, hi_name, a, b)
    cnt = [0]

    def fresh():
        cnt[0] += 1
        return ("t", cnt[0])

    d = P // 2
    while d >= 1:
        for i in range(P):
            if i & d:
                continue
            a, b = X[i], X[i + d]
            if b == "INF":
                continue
            if a == "INF":
                X[i], X[i + d] = b, "INF"
                continue
            if a[0] == "L" and b[0] == "L" and a[1] < b[1]:
                continue
            lo, hi = fresh(), fresh()
            ops.append((lo, hi, a, b))
            X[i], X[i + d] = lo, hi
        d //= 2
    out = X[:K]
    assert all(o != "INF" for o in out)
    live = set(out)
    kept = []
    for lo, hi, a, b in reversed(ops):
        l, h = lo in live, hi in live
        if not l and not h:
            continue
        kept.append((lo if l else None, hi if h else None, a, b))
        live.add(a)
        live.add(b)
    kept.reverse()
    return kept, out


def run_merge(kept, out, L, B):
    env = {}
    for i, v in enumerate(L):
        env[("L", i)] = v
    for i, v in enumerate(B):
        env[("B", i)] = v
    for lo, hi, a, b in kept:
        if lo:
            env[lo] = min(env[a], env[b])
        if hi:
            env[hi] = max(env[a], env[b])
    return [env[o] for o in out]


def check_merge(K, kept, out):
    rng = random.Random(K)
    for trial in range(4000):
        span = rng.choice([3, 10, 1000, 1 << 30])
        L = sorted(rng.randrange(span) for _ in range(K))
        B = sorted(rng.randrange(span) for _ in range(8))
        if trial % 7 == 0:
            L = sorted(L[: rng.randrange(K + 1)] + [0x7F800000] * K)[:K]   # a list that is not full yet
        if trial % 11 == 0:
            B = sorted(B[: rng.randrange(9)] + [0x7F800000] * 8)[:8]       # masked chunk entries
        assert run_merge(kept, out, L, B) == sorted(L + B)[:K], (K, L, B)


def oddeven_merge_sort(n):
    """Batcher's odd-even merge sort, n a power of two -> comparator list"""
    ces = []
    p = 1
    while p < n:
        k = p
        while k >= 1:
            for j in range(k % p, n - k, 2 * k):
                for i in range(min(k, n - j - k)):
                    if (i + j) // (2 * p) == (i + j + k) // (2 * p):
                        ces.append((i + j, i + j + k))
            k //= 2
        p *= 2
    return ces


def pair_sort(K):
    P = 1
    while P < K:
        P *= 2
    return [(i, j) for i, j in oddeven_merge_sort(P) if j < K]


def check_pair_sort(K, ces):
    rng = random.Random(100 + K)
    for _ in range(3000):
        v = [rng.randrange(rng.choice([2, 5, 1 << 40])) for _ in range(K)]
        w = list(v)
        for i, j in ces:
            if w[i] > w[j]:
                w[i], w[j] = w[j], w[i]
        assert w == sorted(v)


def name(x):
    if x[0] == "L":
        return "L[%d]" % x[1]
    if x[0] == "B":
        return "B[%d]" % x[1]
    return "t%d" % x[1]


def main():
    check_sort8()
    lines = []
    w = lines.append
    w("// lh_knn_net.hpp -- GENERATED by tools/gen_knn_net.py (do not edit; rerun the generator): the compare-exchange networks of the")
    w("// block k-NN kernel.  Keys are the float bits of non-negative squared distances (unsigned order = float order); every network was")
    w("// checked by the generator (0-1 principle / random inputs with ties against a plain sort) before it was written.")
    w("#pragma once")
    w("#include <stdint.h>")
    w("namespace lh {")
    w("#define LH_KNN_MIN(a, b) ((a) < (b) ? (a) : (b))")
    w("#define LH_KNN_MAX(a, b) ((a) < (b) ? (b) : (a))")
    w("// 8 keys ascending, 19 comparators")
    w("__host__ __device__ __forceinline__ void knn_sort8(uint32_t* B) {")
    w("  uint32_t x, y;")
    for i, j in SORT8:
        w("  x = LH_KNN_MIN(B[%d], B[%d]); y = LH_KNN_MAX(B[%d], B[%d]); B[%d] = x; B[%d] = y;" % (i, j, i, j, i, j))
    w("}")
    w("template <int K> struct KnnNet;")
    stats = {}
    for K in (8, 20, 32):
        kept, out = merge_ops(K)
        check_merge(K, kept, out)
        n_ops = sum((1 if lo else 0) + (1 if hi else 0) for lo, hi, _, _ in kept)
        ces = pair_sort(K)
        check_pair_sort(K, ces)
        stats[K] = (n_ops, len(ces))
        w("template <> struct KnnNet<%d> {" % K)
        w("  // L (%d keys ascending) := the %d smallest of L u B (B: 8 keys ascending), ascending: %d min / max operations" % (K, K, n_ops))
        w("  static __host__ __device__ __forceinline__ void merge(uint32_t* L, const uint32_t* B) {")
        for lo, hi, a, b in kept:
            s = "   "
            if lo:
                s += " const uint32_t %s = LH_KNN_MIN(%s, %s);" % (name(lo), name(a), name(b))
            if hi:
                s += " const uint32_t %s = LH_KNN_MAX(%s, %s);" % (name(hi), name(a), name(b))
            w(s)
        for i, o in enumerate(out):
            if o != ("L", i):
                w("    L[%d] = %s;" % (i, name(o)))
        w("  }")
        w("  // %d 64-bit keys ascending: %d comparators (odd-even merge sort, pad comparators dropped)" % (K, len(ces)))
        w("  static __host__ __device__ __forceinline__ void sort_pairs(uint64_t* a) {")
        w("    uint64_t x, y;")
        for i, j in ces:
            w("    x = LH_KNN_MIN(a[%d], a[%d]); y = LH_KNN_MAX(a[%d], a[%d]); a[%d] = x; a[%d] = y;" % (i, j, i, j, i, j))
        w("  }")
        w("};")
    w("#undef LH_KNN_MIN")
    w("#undef LH_KNN_MAX")
    w("}  // namespace lh")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "locus_amd", "csrc", "lh_knn_net.hpp")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", os.path.normpath(path), {k: {"merge_ops": v[0], "pair_sort_ces": v[1]} for k, v in stats.items()})


if __name__ == "__main__":
    main()
