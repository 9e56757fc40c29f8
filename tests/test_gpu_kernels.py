"""GPU suite (-m gpu): kernel-level parity of the HIP path against the CPU oracle, through the C ABI.
Bit-exact for indices / integer work; stated tolerances for floating point."""
import numpy as np
import pytest

from locus_amd import synth

pytestmark = pytest.mark.gpu


def _cloud_pts(seed=0, n=20000, dup=True):
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.normal(size=(n, 3)) * [8, 5, 1.5], rng.uniform(-20, 20, size=(n // 4, 3))]).astype(np.float32)
    if dup:
        pts = np.concatenate([pts, pts[: n // 10]])  # exact duplicates exercise the tie rule (lowest index)
    return pts


@pytest.mark.parametrize("n", [1, 7, 8, 9, 33, 1000, 50000])
def test_nn1_bit_exact(ctx, capi, oracle, n):
    pts = _cloud_pts(1, max(n, 8), dup=n > 100)[:n]
    rng = np.random.default_rng(2)
    q = np.concatenate([rng.normal(size=(3000, 3)) * [9, 6, 2], pts[: min(n, 500)]]).astype(np.float32)
    tgt = capi.Cloud(ctx, pts)
    qc = capi.Cloud(ctx, q)
    idx, d2 = tgt.nn1(qc)
    io, do = oracle.Tree(oracle.xyz4(pts)).nn1(oracle.xyz4(q), threads=4)
    assert (idx == io).all()
    assert (d2 == do).all()  # float distances bit-identical (same operation order, no FMA)


def test_index_heavy_duplicates_and_degenerate_clouds(ctx, capi, oracle):
    # runs of 50 identical points (more than a leaf holds: the finest key cell is cut into fixed chunks and the radix tree
    # breaks ties by leaf index), a cloud on a line, a cloud of one repeated point, and far outliers stretching the key grid
    rng = np.random.default_rng(7)
    base = (rng.normal(size=(200, 3)) * [5, 3, 1]).astype(np.float32)
    clouds = {
        "dup50": np.repeat(base, 50, axis=0),
        "line": np.stack([np.linspace(-30, 30, 5000), np.zeros(5000), np.zeros(5000)], 1).astype(np.float32),
        "one_point": np.repeat(np.array([[1.5, -2.0, 0.25]], np.float32), 300, axis=0),
        "outliers": np.concatenate([base, np.array([[4000.0, 0, 0], [0, -4000.0, 10.0]], np.float32)]),
    }
    q = np.concatenate([rng.normal(size=(2000, 3)) * [6, 4, 2], base[:100]]).astype(np.float32)
    qc = capi.Cloud(ctx, q)
    for name, pts in clouds.items():
        tgt = capi.Cloud(ctx, pts)
        idx, d2 = tgt.nn1(qc)
        io, do = oracle.nn1_brute(oracle.xyz4(pts), oracle.xyz4(q))
        assert (idx == io).all() and (d2 == do).all(), name
        k = 20
        ki, kd = tgt.knn(qc, k)
        ko, kdo = oracle.knn_brute(oracle.xyz4(pts), oracle.xyz4(q), k)
        assert (ki == ko).all() and (kd == kdo).all(), name


def test_batched_index_build_mixed_sizes(ctx, capi, oracle):
    # one batched build (one key sort, one leaf scan, one radix tree over all clouds' leaves) for clouds of very different
    # sizes, including single-leaf clouds between big ones: every cloud must get exactly its own tree
    rng = np.random.default_rng(8)
    sizes = [30000, 3, 8, 9, 12000, 1, 700, 64]
    tg = [(_cloud_pts(20 + i, max(n, 8), dup=n > 100)[:n] + np.float32(3.0 * i)) for i, n in enumerate(sizes)]
    src = [(t[: min(len(t), 400)] + rng.normal(scale=0.05, size=(min(len(t), 400), 3))).astype(np.float32) for t in tg]
    S = [capi.Cloud(ctx, capi.make_pointf(s, np.tile([0, 0, 1.0], (len(s), 1)).astype(np.float32))) for s in src]
    T = [capi.Cloud(ctx, capi.make_pointf(t, np.tile([0, 0, 1.0], (len(t), 1)).astype(np.float32))) for t in tg]
    P = capi.default_params(max_iterations=2, corr_dist=5.0)
    capi.align_batch(ctx, P, S, T, max_in_flight=8)   # builds the 8 target indexes in one batch
    for t_cloud, t_pts, s_pts in zip(T, tg, src):
        idx, d2 = t_cloud.nn1(capi.Cloud(ctx, s_pts))  # reuses the batch-built index
        io, do = oracle.nn1_brute(oracle.xyz4(t_pts), oracle.xyz4(s_pts))
        assert (idx == io).all() and (d2 == do).all(), len(t_pts)


@pytest.mark.parametrize("k", [1, 5, 20])
def test_knn_bit_exact(ctx, capi, oracle, k):
    pts = _cloud_pts(3, 8000)
    tgt = capi.Cloud(ctx, pts)
    idx, d2 = tgt.knn(tgt, k)
    io, do = oracle.Tree(oracle.xyz4(pts)).knn(oracle.xyz4(pts), k, threads=4)
    assert (idx == io).all() and (d2 == do).all()


@pytest.mark.parametrize("k", [3, 8, 10, 20, 25, 32, 40])
def test_block_knn_bit_exact_every_list_size(ctx, capi, oracle, k):
    """a cloud against itself goes through the block search (lh_knn_block.hpp) for k <= 32 -- lists of 8 / 20 / 32 keys, k below the
    list size (phantom keys), duplicates (ties at the k-th distance: the redo list) -- and through the one-query-per-lane kernel above"""
    pts = _cloud_pts(30 + k, 6000)
    tgt = capi.Cloud(ctx, pts)
    idx, d2 = tgt.knn(tgt, k)
    io, do = oracle.Tree(oracle.xyz4(pts)).knn(oracle.xyz4(pts), k, threads=4)
    assert (idx == io).all() and (d2 == do).all()


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 19, 20, 21, 63, 64, 65, 127, 129, 1000])
def test_block_knn_small_and_ragged_clouds(ctx, capi, oracle, n):
    pts = _cloud_pts(5, 1000, dup=False)[:n]
    tgt = capi.Cloud(ctx, pts)
    k = 20
    idx, d2 = tgt.knn(tgt, k)
    io, do = oracle.Tree(oracle.xyz4(pts)).knn(oracle.xyz4(pts), k)
    m = min(n, k)
    assert (idx[:, :m] == io[:, :m]).all() and (d2[:, :m] == do[:, :m]).all()
    assert (idx[:, m:] == -1).all() and np.isinf(d2[:, m:]).all()


def test_block_knn_ties_everywhere(ctx, capi, oracle):
    """a lattice, runs of identical points and a lidar sweep with every 5th point repeated: the k-th distance is shared by several
    points, the block search hands those queries to the redo list, and the lowest index still wins"""
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.1)
    base = (rng.normal(size=(100, 3)) * [5, 3, 1]).astype(np.float32)
    scan = synth.scan(rings=16, azimuths=300, scale=1.0, seed=8)
    clouds = {"lattice": g, "dup50": np.repeat(base, 50, axis=0), "scan_dup": np.concatenate([scan, scan[::5]])}
    for name, pts in clouds.items():
        tgt = capi.Cloud(ctx, pts)
        idx, d2 = tgt.knn(tgt, 20)
        io, do = oracle.Tree(oracle.xyz4(pts)).knn(oracle.xyz4(pts), 20, threads=4)
        assert (idx == io).all() and (d2 == do).all(), name


def test_normals_and_covariances_batch_equals_one_by_one(ctx, capi, oracle):
    """lh_normals_knn_batch / lh_cov_knn_batch: ragged clouds (one without an index, one with), one launch -- bit-identical to the
    per-cloud calls, and the normals against the oracle"""
    sizes = [(16, 300, 1), (16, 431, 2), (8, 50, 3), (32, 500, 4), (4, 5, 5), (16, 300, 6), (16, 257, 7), (16, 300, 8), (24, 300, 9)]
    pts = [synth.scan(rings=r, azimuths=a, scale=1.0, seed=s) for r, a, s in sizes]
    A = [capi.Cloud(ctx, p) for p in pts]
    B = [capi.Cloud(ctx, p) for p in pts]
    A[3].build_index()
    capi.normals_knn_batch(A, 20)
    capi.cov_knn_batch(A, 20, 1e-3)
    for a, b, p in zip(A, B, pts):
        b.normals_knn(20)
        da, db = a.download(), b.download()
        for f in ("normal_x", "normal_y", "normal_z", "curvature"):
            assert (da[f].view(np.uint32) == db[f].view(np.uint32)).all(), (len(p), f)
        assert (a.cov_knn(20, 1e-3) == b.cov_knn(20, 1e-3)).all()
    p = pts[3]
    ref = oracle.normals_knn(oracle.xyz4(p), 20, threads=4)
    d = A[3].download()
    out = np.stack([d["normal_x"], d["normal_y"], d["normal_z"]], 1)
    assert np.quantile(np.abs((out * ref[:, :3]).sum(1)), 0.01) > 1 - 1e-4


def _reachable_tree(srt, nodes, hdr):
    """the index as a comparable value: header fields, sorted points, and every node reachable from the root (unreached slots hold leftovers)"""
    root = int(hdr.view(np.int32)[0])
    out = {"hdr": hdr[:11].tobytes(), "sorted": srt.tobytes(), "nodes": {}}
    stack = [root] if root >= 0 else []
    while stack:
        i = stack.pop()
        nd = nodes[i]
        out["nodes"][i] = nd.tobytes()
        for c in nd[12:16].view(np.int32):
            if 0 <= c < 0x7fffffff:
                stack.append(int(c))
    return out


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 33, 64, 100, 1000, 2933, 4095, 4096])
def test_small_cloud_index_is_the_general_build_byte_for_byte(ctx, capi, oracle, n):
    """clouds of <= 4 096 points (LOCUS's operating point: ~3 000) are indexed by ONE launch of one workgroup (lh_index_small.hip): the same
    tree as the 13-launch general build -- header, sorted array and every reachable 4-ary node compared as bytes -- on a lidar sweep, on a cloud
    with duplicates and runs of identical points, and the neighbours it gives against the oracle"""
    rng = np.random.default_rng(n)
    base = _cloud_pts(40 + n, max(n, 8), dup=True)
    clouds = [base[:n], np.repeat((rng.normal(size=(max(1, n // 12), 3)) * [5, 3, 1]).astype(np.float32), 12, axis=0)[:n],
              synth.scan(rings=16, azimuths=300, scale=1.0, seed=n)[:n]]
    for pts in clouds:
        if len(pts) < n:
            pts = np.concatenate([pts, base[: n - len(pts)]])
        prev = capi.small_index(1)
        try:
            a = capi.Cloud(ctx, pts)
            ta = _reachable_tree(*a.index_dump())
            capi.small_index(0)
            b = capi.Cloud(ctx, pts)
            tb = _reachable_tree(*b.index_dump())
        finally:
            capi.small_index(prev)
        assert ta["hdr"] == tb["hdr"] and ta["sorted"] == tb["sorted"]
        assert ta["nodes"].keys() == tb["nodes"].keys() and all(ta["nodes"][k] == tb["nodes"][k] for k in ta["nodes"])
        q = (pts[: min(n, 200)] + rng.normal(scale=0.05, size=(min(n, 200), 3))).astype(np.float32)
        idx, d2 = a.nn1(capi.Cloud(ctx, q))
        io, do = oracle.nn1_brute(oracle.xyz4(pts), oracle.xyz4(q))
        assert (idx == io).all() and (d2 == do).all()


def test_small_and_large_clouds_in_one_batched_build(ctx, capi, oracle):
    """a batch that mixes both kinds: the large clouds go through the batched launches, the small ones through the one-workgroup build, on
    disjoint slices of the same scratch"""
    sizes = [30000, 300, 9000, 8, 4096, 4097, 1, 2500]
    pts = [_cloud_pts(60 + i, max(n, 8), dup=n > 100)[:n] + np.float32(2.0 * i) for i, n in enumerate(sizes)]
    C_ = [capi.Cloud(ctx, p) for p in pts]
    capi.normals_knn_batch([c for c, p in zip(C_, pts) if len(p) >= 3], 5)   # ONE build for all of them
    rng = np.random.default_rng(0)
    for c, p in zip(C_, pts):
        q = (p[: min(len(p), 300)] + rng.normal(scale=0.05, size=(min(len(p), 300), 3))).astype(np.float32)
        idx, d2 = c.nn1(capi.Cloud(ctx, q))
        io, do = oracle.nn1_brute(oracle.xyz4(p), oracle.xyz4(q))
        assert (idx == io).all() and (d2 == do).all(), len(p)


def test_knn_more_neighbours_than_points(ctx, capi, oracle):
    pts = _cloud_pts(4, 8, dup=False)[:5]
    tgt = capi.Cloud(ctx, pts)
    idx, d2 = tgt.knn(tgt, 8)
    io, do = oracle.Tree(oracle.xyz4(pts)).knn(oracle.xyz4(pts), 8)
    assert (idx == io).all() and (d2[:, :5] == do[:, :5]).all() and np.isinf(d2[:, 5:]).all()


def test_transform_bit_exact(ctx, capi, oracle):
    pts, nrm = synth.scan(rings=16, azimuths=300, scale=1.0, seed=5, with_normals=True)
    T = synth.pose_matrix(0.3, -0.2, 0.1, 0.01, -0.02, 0.3).astype(np.float32)
    c = capi.Cloud(ctx, capi.make_pointf(pts, nrm))
    out = c.transform(oracle.mat_to_T(T), with_normals=True).download()
    po, no = oracle.transform(oracle.xyz4(pts), oracle.mat_to_T(T), oracle.nrm4(nrm))
    got = np.stack([out["x"], out["y"], out["z"]], 1)
    gotn = np.stack([out["normal_x"], out["normal_y"], out["normal_z"]], 1)
    assert (got == po[:, :3]).all() and (gotn == no[:, :3]).all()


def test_cov_knn_matches_oracle(ctx, capi, oracle):
    pts = synth.scan(rings=16, azimuths=400, scale=1.0, seed=6)
    c = capi.Cloud(ctx, pts)
    cov = c.cov_knn(20, 1e-3)
    co = oracle.cov_knn(oracle.xyz4(pts), 20, 1e-3, threads=4)
    # C = I - (1-eps) u u^T: compare as matrices (sign of u irrelevant); Jacobi vs Jacobi, same neighbour order
    err = np.abs(cov - co).max(axis=(1, 2))
    assert np.quantile(err, 0.999) < 1e-9, np.sort(err)[-5:]
    assert (err < 1e-6).mean() > 0.9995  # near-degenerate patches (two equal small eigenvalues) may rotate


def test_sweep_and_cost_match_oracle(ctx, capi, oracle):
    src, tgt, delta = synth.scan_pair(n_rings=16, n_az=500, scale=1.0, noise=0.01, seed=7)
    ttree = oracle.Tree(oracle.xyz4(tgt))
    nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4, tree=ttree)
    ns = oracle.normals_knn(oracle.xyz4(src), 20, threads=4)
    P = capi.default_params(corr_dist=1.0)
    g = capi.Gicp(ctx, P)
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    x = np.array([0.05, -0.03, 0.01, 0.002, -0.001, 0.01])
    T16 = oracle.apply_state(x)
    idx, maha = g.debug_sweep(T16, src.shape[0])
    Tm = oracle.T_to_mat(T16)
    io, mo = oracle.nn_mahalanobis(oracle.xyz4(src), ttree, oracle.cov_from_normals(ns), oracle.cov_from_normals(nt), T16,
                                   Tm[:3, :3], 1.0, threads=4)
    assert (idx == io).all()  # correspondences bit-exact
    sel = io >= 0
    assert sel.sum() > 0.9 * len(sel)
    rel = np.abs(maha[sel] - mo[sel]).max(axis=(1, 2)) / np.abs(mo[sel]).max(axis=(1, 2))
    assert rel.max() < 1e-11  # tolerance: 6-entry symmetric storage vs the general 3x3 inverse (few ulp)
    # cost functor on those correspondences
    f, gr, sums, m = g.debug_cost(x)
    si = np.nonzero(sel)[0].astype(np.int32)
    fo, go, so = oracle.cost_fdf(oracle.xyz4(src), oracle.xyz4(tgt), si, io[sel], mo, x)
    assert m == sel.sum()
    assert abs(f - fo) <= 1e-11 * abs(fo)
    assert np.allclose(gr, go, rtol=1e-9, atol=1e-12)
    assert np.allclose(sums, so, rtol=1e-10, atol=1e-9)
    # second evaluation is bitwise reproducible (fixed reduction tree)
    f2, gr2, sums2, _ = g.debug_cost(x)
    assert f2 == f and (sums2 == sums).all()


def test_warm_sweeps_with_certificates_stay_bit_exact(ctx, capi, oracle):
    # consecutive sweeps reuse the previous neighbour as a bound and skip the traversal when a triangle-inequality
    # certificate proves the neighbour cannot change; every sweep must still equal the exhaustive (oracle) search
    src, tgt, delta = synth.scan_pair(n_rings=32, n_az=700, scale=2.0, noise=0.02, seed=17)
    ttree = oracle.Tree(oracle.xyz4(tgt))
    nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4, tree=ttree)
    ns = oracle.normals_knn(oracle.xyz4(src), 20, threads=4)
    g = capi.Gicp(ctx, capi.default_params(corr_dist=1.0))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    x_true = np.array([delta[0, 3], delta[1, 3], delta[2, 3], 0.0, 0.0, np.arctan2(delta[1, 0], delta[0, 0])])
    fracs = []
    g.debug_stats()
    for s in [0.0, 0.5, 0.8, 0.95, 0.99, 0.999, 0.9999, 1.0, 1.0, 0.3]:  # converging iterates, a repeat, then a jump back
        T16 = oracle.apply_state(x_true * s)
        idx, _ = g.debug_sweep(T16, src.shape[0])
        q = oracle.transform(oracle.xyz4(src), T16)
        io, do = ttree.nn1(q, threads=4)
        io = np.where(do.astype(np.float64) < 1.0, io, -1)
        assert (idx == io).all(), (s, int((idx != io).sum()))
        searched, total = g.debug_stats()
        fracs.append(searched / total)
    print("fraction of queries that ran the traversal per sweep:", np.round(fracs, 4))
    assert fracs[0] == 1.0 and fracs[8] < 0.01  # a repeated transform needs (almost) no traversal


def test_mode1_sweeps_keep_exact_neighbours(ctx, capi, oracle):
    # the cost_mode 1 kernels keep their neighbour state across sweeps with certificates and, from the fourth sweep on, the
    # two-launch form (k_late + k_walk with persistent lanes).  Whatever path a point takes, after EVERY sweep its neighbour must
    # be the exhaustive search's (lowest index among equals), and the matched count must be the gate's.
    src, tgt, delta = synth.scan_pair(n_rings=32, n_az=700, scale=2.0, noise=0.02, seed=29)
    ttree = oracle.Tree(oracle.xyz4(tgt))
    nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4, tree=ttree)
    ns = oracle.normals_knn(oracle.xyz4(src), 20, threads=4)
    g = capi.Gicp(ctx, capi.default_params(corr_dist=1.0))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    x_true = np.array([delta[0, 3], delta[1, 3], delta[2, 3], 0.0, 0.0, np.arctan2(delta[1, 0], delta[0, 0])])
    # a GICP-like sequence: identity, a big first step, converging iterates down to float noise, repeats, a jump back, converging again
    scales = [0.0, 0.9, 0.97, 0.99, 0.997, 0.999, 0.9997, 0.9999, 0.99997, 0.99999, 1.0, 1.0, 1.0 + 1e-6, 0.5, 0.98, 0.995, 0.999, 0.9999, 1.0, 1.0]
    walks_log = []
    for k, sc in enumerate(scales):
        T16 = oracle.apply_state(x_true * sc)
        idx, walks, sums = g.debug_sweep_fused(T16, src.shape[0], k)
        q = oracle.transform(oracle.xyz4(src), T16)
        io, do = ttree.nn1(q, threads=4)
        assert (idx == io).all(), (k, sc, int((idx != io).sum()))
        assert sums[73] == float((do.astype(np.float64) < 1.0).sum()), (k, sums[73])
        walks_log.append(walks)
    print("tree walks per sweep:", walks_log)
    n = src.shape[0]
    assert walks_log[0] == n                      # cold
    assert walks_log[11] <= 30 and walks_log[19] <= 30  # a repeated transform: only the near-ties (within the test's safety margin) walk
    # and again from a cold start with every sweep in the two-launch form from the start of the warm phase (sweep_index >= 3 only
    # selects the kernels): same neighbours
    g2 = capi.Gicp(ctx, capi.default_params(corr_dist=1.0))
    g2.set_source(capi.make_pointf(src, ns))
    g2.set_target(capi.make_pointf(tgt, nt))
    for k, sc in enumerate([0.0, 0.9, 0.99, 0.999, 1.0]):
        T16 = oracle.apply_state(x_true * sc)
        idx, walks, _ = g2.debug_sweep_fused(T16, src.shape[0], 0 if k == 0 else 3 + k)
        q = oracle.transform(oracle.xyz4(src), T16)
        io, _ = ttree.nn1(q, threads=4)
        assert (idx == io).all(), ("two-launch", k, int((idx != io).sum()))


@pytest.mark.parametrize("n_src", [1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 1000, 1025, 2049])
def test_mode1_sweeps_ragged_sizes(ctx, capi, oracle, n_src):
    # source sizes around every granularity of the two-launch sweep (64-point waves, 256-point workgroups, 512-point spans, four spans
    # per k_walk workgroup): neighbours after a cold sweep, a warm fused sweep and three k_late + k_walk sweeps must be the exhaustive
    # search's, and the matched count the gate's
    rng = np.random.default_rng(1000 + n_src)
    tgt = synth.scan(rings=16, azimuths=300, scale=1.0, seed=5)[:, :3].astype(np.float32)
    pick = rng.choice(tgt.shape[0], n_src, replace=n_src > tgt.shape[0])
    src = (tgt[pick] + rng.normal(0, 0.03, (n_src, 3))).astype(np.float32)
    ttree = oracle.Tree(oracle.xyz4(tgt))
    nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4, tree=ttree)
    ns = np.tile(np.array([0.0, 0.0, 1.0, 0.0], np.float32), (n_src, 1))      # any finite normals do: the neighbours do not depend on them
    g = capi.Gicp(ctx, capi.default_params(corr_dist=0.5))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    x = np.array([0.2, -0.1, 0.05, 0.01, -0.02, 0.03])
    for k, (sc, sweep_index) in enumerate([(0.0, 0), (0.8, 1), (0.97, 3), (0.999, 4), (1.0, 5), (1.0, 6)]):
        T16 = oracle.apply_state(x * sc)
        idx, walks, sums = g.debug_sweep_fused(T16, n_src, sweep_index)
        q = oracle.transform(oracle.xyz4(src), T16)
        io, do = ttree.nn1(q, threads=1)
        assert (idx == io).all(), (n_src, k, int((idx != io).sum()))
        assert sums[73] == float((do.astype(np.float64) < 0.25).sum()), (n_src, k, sums[73])
        assert walks <= n_src


def test_sweep_queries_outside_the_target_box_stay_bit_exact(ctx, capi, oracle):
    # node boxes are 16-bit fixed point on the target's own grid; a query outside that grid is clamped onto it and carries
    # its overshoot as a separate term.  Shift the source a little, a lot and absurdly far out of the target's bounding box:
    # the nearest neighbours must still be the exhaustive-search ones, bit for bit.
    src, tgt, _ = synth.scan_pair(n_rings=16, n_az=400, scale=1.0, noise=0.01, seed=23)
    ttree = oracle.Tree(oracle.xyz4(tgt))
    nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4, tree=ttree)
    ns = oracle.normals_knn(oracle.xyz4(src), 20, threads=4)
    g = capi.Gicp(ctx, capi.default_params(corr_dist=1.0e7))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    for shift in [(0.7, -0.4, 0.2), (35.0, 5.0, -3.0), (-900.0, 1400.0, 60.0), (2.0e5, -1.0e5, 3.0e4), (0.0, 0.0, 0.0)]:
        T16 = oracle.apply_state(np.array([shift[0], shift[1], shift[2], 0.01, -0.02, 0.3]))
        idx, _ = g.debug_sweep(T16, src.shape[0])
        q = oracle.transform(oracle.xyz4(src), T16)
        io, do = ttree.nn1(q, threads=4)
        io = np.where(do.astype(np.float64) < 1.0e14, io, -1)
        assert (idx == io).all(), (shift, int((idx != io).sum()))


def test_start_grid_edge_cases_stay_bit_exact(ctx, capi, oracle):
    # clouds of >= 8 192 points carry the start grid (a warm walk begins at the query's own cell and owes the rest of the cloud only the
    # neighbour cells its candidate's ball reaches).  Edge cases of that construction, every sweep against the exhaustive search: far
    # outliers stretch the key grid until the whole scene sits in a handful of cells and the outliers' leaves span thousands of empty
    # ones; queries inside, just outside and far outside the grid; candidates from nearby (finest table) to metres away (coarser tables,
    # then the walk from the root); both sweep flavours (reference-arithmetic kernel, cost_mode 1 kernels incl. k_late + k_walk).
    base = synth.scan(rings=24, azimuths=600, scale=2.0, seed=41)[:, :3].astype(np.float32)
    assert base.shape[0] >= 8192
    rng = np.random.default_rng(42)
    targets = {"plain": base,
               "outliers": np.concatenate([base, np.array([[4000.0, 0.0, 0.0], [0.0, -3000.0, 10.0], [-50.0, 60.0, 900.0]], np.float32)])}
    src = (base[rng.choice(base.shape[0], 9000, replace=False)] + rng.normal(0, 0.02, (9000, 3))).astype(np.float32)
    ns = np.tile(np.array([0.0, 0.0, 1.0, 0.0], np.float32), (src.shape[0], 1))
    for name, tgt in targets.items():
        ttree = oracle.Tree(oracle.xyz4(tgt))
        nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4, tree=ttree)
        g = capi.Gicp(ctx, capi.default_params(corr_dist=1.0e7))
        g.set_source(capi.make_pointf(src, ns))
        g.set_target(capi.make_pointf(tgt, nt))
        # consecutive sweeps: each one's candidates are the previous one's neighbours, so the shifts between them set the ball radii
        shifts = [(0.0, 0.0, 0.0), (0.05, -0.03, 0.01), (0.6, 0.4, -0.1), (3.0, -2.0, 0.5), (11.0, 7.0, 1.0), (60.0, -45.0, 6.0),
                  (-900.0, 1400.0, 60.0), (0.0, 0.0, 0.0), (0.01, 0.0, 0.0)]
        for k, sh in enumerate(shifts):
            T16 = oracle.apply_state(np.array([sh[0], sh[1], sh[2], 0.004 * k, -0.003 * k, 0.01 * k]))
            idx, _ = g.debug_sweep(T16, src.shape[0])
            q = oracle.transform(oracle.xyz4(src), T16)
            io, do = ttree.nn1(q, threads=4)
            assert (idx == io).all(), (name, "sweep", k, int((idx != io).sum()))
        g2 = capi.Gicp(ctx, capi.default_params(corr_dist=1.0e7))
        g2.set_source(capi.make_pointf(src, ns))
        g2.set_target(capi.make_pointf(tgt, nt))
        for k, sh in enumerate(shifts):
            T16 = oracle.apply_state(np.array([sh[0], sh[1], sh[2], 0.004 * k, -0.003 * k, 0.01 * k]))
            idx, walks, sums = g2.debug_sweep_fused(T16, src.shape[0], k)
            q = oracle.transform(oracle.xyz4(src), T16)
            io, do = ttree.nn1(q, threads=4)
            assert (idx == io).all(), (name, "fused", k, int((idx != io).sum()))
            assert sums[73] == float(src.shape[0]), (name, k, sums[73])
        # the cold 1-NN lookups (lh_nn1: a descent below the query's own cell for a bound, then the grid walk)
        qs = np.concatenate([src, src + np.array([40.0, -30.0, 5.0], np.float32), (rng.normal(size=(2000, 3)) * [3000, 3000, 500]).astype(np.float32)])
        idx, d2 = capi.Cloud(ctx, tgt).nn1(capi.Cloud(ctx, qs))
        io, do = ttree.nn1(oracle.xyz4(qs), threads=4)
        assert (idx == io).all() and (d2 == do).all(), name


def test_voxel_grid_bit_exact(ctx, capi, oracle):
    pts = synth.scan(rings=32, azimuths=900, scale=2.0, seed=8)
    rng = np.random.default_rng(9)
    xyzi = np.concatenate([pts, rng.uniform(0, 255, size=(pts.shape[0], 1)).astype(np.float32)], 1)
    for leaf, ax, lo, hi in [(0.25, -1, -np.inf, np.inf), (0.1, 2, -100.0, 100.0), (0.5, 2, -1.0, 1.0)]:
        out, cnt = ctx.voxel_grid(capi.make_pointxyzi(xyzi[:, :3], xyzi[:, 3]), leaf, ax, lo, hi)
        ref = oracle.voxel_grid(xyzi, leaf, ax, lo, hi)
        assert cnt == ref.shape[0]
        assert (out == ref).all()  # same voxel order, same in-voxel summation order => bit-identical centroids
    with pytest.raises(capi.LocusHipError):
        ctx.voxel_grid(capi.make_pointxyzi(xyzi[:, :3]), 1e-4)  # int32 voxel index overflow guard


def test_voxel_grid_pointf_bit_exact(ctx, capi, oracle):
    # pcl::VoxelGrid<PointXYZINormal> of PointCloudFilter::Filter (PointCloudFilter.cc:119-124): all eight fields, same order
    pts = synth.scan(rings=32, azimuths=900, scale=2.0, seed=8)
    rng = np.random.default_rng(11)
    inten = rng.uniform(0, 255, size=pts.shape[0]).astype(np.float32)
    c = capi.Cloud(ctx, capi.make_pointxyzi(pts, inten))
    c.normals_knn(10)                      # normals + curvature from the K3 kernel
    a = c.download()
    nrm4 = np.stack([a["normal_x"], a["normal_y"], a["normal_z"], a["curvature"]], 1)
    c = capi.Cloud(ctx, capi.make_pointf(pts, nrm4[:, :3], inten, nrm4[:, 3]))
    xyzi = np.concatenate([pts, inten[:, None]], 1)
    for leaf in (0.25, 0.1, 1.0):
        out = c.voxel_grid_pointf(leaf).download()
        ref, ref_n = oracle.voxel_grid_pointf(xyzi, nrm4, leaf)
        assert len(out) == ref.shape[0]
        got = np.stack([out["x"], out["y"], out["z"], out["intensity"]], 1)
        got_n = np.stack([out["normal_x"], out["normal_y"], out["normal_z"], out["curvature"]], 1)
        assert (got == ref).all()
        assert (got_n == ref_n).all()      # same summation order, float sqrt and division: bit-identical
    with pytest.raises(capi.LocusHipError):
        capi.Cloud(ctx, capi.make_pointxyzi(pts, inten)).voxel_grid_pointf(0.25)   # no normal fields: LH_EINVAL


def test_normals_match_oracle(ctx, capi, oracle):
    pts = synth.scan(rings=16, azimuths=600, scale=1.0, seed=10)
    out = ctx.normals_knn(pts, 20)
    ref = oracle.normals_knn(oracle.xyz4(pts), 20, threads=4)
    cosang = np.abs((out[:, :3] * ref[:, :3]).sum(1))
    # same float algorithm (pcl::eigen33 closed form); libm vs device atan2f/cosf/sinf differ by ulps, which the
    # closed form amplifies on near-isotropic patches -> tolerance on the angle, not bit-exactness
    assert np.quantile(cosang, 0.01) > 1 - 1e-4
    assert (np.sign((out[:, :3] * ref[:, :3]).sum(1)) > 0).mean() > 0.999  # same viewpoint flip
    assert np.allclose(np.linalg.norm(out[:, :3], axis=1), 1.0, atol=1e-5)


def test_radius_normals_match_oracle_and_nan_compaction(ctx, capi, oracle):
    # normal_search_method = radius (normal_computation.cc:71-74, radius 0.3) on a voxelised VLP-16-style scan: far, sparse
    # returns have < 3 neighbours -> NaN normal, and the nodelet drops them (normal_computation.cc:52-56)
    pts = synth.scan(rings=16, azimuths=900, scale=1.0, seed=21)
    far = np.array([[300.0, 0.0, 0.0], [0.0, 300.0, 1.0], [301.0, 0.1, 0.0]], np.float32)  # isolated: 1-2 neighbours only
    pts = np.concatenate([pts[:4000], far, pts[4000:]], 0)
    out = ctx.normals_radius(pts, 0.3)
    ref = oracle.normals_radius(oracle.xyz4(pts), 0.3, threads=4)
    nan_g, nan_o = np.isnan(out[:, 0]), np.isnan(ref[:, 0])
    assert (nan_g == nan_o).all() and nan_g[4000:4003].all()   # the neighbour SET (d2 < r2, float) is exact
    ok = ~nan_o
    assert ok.sum() > 0.5 * len(pts)
    # same float moments, but the device adds them in traversal order and PCL/the oracle in distance order: the one-pass
    # float covariance (E[xx] - E[x]E[x]) is ill-conditioned by |p|^2 / r^2, so parity is an angle tolerance, not bit-exact
    cosang = np.abs((out[ok, :3] * ref[ok, :3]).sum(1))
    assert np.quantile(cosang, 0.05) > 1 - 1e-3, np.quantile(cosang, [0.01, 0.05, 0.5])
    assert np.median(cosang) > 1 - 1e-5
    assert np.allclose(np.linalg.norm(out[ok, :3], axis=1), 1.0, atol=1e-5)
    # device-resident flavour + compaction
    c = capi.Cloud(ctx, capi.make_pointxyzi(pts, np.arange(len(pts), dtype=np.float32)))
    c.normals_radius(0.3)
    full = c.download()
    assert (np.isnan(full["normal_x"]) == nan_o).all()
    kept = c.remove_nan_normals()
    assert len(kept) == int(ok.sum())
    k = kept.download()
    assert (k["intensity"] == np.arange(len(pts), dtype=np.float32)[ok]).all()       # order preserved, same survivors
    assert (np.stack([k["x"], k["y"], k["z"]], 1) == pts[ok]).all()
    assert np.array_equal(k["normal_x"], full["normal_x"][ok])


def test_cloud_slice_and_concat(ctx, capi):
    # lh_cloud_concat = PointCloudMerger.cc:158-159 (`*merged = *a + *b`); lh_cloud_slice = a rank's source shard
    rng = np.random.default_rng(5)
    a = capi.make_pointf(rng.normal(size=(1000, 3)).astype(np.float32), rng.normal(size=(1000, 3)).astype(np.float32))
    b = capi.make_pointf(rng.normal(size=(37, 3)).astype(np.float32), rng.normal(size=(37, 3)).astype(np.float32))
    a["intensity"] = np.arange(1000)
    b["intensity"] = 5000 + np.arange(37)
    ca, cb = capi.Cloud(ctx, a), capi.Cloud(ctx, b)
    m = capi.Cloud.concat([ca, cb, ca]).download()
    for f in ("x", "y", "z", "normal_x", "normal_y", "normal_z", "intensity"):
        assert np.array_equal(m[f], np.concatenate([a[f], b[f], a[f]]))
    sl = ca.slice(123, 456).download()
    for f in ("x", "y", "z", "normal_x", "normal_y", "normal_z", "intensity"):
        assert np.array_equal(sl[f], a[f][123:123 + 456])
    with pytest.raises(capi.LocusHipError):
        ca.slice(900, 200)
    # a cloud without normals in the mix: the merged cloud has none either (zeros on download)
    cx = capi.Cloud(ctx, capi.make_pointxyzi(rng.normal(size=(10, 3)).astype(np.float32)))
    mm = capi.Cloud.concat([ca, cx]).download()
    assert len(mm) == 1010 and np.array_equal(mm["x"][:1000], a["x"])


def test_body_filter_crop_box(ctx, capi):
    # BodyFilter (body_filter.cc:27-52): CropBox(min, max, yaw) with setNegative(true) removes the robot's own returns
    rng = np.random.default_rng(13)
    pts = (rng.uniform(-3, 3, size=(20000, 3))).astype(np.float32)
    pts[7] = [np.nan, 0, 0]
    inten = np.arange(len(pts), dtype=np.float32)
    mn, mx, yaw = np.array([-0.8, -0.5, -0.4], np.float32), np.array([0.9, 0.5, 0.6], np.float32), 0.3
    c, s = np.float32(np.cos(np.float32(yaw))), np.float32(np.sin(np.float32(yaw)))
    lx, ly, lz = c * pts[:, 0] + s * pts[:, 1], c * pts[:, 1] - s * pts[:, 0], pts[:, 2]
    with np.errstate(invalid="ignore"):
        outside = (lx < mn[0]) | (ly < mn[1]) | (lz < mn[2]) | (lx > mx[0]) | (ly > mx[1]) | (lz > mx[2])
    finite = np.isfinite(pts).all(1)
    # points whose rotated coordinate is within rounding of a face may differ between numpy's and the device's cosf/sinf
    margin = np.minimum.reduce([np.abs(lx - mn[0]), np.abs(lx - mx[0]), np.abs(ly - mn[1]), np.abs(ly - mx[1])]) < 1e-5
    cloud = capi.Cloud(ctx, capi.make_pointxyzi(pts, inten))
    for negative in (True, False):
        keep = finite & (outside if negative else ~outside)
        got = cloud.crop_box(mn, mx, yaw, negative).download()
        ids = got["intensity"].astype(np.int64)
        assert (np.diff(ids) > 0).all()                                    # order preserved
        expect = np.where(keep)[0]
        diff = np.setxor1d(ids, expect)
        assert all(margin[d] for d in diff) and len(diff) <= 2, diff       # identical away from the faces
        assert np.array_equal(np.stack([got["x"], got["y"], got["z"]], 1), pts[ids])
    assert 0 < (finite & ~outside).sum() < 2000                            # the box really removed something


def test_p2plane_information_kat_and_oracle(ctx, capi, oracle):
    pts, nrm = synth.plane_grid(10, 10, 0.1)
    c = capi.Cloud(ctx, capi.make_pointf(pts, nrm))
    Ap = ctx.p2plane_information(c, c, np.arange(100))
    assert abs(Ap[0, 0] - 56.7753) < 1e-4 and abs(Ap[1, 1] - 56.7753) < 1e-4 and abs(Ap[5, 5] - 100.0) < 1e-4  # reference KAT at the reference's epsilion (test_point_cloud_localization.cpp:24,337-339)
    src, nrms = synth.scan(rings=16, azimuths=500, scale=1.0, seed=11, with_normals=True)
    rng = np.random.default_rng(12)
    corr = rng.integers(0, src.shape[0], size=src.shape[0])
    cs = capi.Cloud(ctx, capi.make_pointf(src, nrms))
    Ap = ctx.p2plane_information(cs, cs, corr)
    Ao = oracle.p2plane_Ap(oracle.normalize_cloud(oracle.xyz4(src)), oracle.nrm4(nrms), corr)
    # the reference/oracle normalisation sums sequentially in float; the HIP path sums in double -> 1e-4 relative
    assert np.allclose(Ap, Ao, rtol=2e-4, atol=2e-4 * np.abs(Ao).max())


def test_index_sort_matches_library_sort():
    """K2's segmented radix sort (three 10-bit passes inside each cloud's segment, packed pairs) against the one-segment 64-bit sort
    of the same (cloud id << 32 | key) array -- the sort the voxel grid and the local map use, itself held bit for bit to the
    sequential restatements by test_voxel_grid_bit_exact / test_local_map_insert_refresh_and_scan_to_map: LH_SORT=check runs both
    inside every index build and fails the call on the first differing element.  Ragged batches, tiny clouds, runs of identical
    keys (ties must keep ascending point index), a 300 k-point cloud (147 tiles).  (No library sort is linked any more.)"""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from locus_amd import capi, synth
ctx = capi.Context(0)
rng = np.random.default_rng(5)
clouds = []
for n in (1, 2, 7, 63, 64, 65, 2047, 2048, 2049, 5000, 100032, 300000):
    pts = rng.uniform(-30, 30, size=(n, 3)).astype(np.float32)
    clouds.append(pts)
dup = np.repeat(rng.uniform(-5, 5, size=(40, 3)).astype(np.float32), 100, axis=0)   # 40 points x 100 copies: long runs of equal keys
clouds.append(dup)
clouds.append(synth.scan(rings=32, azimuths=900, scale=2.0, seed=8))
C = [capi.Cloud(ctx, c) for c in clouds]
for c in C:
    c.build_index()                     # one cloud per build
# batched builds (one sort for all targets of a batch): ragged sizes, more clouds than one launch of 64
S, T = [], []
for k in range(70):
    src, tgt, _ = synth.scan_pair(n_rings=4 + k %% 13, n_az=100 + 37 * (k %% 7), scale=1.0, noise=0.01, seed=700 + k)
    cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
    cs.normals_knn(5); ct.normals_knn(5); ct.drop_index()
    S.append(cs); T.append(ct)
out = capi.align_batch(ctx, capi.default_params(max_iterations=3, corr_dist=1.0), S, T, max_in_flight=70)
assert all(o["status"] == 0 for o in out)
print("SORT_CHECK_OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LH_SORT="check")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SORT_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

