"""GPU suite (-m gpu): whole-alignment parity (lh_gicp_align / lh_gicp_align_batch) against the CPU oracle.
Tolerances: |dt| <= 1e-4 m, |dR| <= 1e-4 (max abs entry ~ rad), fitness rel 1e-4."""
import numpy as np
import pytest

from locus_amd import synth

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-4, 1e-4
# cost_mode 1 evaluates T*p in double instead of float: it differs from the reference arithmetic by the reference's own
# float rounding noise.  The same source built with / without FMA contraction moves the reference's result by up to
# 1e-3 m (tests/test_oracle_kats.py::test_reference_float_noise_floor) -- the order of the stopping threshold
# transformation_epsilon = 1e-3 / rotation_epsilon = 2e-3 themselves (the iteration stops as soon as the update is below
# them) -- so mode 1 is held to 2e-3 m / 2.5e-3 under production stopping (measured: <= 1e-3 m, <= 1.8e-3), and to
# 2e-4 when the iteration count is forced (test_forced_20_iterations_and_guess).
TOL_T1, TOL_R1 = 2e-3, 2.5e-3


def _tol(cost_mode):
    return (TOL_T, TOL_R) if cost_mode == 0 else (TOL_T1, TOL_R1)


def _pose_err(capi_T, oracle_T, oracle):
    A, B = oracle.T_to_mat(capi_T), oracle.T_to_mat(oracle_T)
    return np.abs(A[:3, 3] - B[:3, 3]).max(), np.abs(A[:3, :3] - B[:3, :3]).max()


def _pair_with_normals(oracle, seed, rings=16, az=600, scale=1.0):
    src, tgt, delta = synth.scan_pair(n_rings=rings, n_az=az, scale=scale, noise=0.01, seed=seed)
    ns = oracle.normals_knn(oracle.xyz4(src), 20, threads=4)
    nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4)
    return src, ns, tgt, nt, delta


@pytest.mark.parametrize("cost_mode", [0, 1])
def test_garage_fixture_knn_covariances(ctx, capi, oracle, garage, cost_mode):
    # the reference's own fixture + parameter set (test_same_output_different_num_threads.cpp:31-36), k-NN branch
    q, r = garage
    kw = dict(transformation_epsilon=1e-10, corr_dist=0.2, max_iterations=20, max_inner_iterations=50,
              recompute_source_cov=1, recompute_target_cov=1)
    g = capi.Gicp(ctx, capi.default_params(cost_mode=cost_mode, **kw))
    g.set_source(capi.make_pointxyzi(q[:, :3], q[:, 3]))
    g.set_target(capi.make_pointxyzi(r[:, :3], r[:, 3]))
    res = g.align()
    ro = oracle.gicp_align(oracle.xyz4(q), None, oracle.xyz4(r), None, oracle.default_params(num_threads=4, **kw))
    assert res["status"] == 0 and ro["status"] == 0
    dt, dR = _pose_err(res["T"], ro["T"], oracle)
    print("garage cost_mode", cost_mode, "dt", dt, "dR", dR)
    assert dt < _tol(cost_mode)[0] and dR < _tol(cost_mode)[1], (dt, dR)
    assert res["converged"] == ro["converged"]
    fit = g.fitness()
    fo = oracle.fitness(oracle.xyz4(q), ro["T"], oracle.Tree(oracle.xyz4(r)), threads=4)
    assert abs(fit - fo) <= (1e-4 if cost_mode == 0 else 2e-3) * fo
    # "identical output for any thread count" -> here: identical output run to run (fixed reduction shapes)
    res2 = g.align()
    assert (res2["T"] == res["T"]).all()


@pytest.mark.parametrize("cost_mode", [0, 1])
@pytest.mark.parametrize("seed", [21, 22, 23])
def test_scan_pair_from_normals_matches_oracle(ctx, capi, oracle, seed, cost_mode):
    src, ns, tgt, nt, delta = _pair_with_normals(oracle, seed)
    kw = dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)  # point_cloud_odometry/config/parameters.yaml
    g = capi.Gicp(ctx, capi.default_params(cost_mode=cost_mode, **kw))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    res = g.align()
    ro = oracle.gicp_align(oracle.xyz4(src), ns, oracle.xyz4(tgt), nt, oracle.default_params(num_threads=4, **kw))
    dt, dR = _pose_err(res["T"], ro["T"], oracle)
    print("cost_mode", cost_mode, "seed", seed, "dt", dt, "dR", dR, "iters", res["iterations"], ro["iterations"])
    assert dt < _tol(cost_mode)[0] and dR < _tol(cost_mode)[1], (dt, dR)
    assert res["converged"] == ro["converged"]
    k = min(len(res["trace"]["n_corr"]), len(ro["trace"]["n_corr"]))
    assert res["trace"]["n_corr"][0] == ro["trace"]["n_corr"][0]  # first sweep: identical inputs => identical correspondences
    if cost_mode == 0:  # same arithmetic: the whole trajectory matches
        assert res["iterations"] == ro["iterations"] and res["n_corr_last"] == ro["n_corr_last"]
        assert (res["trace"]["n_corr"][:k] == ro["trace"]["n_corr"][:k]).all()
    assert np.abs(res["trace"]["T"][:k] - ro["trace"]["T"][:k]).max() < _tol(cost_mode)[0]
    assert np.allclose(res["trace"]["f_end"][:k], ro["trace"]["f_end"][:k], rtol=1e-6 if cost_mode == 0 else 5e-3)
    # recovered the simulated motion
    Tm = oracle.T_to_mat(res["T"])
    assert np.abs(Tm[:3, 3] - delta[:3, 3]).max() < 0.03


@pytest.mark.parametrize("cost_mode,solver", [(0, 0), (1, 1), (1, 2)])
def test_bfgs_curvature_switch_follows_the_oracle(ctx, capi, oracle, cost_mode, solver):
    """lh_gicp_params::bfgs_quad_curv = 1 (pcl::BFGS's reported `c > a` reading of the quadratic interpolation's curvature test; the reference calls
    pcl::BFGS at gicp.hpp:249-271 and its source is not in the tree): the product with the switch follows the ORACLE with the same switch
    (lo_set_bfgs_variant) as closely as it follows it without -- on the host loop, the device loop (k_solve) and in the strict mode."""
    src, ns, tgt, nt, delta = _pair_with_normals(oracle, 22)
    kw = dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    L = oracle.lib()
    out = {}
    for variant in (0, 1):
        g = capi.Gicp(ctx, capi.default_params(cost_mode=cost_mode, solver=solver, bfgs_quad_curv=variant, **kw))
        g.set_source(capi.make_pointf(src, ns))
        g.set_target(capi.make_pointf(tgt, nt))
        res = g.align()
        L.lo_set_bfgs_variant(variant)
        try:
            ro = oracle.gicp_align(oracle.xyz4(src), ns, oracle.xyz4(tgt), nt, oracle.default_params(num_threads=4, **kw))
        finally:
            L.lo_set_bfgs_variant(0)
        dt, dR = _pose_err(res["T"], ro["T"], oracle)
        assert res["status"] == 0 and dt < 2e-4 and dR < 2e-4, (variant, dt, dR)
        if cost_mode == 0:   # same arithmetic: the whole trajectory matches, iteration by iteration
            k = min(len(res["trace"]["n_corr"]), len(ro["trace"]["n_corr"]))
            assert res["iterations"] == ro["iterations"] and (res["trace"]["n_corr"][:k] == ro["trace"]["n_corr"][:k]).all()
            assert np.abs(res["trace"]["T"][:k] - ro["trace"]["T"][:k]).max() < 1e-5
        out[variant] = (res, ro)
    # the switch is not a no-op on this pair: the oracle's two readings take different numbers of functor evaluations
    assert sum(out[0][1]["trace"]["n_passes"]) != sum(out[1][1]["trace"]["n_passes"]) or not np.array_equal(np.asarray(out[0][1]["T"]), np.asarray(out[1][1]["T"]))


@pytest.mark.parametrize("cost_mode", [0, 1])
def test_forced_20_iterations_and_guess(ctx, capi, oracle, cost_mode):
    src, ns, tgt, nt, delta = _pair_with_normals(oracle, 31)
    kw = dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)  # delta<1 never true
    guess = synth.pose_matrix(0.05, -0.02, 0.0, 0, 0, 0.01).astype(np.float32)
    g = capi.Gicp(ctx, capi.default_params(cost_mode=cost_mode, **kw))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    res = g.align(guess=oracle.mat_to_T(guess))
    ro = oracle.gicp_align(oracle.xyz4(src), ns, oracle.xyz4(tgt), nt, oracle.default_params(num_threads=4, **kw),
                           guess=oracle.mat_to_T(guess))
    dt, dR = _pose_err(res["T"], ro["T"], oracle)
    print("cost_mode", cost_mode, "dt", dt, "dR", dR, "iters", res["iterations"], ro["iterations"])
    if cost_mode == 0:
        assert res["iterations"] == ro["iterations"]
    assert dt < 2e-4 and dR < 2e-4, (dt, dR)  # fully iterated: both modes agree with the oracle to 2e-4


def test_hollow_cube_kat_on_gpu(ctx, capi, oracle):
    # UpdateEstimateUpdateICP (test_point_cloud_odometry.cpp:280-305)
    cube = synth.hollow_cube()
    nrm = ctx.normals_knn(cube, 5)
    moved = cube + np.array([0.05, 0.05, 0.0], np.float32)
    g = capi.Gicp(ctx, capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3))
    g.set_source(capi.make_pointf(moved, np.zeros_like(moved)))
    g.set_target(capi.make_pointf(cube, nrm))
    res = g.align()
    assert res["status"] == 0 and res["converged"] == 1
    Tinv = np.linalg.inv(oracle.T_to_mat(res["T"]))
    assert abs(Tinv[0, 3] - 0.05) < 1e-2 and abs(Tinv[1, 3] - 0.05) < 1e-2 and abs(Tinv[2, 3]) < 1e-2
    assert g.fitness() < 0.1


def test_error_paths(ctx, capi, oracle):
    P = capi.default_params()
    g = capi.Gicp(ctx, P)
    three = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    g.set_source(capi.make_pointf(three, np.zeros_like(three)))
    g.set_target(capi.make_pointf(three, np.zeros_like(three)))
    res = g.align()
    assert res["status"] == capi.LH_ETOO_FEW_CORR and res["converged"] == 0  # gicp.hpp:225, 542-547
    assert np.allclose(oracle.T_to_mat(res["T"]), np.eye(4))
    # missing normals while recompute_*=false
    g2 = capi.Gicp(ctx, P)
    g2.set_source(capi.make_pointxyzi(three))
    g2.set_target(capi.make_pointxyzi(three))
    with pytest.raises(capi.LocusHipError):
        g2.align()
    # k_correspondences > cloud size (gicp.hpp:72-79)
    g3 = capi.Gicp(ctx, capi.default_params(recompute_source_cov=1, recompute_target_cov=1))
    g3.set_source(capi.make_pointxyzi(three))
    g3.set_target(capi.make_pointxyzi(three))
    with pytest.raises(capi.LocusHipError):
        g3.align()


def test_batch_equals_single_and_promote(ctx, capi, oracle):
    pairs = [_pair_with_normals(oracle, 40 + i, rings=16, az=300) for i in range(5)]
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    singles = []
    for src, ns, tgt, nt, _ in pairs:
        g = capi.Gicp(ctx, P)
        g.set_source(capi.make_pointf(src, ns))
        g.set_target(capi.make_pointf(tgt, nt))
        singles.append(g.align(want_trace=False))
    S = [capi.Cloud(ctx, capi.make_pointf(p[0], p[1])) for p in pairs]
    T = [capi.Cloud(ctx, capi.make_pointf(p[2], p[3])) for p in pairs]
    for in_flight in (1, 2, 5):
        out = capi.align_batch(ctx, P, S, T, max_in_flight=in_flight)
        for a, b in zip(out, singles):
            assert (a["T"] == b["T"]).all() and a["iterations"] == b["iterations"]  # batching never changes results
    # odometry fast path: promote_source_to_target == set_target of the same data
    g = capi.Gicp(ctx, P)
    g.set_source(capi.make_pointf(pairs[0][2], pairs[0][3]))
    g.promote_source_to_target()
    g.set_source(capi.make_pointf(pairs[0][0], pairs[0][1]))
    r = g.align(want_trace=False)
    assert (r["T"] == singles[0]["T"]).all()


def test_batch_out_writes_the_aligned_clouds(ctx, capi, oracle):
    """lh_gicp_align_batch_out: align()'s `output` cloud (gicp.hpp:586) for every pair of a batch -- final T * input, every other
    field of the point copied -- into clouds the call creates, or into clouds the caller re-uses"""
    pairs = [_pair_with_normals(oracle, 60 + i, rings=16, az=250 + 40 * i) for i in range(4)]
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    S = [capi.Cloud(ctx, capi.make_pointf(p[0], p[1], intensity=np.arange(p[0].shape[0]) % 200)) for p in pairs]
    T = [capi.Cloud(ctx, capi.make_pointf(p[2], p[3])) for p in pairs]
    plain = capi.align_batch(ctx, P, S, T, max_in_flight=3)
    res, A = capi.align_batch_out(ctx, P, S, T, max_in_flight=3)
    for r0, r, cs, ca, p in zip(plain, res, S, A, pairs):
        assert (r["T"] == r0["T"]).all() and r["iterations"] == r0["iterations"]      # asking for the output changes nothing
        want, got, src = cs.transform(r["T"]).download(), ca.download(), cs.download()
        for f in ("x", "y", "z"):
            assert (got[f] == want[f]).all()                                           # the K6 transform, bit for bit
        for f in ("normal_x", "normal_y", "normal_z", "intensity", "curvature"):
            assert (got[f] == src[f]).all()                                            # pcl::transformPointCloud leaves them alone
        # and it is the reference's output: oracle.transform of the input by the same matrix
        o = oracle.transform(oracle.xyz4(p[0]), r["T"])
        assert (np.stack([got["x"], got["y"], got["z"]], 1) == o[:, :3]).all()
    # second call into the SAME output clouds (a streaming caller keeps them): overwritten in place
    guesses = np.stack([oracle.mat_to_T(synth.pose_matrix(0.02 * i, -0.01, 0, 0, 0, 0.003 * i).astype(np.float32)) for i in range(4)])
    res2, A2 = capi.align_batch_out(ctx, P, S, T, guesses=guesses, max_in_flight=4, aligned=A)
    for r, cs, ca, ca2 in zip(res2, S, A, A2):
        assert ca2 is ca
        want, got = cs.transform(r["T"]).download(), ca.download()
        assert (got["x"] == want["x"]).all() and (got["z"] == want["z"]).all()
    with pytest.raises(capi.LocusHipError):     # an output cloud of the wrong size is refused
        capi.align_batch_out(ctx, P, S[:1], T[:1], aligned=[A[1]])


def test_batch_multi_contexts_and_views(ctx, capi, oracle):
    """lh_gicp_align_batch_multi(_views): pairs dispatched to the context that owns their clouds, one host
    thread per device, results in pair order and bit-identical to the single-context batch.  A 1-GPU box exercises the
    dispatch with two contexts on the same device (served one after the other) and with every visible device."""
    assert capi.device_count() >= 1
    pairs = [_pair_with_normals(oracle, 70 + i, rings=16, az=200 + 30 * i) for i in range(6)]
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    S = [capi.Cloud(ctx, capi.make_pointf(p[0], p[1])) for p in pairs]
    T = [capi.Cloud(ctx, capi.make_pointf(p[2], p[3])) for p in pairs]
    single = capi.align_batch(ctx, P, S, T, max_in_flight=4)
    ctx2 = capi.Context(0)
    ctxs = [ctx, ctx2] + [capi.Context(d) for d in range(1, capi.device_count())]
    owner = [ctxs[i % len(ctxs)] for i in range(6)]   # interleaved on purpose: results must come back in pair order
    S2 = [capi.Cloud(owner[i], capi.make_pointf(pairs[i][0], pairs[i][1])) for i in range(6)]
    T2 = [capi.Cloud(owner[i], capi.make_pointf(pairs[i][2], pairs[i][3])) for i in range(6)]
    multi = capi.align_batch_multi(ctxs, P, S2, T2, max_in_flight=4)
    for a, b in zip(multi, single):
        assert a["status"] == b["status"] and a["iterations"] == b["iterations"] and (a["T"] == b["T"]).all()
    with pytest.raises(capi.LocusHipError):           # a pair whose clouds live on a context that was not passed in
        capi.align_batch_multi([ctx2], P, S, T)
    with pytest.raises(capi.LocusHipError):           # source and target on different contexts
        capi.align_batch_multi(ctxs, P, [S2[0]], [T2[1]])
    # host-resident odometry stream: scan i is the source of pair i and the target of pair i + 1 (the same buffer)
    scans = [capi.make_pointf(pairs[i][0], pairs[i][1]) for i in range(5)]
    src_v, tgt_v = scans[1:], scans[:-1]
    views = capi.align_batch_multi_views(ctxs, P, src_v, tgt_v, max_in_flight=4)
    ref = capi.align_batch(ctx, P, [capi.Cloud(ctx, a) for a in src_v], [capi.Cloud(ctx, a) for a in tgt_v], max_in_flight=4)
    for a, b in zip(views, ref):
        assert a["status"] == b["status"] and (a["T"] == b["T"]).all()
    for c in S2 + T2:
        c.close()
    for c in ctxs[1:]:
        c.close()


# ---- the benched configuration itself (BASELINE configs[1]): bench.py's own pairs, full size, against the oracle -------------
# The reference's own float noise floor at this configuration (its source built with vs. without FMA contraction in the
# functor's float T*p, tests/perf/reference_noise_floor.py -> profiles/r02_reference_noise_floor.json, 16 bench pairs):
#   |dt| median 9.8e-5 m, 15 of 16 pairs <= 2.4e-4 m, one pair 2.5e-3 m; |dR| max 1.24e-4; intermediate iterates up to 3.4e-3
# (the loop stalls where BFGS reports "no progress" / |g| < 1e-2, and a last-bit difference in a line-search comparison moves
# that point).  SURVEY 8d's 1e-4 therefore holds for the reference-arithmetic mode (cost_mode 0: every per-point operation
# as in gicp.hpp:362-402; only the order of the 14 sums differs); the moment mode (cost_mode 1, the benched default) is a
# bit-different evaluation of the same cost and is held to max(1e-4, floor): every pair inside the floor's maximum, the
# typical pair inside its 15-of-16 value.
FLOOR_T_MAX, FLOOR_R_MAX, FLOOR_T_TYPICAL = 2.5e-3, 1.3e-4, 2.4e-4
N_BENCH_PAIRS = 32
BENCH_SEEDS = tuple(10 + 2 * p for p in range(N_BENCH_PAIRS))   # bench.py gen_pairs_host(), rank 0, pairs 0..31
# Quantile bars of the benched mode over those pairs (profiles/r03_fullsize_parity.json: the same table over 64 pairs, next to the
# distance between the reference's own two builds): the typical pair meets SURVEY 8d's 1e-4 m, nine in ten are within 2.5e-4 m,
# and a pair beyond that is accepted only if the reference ITSELF moves by more than 2.5e-4 m on that pair when its float
# product is contracted (its line search stalls elsewhere on the valley floor), and never beyond HARD_T.
Q_MEDIAN_T, Q_P90_T, HARD_T = 1e-4, 2.5e-4, 5e-3
# The other columns of the same table (profiles/r04_parity_distributions.json, 64 pairs, forced 20 iterations; in brackets the distance between
# the reference's own two builds):
#   fitness, relative: median 5.2e-5 [8.2e-5], p90 2.0e-4 [2.1e-4], max 6.6e-4 [9.7e-4]          -> SURVEY 8d's 1e-4 at the median
#   largest per-iteration |dT| of a pair: median 5.0e-4 [4.0e-4], p90 3.2e-3 [7.1e-3], max 1.5e-1 [1.9e-2] (one pair whose first solve
#     ends 0.15 m from the reference's first solve and which still finishes 4e-5 m from it: north_star's "per-iteration results match"
#     holds in distribution, not pair by pair, for the reference's own builds as well)
Q_FIT_MEDIAN, Q_FIT_P90, FIT_HARD = 1e-4, 5e-4, 2e-3
Q_ITER_MEDIAN, Q_ITER_P90 = 1.5e-3, 1e-2
# ... and under the production stopping rule (the reference stops as soon as an update is below tf_eps 1e-3 / rotation_eps 2e-3, so its result
# is defined to that scale only): cost_mode 1 |dt| median 2.4e-4 [2.1e-4], p90 1.9e-3 [3.2e-3], max 9.0e-3 [2.1e-2]; |dR| max 1.4e-4 [7.9e-4];
# fitness median 1.6e-4 [1.7e-4], p90 1.8e-3 [2.3e-3]; cost_mode 0 |dt| median 0.0, p90 8.8e-5, max 4.3e-4, iteration counts equal on 64 / 64.
PRODUCTION_KW = dict(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, rotation_epsilon=2e-3)
PROD_Q_MEDIAN_T, PROD_Q_P90_T, PROD_HARD_T, PROD_HARD_R = 5e-4, 3e-3, 2e-2, 5e-4


def _oracle_inputs(cloud_s, cloud_t, oracle, src, tgt):
    a, b = cloud_s.download(), cloud_t.download()
    ns = oracle.nrm4(np.stack([a["normal_x"], a["normal_y"], a["normal_z"]], 1))
    nt = oracle.nrm4(np.stack([b["normal_x"], b["normal_y"], b["normal_z"]], 1))
    return oracle.xyz4(src), ns, oracle.xyz4(tgt), nt


@pytest.fixture(scope="module")
def bench_pairs(ctx, capi, oracle):
    """the bench's own first 32 pairs at full size, with the oracle's alignment of each (trace included).  The oracle runs are
    independent: they go through a thread pool (ctypes releases the GIL; 4 OMP threads each, like LOCUS on a Husky)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    kw = dict(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    out = []
    for seed in BENCH_SEEDS:
        src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=seed)
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        cs.normals_knn(20)   # k = 20 normals from the K3 kernel, exactly as bench.py prepares its inputs
        ct.normals_knn(20)
        out.append(dict(seed=seed, cs=cs, ct=ct, delta=delta, inputs=_oracle_inputs(cs, ct, oracle, src, tgt)))
    omp = 4
    workers = max(1, min(len(out), (os.cpu_count() or 4) // omp))

    def run(p):
        a = p["inputs"]
        tree = oracle.Tree(a[2])
        ro = oracle.gicp_align(a[0], a[1], a[2], a[3], oracle.default_params(num_threads=omp, **kw))
        fo = oracle.fitness(a[0], ro["T"], tree, threads=omp)
        # ... and under the stopping rule LOCUS runs with (parameters.yaml:12 tf_eps 1e-3, gicp.h:119 rotation_epsilon 2e-3; gicp.hpp:566)
        rp = oracle.gicp_align(a[0], a[1], a[2], a[3], oracle.default_params(num_threads=omp, **PRODUCTION_KW))
        fp = oracle.fitness(a[0], rp["T"], tree, threads=omp)
        return ro, fo, rp, fp
    with ThreadPoolExecutor(workers) as ex:
        for p, (ro, fo, rp, fp) in zip(out, ex.map(run, out)):
            p["ro"], p["fo"], p["ro_prod"], p["fo_prod"] = ro, fo, rp, fp
    return out, kw


def test_bench_pairs_device_loop_32_in_flight_vs_oracle(ctx, capi, oracle, bench_pairs):
    """The path bench.py times -- lh_gicp_align_batch_out, cost_mode 1, the DEVICE-driven loop (k_solve, group admission, two
    scheduler streams at 32 pairs in flight) -- on 32 of the bench's own 100 032-point pairs against the oracle
    (gicp.hpp:445-568): every pose, every iteration count, the cost at the end of the last solve; and the bridge at this size:
    device loop == host loop, bit for bit."""
    pairs, kw = bench_pairs
    S, T = [p["cs"] for p in pairs], [p["ct"] for p in pairs]
    P = capi.default_params(cost_mode=1, **kw)            # solver 0: the device loop from 8 pairs in flight on
    res, A = capi.align_batch_out(ctx, P, S, T, max_in_flight=32)
    host = capi.align_batch(ctx, capi.default_params(cost_mode=1, solver=1, **kw), S, T, max_in_flight=32)
    dts, drs, its, fits, iter_dts = [], [], [], [], []
    for p, r, h, a in zip(pairs, res, host, A):
        ro = p["ro"]
        assert r["status"] == 0 and ro["status"] == 0
        assert (r["T"] == h["T"]).all() and r["iterations"] == h["iterations"] and r["n_corr_last"] == h["n_corr_last"], p["seed"]   # device loop == host loop
        g = capi.Gicp(ctx, capi.default_params(cost_mode=1, solver=2, **kw))   # the same pair alone through k_solve, for its trace
        g.set_source(p["cs"])
        g.set_target(p["ct"])
        r1 = g.align()
        assert (r1["T"] == r["T"]).all() and r1["iterations"] == r["iterations"], p["seed"]   # batching / admission never change results
        fits.append(abs(g.fitness() - p["fo"]) / p["fo"])                      # getFitnessScore of the GPU's pose vs the oracle's score of its own
        kk = min(len(r1["trace"]["T"]), len(ro["trace"]["T"]))
        iter_dts.append(float(np.abs(r1["trace"]["T"][:kk] - ro["trace"]["T"][:kk]).max()))
        dt, dR = _pose_err(r["T"], ro["T"], oracle)
        dts.append(dt)
        drs.append(dR)
        its.append((r["iterations"], ro["iterations"]))
        assert r1["trace"]["n_corr"][0] == ro["trace"]["n_corr"][0]          # first sweep: identical inputs => identical correspondences
        assert abs(r["n_corr_last"] - ro["n_corr_last"]) <= 2e-3 * ro["n_corr_last"]
        assert abs(r1["trace"]["f_end"][-1] - ro["trace"]["f_end"][-1]) <= 2e-3 * ro["trace"]["f_end"][-1]   # the same valley floor
        assert r["iterations"] >= 4 and (r["iterations"] == 20 or r["converged"] == 1)
        # align()'s output cloud, written by the device loop as the pair retired (gicp.hpp:586)
        want, got = p["cs"].transform(r["T"]).download(), a.download()
        assert (got["x"] == want["x"]).all() and (got["y"] == want["y"]).all() and (got["z"] == want["z"]).all()
        Tm = oracle.T_to_mat(r["T"])
        assert np.abs(Tm[:3, 3] - p["delta"][:3, 3]).max() < 0.02
    dts, drs = np.array(dts), np.array(drs)
    print("device loop, %d bench pairs vs oracle: |dt| median %.2e p90 %.2e max %.2e | |dR| max %.2e | iterations (gpu, oracle) %s"
          % (len(pairs), np.median(dts), np.quantile(dts, 0.9), dts.max(), drs.max(), its))
    assert np.median(dts) <= Q_MEDIAN_T, np.median(dts)
    assert np.quantile(dts, 0.9) <= Q_P90_T, np.quantile(dts, 0.9)
    assert drs.max() <= max(TOL_R, FLOOR_R_MAX), drs.max()
    fits, iter_dts = np.array(fits), np.array(iter_dts)
    print("   fitness rel: median %.2e p90 %.2e max %.2e | largest per-iteration |dT|: median %.2e p90 %.2e max %.2e"
          % (np.median(fits), np.quantile(fits, 0.9), fits.max(), np.median(iter_dts), np.quantile(iter_dts, 0.9), iter_dts.max()))
    assert np.median(fits) <= Q_FIT_MEDIAN and np.quantile(fits, 0.9) <= Q_FIT_P90 and fits.max() <= FIT_HARD, fits
    assert np.median(iter_dts) <= Q_ITER_MEDIAN and np.quantile(iter_dts, 0.9) <= Q_ITER_P90, iter_dts
    # pairs beyond the p90 bar: only where the reference's own two builds part by more than that on the SAME pair
    L = oracle.lib()
    import os
    for k in np.nonzero(dts > Q_P90_T)[0]:
        a = pairs[k]["inputs"]
        L.lo_set_cost_variant(1)
        try:
            rf = oracle.gicp_align(a[0], a[1], a[2], a[3], oracle.default_params(num_threads=os.cpu_count() or 4, **kw), want_trace=False)
        finally:
            L.lo_set_cost_variant(0)
        floor_k, _ = _pose_err(rf["T"], pairs[k]["ro"]["T"], oracle)
        print("  pair seed %d: |dt| %.2e, reference FMA / non-FMA distance on this pair %.2e" % (pairs[k]["seed"], dts[k], floor_k))
        assert floor_k > Q_P90_T and dts[k] <= HARD_T, (pairs[k]["seed"], dts[k], floor_k)


def test_bench_pairs_reference_arithmetic_32_pairs(ctx, capi, oracle, bench_pairs):
    """cost_mode 0 (every per-point operation as in gicp.hpp:362-402) on the same 32 pairs as one batch.  What differs from the
    reference is the ORDER in which the 14 sums of an evaluation are added (the reference adds the correspondences one after the
    other; a parallel reduction cannot), i.e. the last bits of f and g -- and on some pairs that flips one comparison of the line
    search, after which the run stalls elsewhere on the valley floor, like the reference's own two builds do.  Measured over 64
    pairs (profiles/r03_fullsize_parity.json): median 0.0, 60 of 64 within SURVEY 8d's 1e-4 m, max 1.9e-4 m.  Held to: the typical
    pair identical, nine in ten within 1e-4 m, every pair within 2.5e-4 m; where the path was the same, the oracle's counts."""
    pairs, kw = bench_pairs
    out = capi.align_batch(ctx, capi.default_params(cost_mode=0, **kw), [p["cs"] for p in pairs], [p["ct"] for p in pairs], max_in_flight=32)
    dts = []
    for p, r in zip(pairs, out):
        dt, dR = _pose_err(r["T"], p["ro"]["T"], oracle)
        dts.append(dt)
        assert r["status"] == 0 and dt <= Q_P90_T and dR <= TOL_R, (p["seed"], dt, dR)
        if dt <= 1e-6:   # the same path through every line search
            assert r["iterations"] == p["ro"]["iterations"] and r["n_corr_last"] == p["ro"]["n_corr_last"], p["seed"]
    dts = np.array(dts)
    print("cost_mode 0, %d bench pairs vs oracle: |dt| median %.2e p90 %.2e max %.2e, %d identical" % (len(dts), np.median(dts), np.quantile(dts, 0.9), dts.max(), int((dts == 0).sum())))
    assert np.median(dts) <= 1e-6 and np.quantile(dts, 0.9) <= TOL_T, (np.median(dts), np.quantile(dts, 0.9))


@pytest.mark.parametrize("cost_mode", [0, 1])
def test_bench_pairs_full_size_vs_oracle(ctx, capi, oracle, bench_pairs, cost_mode):
    """>= 4 of the bench's own 100 032-point pairs, forced 20 iterations, odometry parameters: pose, fitness and the
    per-iteration trace (T, n_corr, f_end) of lh_gicp_align_batch / lh_gicp_align against oracle.gicp_align
    (gicp.hpp:445-568)."""
    pairs, kw = bench_pairs
    pairs = pairs[:4]   # with their per-iteration traces, one at a time (the other tests take all of them as a batch)
    P = capi.default_params(cost_mode=cost_mode, **kw)
    batch = capi.align_batch(ctx, P, [p["cs"] for p in pairs], [p["ct"] for p in pairs], max_in_flight=4)
    dts, drs, iter_dts = [], [], []
    for p, rb in zip(pairs, batch):
        ro = p["ro"]
        g = capi.Gicp(ctx, P)
        g.set_source(p["cs"])
        g.set_target(p["ct"])
        r = g.align()   # the same alignment one at a time, for the trace
        assert r["status"] == 0 and (r["T"] == rb["T"]).all() and r["iterations"] == rb["iterations"]   # batch == single, bit for bit
        dt, dR = _pose_err(r["T"], ro["T"], oracle)
        fit = g.fitness()
        k = min(len(r["trace"]["n_corr"]), len(ro["trace"]["n_corr"]))
        dT_it = np.abs(r["trace"]["T"][:k] - ro["trace"]["T"][:k]).max(1)
        print("bench pair seed %d cost_mode %d: |dt| %.2e |dR| %.2e iters %d/%d fitness rel %.1e max per-iteration |dT| %.2e"
              % (p["seed"], cost_mode, dt, dR, r["iterations"], ro["iterations"], abs(fit - p["fo"]) / p["fo"], dT_it.max()))
        dts.append(dt)
        drs.append(dR)
        assert r["trace"]["n_corr"][0] == ro["trace"]["n_corr"][0]   # first sweep: identical inputs => identical correspondences
        # same correspondences, same start: the BFGS end cost of the first iteration (mode 1: its line search stops elsewhere on the same valley floor)
        assert abs(r["trace"]["f_end"][0] - ro["trace"]["f_end"][0]) <= (1e-9 if cost_mode == 0 else 5e-3) * abs(ro["trace"]["f_end"][0])
        if cost_mode == 0:   # reference arithmetic: SURVEY 8d's bar, and the whole trajectory
            assert dt <= TOL_T and dR <= TOL_R, (p["seed"], dt, dR)
            assert abs(fit - p["fo"]) <= 1e-4 * p["fo"]
            assert r["iterations"] == ro["iterations"] and r["n_corr_last"] == ro["n_corr_last"]
            assert (r["trace"]["n_corr"][:k] == ro["trace"]["n_corr"][:k]).all()
            assert dT_it.max() <= TOL_T
            assert np.allclose(r["trace"]["f_end"][:k], ro["trace"]["f_end"][:k], rtol=1e-6)
        else:                # moment model: inside the reference's own noise floor at this configuration
            assert dt <= max(TOL_T, FLOOR_T_MAX) and dR <= max(TOL_R, FLOOR_R_MAX), (p["seed"], dt, dR)
            assert abs(fit - p["fo"]) <= 2e-3 * p["fo"]
            iter_dts.append(float(dT_it.max()))   # held below to the per-iteration quantile bars of the 32-pair test (measured up to 7.6e-3 mid-way; the reference's two builds: 3.4e-3)
            # the cost at the end of each solve: the same valley (mid-way the two line searches stop at different heights: 2 %
            # measured), the same floor at the end
            assert np.allclose(r["trace"]["f_end"][:k], ro["trace"]["f_end"][:k], rtol=5e-2)
            assert abs(r["trace"]["f_end"][-1] - ro["trace"]["f_end"][-1]) <= 2e-3 * ro["trace"]["f_end"][-1]
        # and the simulated motion is recovered
        Tm = oracle.T_to_mat(r["T"])
        assert np.abs(Tm[:3, 3] - p["delta"][:3, 3]).max() < 0.02
    if cost_mode == 1:
        assert np.median(dts) <= max(TOL_T, FLOOR_T_TYPICAL), dts
        # north_star's "per-iteration results match": the quantile bars proper (median <= 1.5e-3, p90 <= 1e-2 of the largest |dT| of any iteration
        # of a pair) are asserted over all 32 pairs in test_bench_pairs_device_loop_32_in_flight_vs_oracle; four pairs have no quantiles, so every
        # one of them is held to the p90 bar (measured here: 1.7e-3, 4.5e-4, 2.8e-3, 7.6e-3; the former order-of-magnitude check was 2e-2)
        assert max(iter_dts) <= Q_ITER_P90, iter_dts


def test_full_size_properties_100k(ctx, capi, oracle):
    # BASELINE config 2 size: size-independent properties instead of an oracle run
    src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10)
    cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
    cs.normals_knn(20)
    ct.normals_knn(20)
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    g = capi.Gicp(ctx, P)
    g.set_source(cs)
    g.set_target(ct)
    r = g.align()
    assert r["status"] == 0 and r["iterations"] >= 10
    Tm = oracle.T_to_mat(r["T"])
    assert np.abs(Tm[:3, 3] - delta[:3, 3]).max() < 0.02 and np.abs(Tm[:3, :3] - delta[:3, :3]).max() < 2e-3
    # idempotence: aligning the aligned cloud gives ~identity; cost is non-increasing over outer iterations
    f = r["trace"]["f_end"]
    assert f[-1] <= f[0] * 1.0001
    aligned = cs.transform(r["T"], with_normals=True)
    g.set_source(aligned)
    r2 = g.align(want_trace=False)
    T2 = oracle.T_to_mat(r2["T"])
    assert np.abs(T2[:3, 3]).max() < 5e-3 and np.abs(T2[:3, :3] - np.eye(3)).max() < 5e-4
    # NN of a cloud against itself is the identity map with d2 = 0 (no duplicate points in a noisy scan)
    idx, d2 = ct.nn1(ct)
    assert (d2 == 0).all() and (idx == np.arange(len(ct))).mean() > 0.9999


def test_ragged_batch_sizes_and_slot_reuse(ctx, capi, oracle):
    # pairs of very different sizes in one batch, fewer slots than pairs (slots are recycled), odd counts that do not fill
    # an XCD group of 8 jobs: every result must equal the one-at-a-time alignment bit for bit
    specs = [(8, 120), (16, 700), (4, 97), (32, 640), (16, 333), (8, 1000), (2, 64), (24, 500), (16, 128), (12, 901), (6, 251)]
    pairs = []
    for i, (rings, az) in enumerate(specs):
        src, tgt, _ = synth.scan_pair(n_rings=rings, n_az=az, scale=1.0, noise=0.01, seed=500 + 3 * i)
        pairs.append((src, tgt))
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    S, T, singles = [], [], []
    for src, tgt in pairs:
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        cs.normals_knn(10)
        ct.normals_knn(10)
        S.append(cs)
        T.append(ct)
        g = capi.Gicp(ctx, P)
        g.set_source(cs)
        g.set_target(ct)
        singles.append(g.align(want_trace=False))
    for in_flight in (3, 8, 16, 32):
        out = capi.align_batch(ctx, P, S, T, max_in_flight=in_flight)
        for k, (a, b) in enumerate(zip(out, singles)):
            assert a["status"] == b["status"] and a["iterations"] == b["iterations"], (in_flight, k)
            assert (a["T"] == b["T"]).all(), (in_flight, k)
    # and the oracle agrees: a tiny 388-point member in the reference-arithmetic mode (its alignment is too loosely
    # constrained for the noise-floor argument), two larger members in the default mode
    for k in (2, 3, 5):
        dl_s, dl_t = S[k].download(), T[k].download()
        ro = oracle.gicp_align(oracle.xyz4(pairs[k][0]), oracle.nrm4(np.stack([dl_s["normal_x"], dl_s["normal_y"], dl_s["normal_z"]], 1)),
                               oracle.xyz4(pairs[k][1]), oracle.nrm4(np.stack([dl_t["normal_x"], dl_t["normal_y"], dl_t["normal_z"]], 1)),
                               oracle.default_params(num_threads=4, max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3))
        if k == 2:
            g0 = capi.Gicp(ctx, capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, cost_mode=0))
            g0.set_source(S[k])
            g0.set_target(T[k])
            dt, dR = _pose_err(g0.align(want_trace=False)["T"], ro["T"], oracle)
            assert dt < TOL_T and dR < TOL_R, (k, dt, dR)
        else:
            dt, dR = _pose_err(singles[k]["T"], ro["T"], oracle)
            assert dt < TOL_T1 and dR < TOL_R1, (k, dt, dR)


def test_empty_and_invalid_inputs(ctx, capi):
    with pytest.raises(capi.LocusHipError):
        capi.Cloud(ctx, np.zeros((0, 3), np.float32))  # empty cloud: LH_EINVAL like pcl::Registration::initCompute
    g = capi.Gicp(ctx, capi.default_params())
    with pytest.raises(capi.LocusHipError):
        g.align()  # no source / target set
    with pytest.raises(capi.LocusHipError):
        g.fitness()
    assert capi.align_batch(ctx, capi.default_params(), [], []) == []


@pytest.mark.parametrize("cost_mode,solver", [(0, 0), (1, 1), (1, 2)])
@pytest.mark.parametrize("bad", ["nan", "inf"])
def test_non_finite_source_point_is_the_no_neighbour_failure(ctx, capi, oracle, cost_mode, solver, bad):
    """gicp.hpp:471-478, 504-506: a query for which searchForNeighbors finds nothing sets `failure`, and computeTransformation
    returns before the solve -- final_transformation_ stays the identity pcl::Registration::align reset it to, converged_ false.
    A non-finite source point is such a query (pcl::KdTreeFLANN: isValid).  Status LH_ENO_NN, in every loop flavour (reference
    arithmetic on the host, moment model on the host, moment model in k_solve), first sweep or a later one, alone or in a batch."""
    src, ns, tgt, nt, _ = _pair_with_normals(oracle, 91, rings=16, az=300)
    kw = dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    P = capi.default_params(cost_mode=cost_mode, solver=solver, **kw)
    bad_src = src.copy()
    bad_src[1234, 1] = np.nan if bad == "nan" else np.inf
    g = capi.Gicp(ctx, P)
    g.set_source(capi.make_pointf(bad_src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    r = g.align()
    ro = oracle.gicp_align(oracle.xyz4(bad_src), ns, oracle.xyz4(tgt), nt, oracle.default_params(num_threads=4, **kw))
    assert ro["status"] == oracle.LO_ENO_NN and ro["converged"] == 0 and ro["iterations"] == 0
    assert (oracle.T_to_mat(ro["T"]) == np.eye(4)).all()
    assert r["status"] == capi.LH_ENO_NN and r["converged"] == 0 and r["iterations"] == 0, r
    assert (oracle.T_to_mat(r["T"]) == np.eye(4)).all()
    # inside a batch: only that pair fails, its neighbours are untouched (and equal their one-at-a-time results)
    good = capi.Gicp(ctx, P)
    good.set_source(capi.make_pointf(src, ns))
    good.set_target(capi.make_pointf(tgt, nt))
    rg = good.align(want_trace=False)
    assert rg["status"] == 0
    S = [capi.Cloud(ctx, capi.make_pointf(src if k != 3 else bad_src, ns)) for k in range(9)]
    T = [capi.Cloud(ctx, capi.make_pointf(tgt, nt)) for _ in range(9)]
    out = capi.align_batch(ctx, P, S, T, max_in_flight=9)
    for k, o in enumerate(out):
        if k == 3:
            assert o["status"] == capi.LH_ENO_NN and o["converged"] == 0 and (oracle.T_to_mat(o["T"]) == np.eye(4)).all()
        else:
            assert o["status"] == 0 and (o["T"] == rg["T"]).all() and o["iterations"] == rg["iterations"]
    # a point that only becomes non-finite... cannot: T is finite.  But a target cloud of non-finite points gives every query no neighbour
    nan_tgt = np.full_like(tgt[:64], np.nan)
    g2 = capi.Gicp(ctx, P)
    g2.set_source(capi.make_pointf(src, ns))
    g2.set_target(capi.make_pointf(nan_tgt, nt[:64]))
    assert g2.align(want_trace=False)["status"] == capi.LH_ENO_NN


def test_reused_output_cloud_drops_its_stale_index(ctx, capi, oracle):
    """lh_gicp_align_batch_out overwrites a caller-supplied aligned[i]: an index (or k-NN covariances) built on its OLD coordinates
    must not survive, or a later search would walk a stale tree (round-2 advisor finding)."""
    pairs = [_pair_with_normals(oracle, 95 + i, rings=16, az=260) for i in range(2)]
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    S = [capi.Cloud(ctx, capi.make_pointf(p[0], p[1])) for p in pairs]
    T = [capi.Cloud(ctx, capi.make_pointf(p[2], p[3])) for p in pairs]
    _, A = capi.align_batch_out(ctx, P, S, T, max_in_flight=2)
    q = capi.Cloud(ctx, pairs[0][2][:500])
    A[0].nn1(q)                                   # builds an index on the first result
    A[0].cov_knn(10)                              # ... and k-NN covariances
    guesses = np.stack([oracle.mat_to_T(synth.pose_matrix(0.4, -0.3, 0.1, 0, 0, 0.2).astype(np.float32))] * 2)
    res2, A2 = capi.align_batch_out(ctx, P, S, T, guesses=guesses, max_in_flight=2, aligned=A)
    assert A2[0] is A[0]
    d = A[0].download()
    now = np.stack([d["x"], d["y"], d["z"]], 1)
    idx, d2 = A[0].nn1(q)                         # must search the NEW coordinates
    io, do = oracle.nn1_brute(oracle.xyz4(now), oracle.xyz4(pairs[0][2][:500]))
    assert (idx == io).all() and (d2 == do).all()


def test_bench_pairs_production_stopping_32_pairs(ctx, capi, oracle, bench_pairs):
    """The same 32 full-size bench pairs under the rule LOCUS actually stops by (tf_eps 1e-3, rotation_epsilon 2e-3; gicp.hpp:566,
    point_cloud_odometry/config/parameters.yaml:12), batched through the device-driven loop, against the oracle under the same rule:
    cost_mode 1 held to the quantiles of profiles/r04_parity_distributions.json (tighter than the distance between the reference's own two
    builds in every quantile), cost_mode 0 (reference arithmetic) to the pose and the iteration count."""
    pairs, _ = bench_pairs
    S, T = [p["cs"] for p in pairs], [p["ct"] for p in pairs]
    for mode in (1, 0):
        res = capi.align_batch(ctx, capi.default_params(cost_mode=mode, **PRODUCTION_KW), S, T, max_in_flight=32)
        dts, drs, its, fits = [], [], [], []
        g = capi.Gicp(ctx, capi.default_params(cost_mode=mode, **PRODUCTION_KW))
        for p, r in zip(pairs, res):
            ro = p["ro_prod"]
            assert r["status"] == 0 and ro["status"] == 0 and r["converged"] == 1 and ro["converged"] == 1
            dt, dR = _pose_err(r["T"], ro["T"], oracle)
            dts.append(dt)
            drs.append(dR)
            its.append(r["iterations"] - ro["iterations"])
            g.set_source(p["cs"])
            g.set_target(p["ct"])
            r1 = g.align(want_trace=False)
            assert (r1["T"] == r["T"]).all() and r1["iterations"] == r["iterations"], p["seed"]   # one at a time == batched, under this rule too
            fits.append(abs(g.fitness() - p["fo_prod"]) / p["fo_prod"])
        dts, drs, its, fits = np.array(dts), np.array(drs), np.array(its), np.array(fits)
        print("production stopping, cost_mode %d, %d bench pairs vs oracle: |dt| median %.2e p90 %.2e max %.2e | |dR| max %.2e | fitness rel median %.2e p90 %.2e | "
              "iteration count differs on %d pairs (max %d)" % (mode, len(pairs), np.median(dts), np.quantile(dts, 0.9), dts.max(), drs.max(), np.median(fits),
                                                                 np.quantile(fits, 0.9), int((its != 0).sum()), int(np.abs(its).max())))
        if mode == 1:
            assert np.median(dts) <= PROD_Q_MEDIAN_T and np.quantile(dts, 0.9) <= PROD_Q_P90_T and dts.max() <= PROD_HARD_T, dts
            assert drs.max() <= PROD_HARD_R
            assert np.median(fits) <= 5e-4 and np.quantile(fits, 0.9) <= 3e-3 and fits.max() <= 2e-2, fits
            assert np.abs(its).max() <= 3 and (its == 0).mean() >= 0.6, its
        else:
            assert np.median(dts) <= 1e-6 and np.quantile(dts, 0.9) <= 3e-4 and dts.max() <= 1e-3, dts
            assert drs.max() <= 1e-4 and (its == 0).all(), (drs.max(), its)
            assert np.median(fits) <= 1e-6 and fits.max() <= 1e-3, fits


def test_stream_of_raw_scans_normals_batch_then_align_stream(ctx, capi, oracle):
    """the streaming path of bench.py's stream_with_normals: raw scans -> ONE batched index build + ONE k-NN launch (lh_normals_knn_batch)
    -> lh_gicp_align_stream, which keeps the indices the normal filter built.  Bit-identical to the scan-at-a-time path (normals per
    cloud, lh_gicp_align_batch rebuilding every target), and the first pair against the oracle."""
    poses = [synth.pose_matrix(0.15 * i, 0.05 * i, 0.0, 0.0, 0.0, 0.02 * i) for i in range(6)]
    pts = [synth.scan(p, 16, 400, (-15.0, 15.0), 1.0, 0.01, seed=40 + i) for i, p in enumerate(poses)]
    A = [capi.Cloud(ctx, p) for p in pts]
    B = [capi.Cloud(ctx, p) for p in pts]
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    capi.normals_knn_batch(A, 20)
    ra = capi.align_stream(ctx, P, A, max_in_flight=8)
    for b in B:
        b.normals_knn(20)
        b.drop_index()
    rb = capi.align_batch(ctx, P, B[1:], B[:-1], max_in_flight=8)
    for x, y in zip(ra, rb):
        assert x["status"] == 0 and (np.asarray(x["T"]) == np.asarray(y["T"])).all() and x["iterations"] == y["iterations"]
    d0, d1 = A[0].download(), A[1].download()
    xyz = lambda d: oracle.xyz4(np.stack([d["x"], d["y"], d["z"]], 1))
    nrm = lambda d: oracle.nrm4(np.stack([d["normal_x"], d["normal_y"], d["normal_z"]], 1))
    po = oracle.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    ro = oracle.gicp_align(xyz(d1), nrm(d1), xyz(d0), nrm(d0), po)
    Tg, To = oracle.T_to_mat(ra[0]["T"]), oracle.T_to_mat(ro["T"])
    assert np.abs(Tg[:3, 3] - To[:3, 3]).max() < 2e-3 and np.abs(Tg[:3, :3] - To[:3, :3]).max() < 2.5e-3   # the stopping scale (include/locus_hip.h)


def test_align_stream_right_behind_the_normal_filter_many_groups_no_synchronise(ctx, capi, oracle):
    """Calls on a context are issued in order (include/locus_hip.h): lh_normals_knn_batch returns with its index build and k-NN launch still
    queued on the context's stream, and lh_gicp_align_stream with >= 32 pairs in flight spreads its groups over SIDE streams that read those
    trees and normals.  The scheduler forks its side streams from the primary stream (fork_side_streams), so the result is the one of the
    synchronised sequence, bit for bit -- 100 k-point scans make the filter's launch long enough for a missing dependency to show."""
    n = 49
    poses = [synth.pose_matrix(0.12 * i, 0.02 * i, 0.0, 0.0, 0.0, 0.01 * i) for i in range(n)]
    pts = [synth.scan(p, 32, 1200, (-25.0, 15.0), 2.0, 0.02, seed=700 + i) for i, p in enumerate(poses)]
    P = capi.default_params(max_iterations=6, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
    A = [capi.Cloud(ctx, p) for p in pts]
    capi.normals_knn_batch(A, 20)
    ctx.synchronize()
    ref = capi.align_stream(ctx, P, A, max_in_flight=48)      # everything the filter left is complete: the reference result
    ctx.synchronize()
    for rep in range(3):
        B = [capi.Cloud(ctx, p) for p in pts]
        capi.normals_knn_batch(B, 20)                         # NO synchronise: the k-NN launch is still running when the groups start
        got = capi.align_stream(ctx, P, B, max_in_flight=48)  # 48 in flight: two or more groups, all but the first on side streams
        for x, y in zip(got, ref):
            assert x["status"] == 0 and (np.asarray(x["T"]) == np.asarray(y["T"])).all() and x["iterations"] == y["iterations"]
        for b in B:
            b.close()
    for a in A:
        a.close()
