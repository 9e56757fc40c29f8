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
