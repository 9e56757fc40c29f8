"""GPU suite: the device-driven loop of cost_mode 1 (k_solve: BFGS solve + convergence test on the GPU, the host only enqueues
iterations) against the host-driven loop (lh_gicp_params.solver = 1: one sync per outer iteration, the solve on the host; solver = 2 forces the device loop, 0 picks by batch size).  Both
run the same templates (lh_bfgs.hpp) on the same 74 moments with the same elementary functions (lh_math.hpp), so every number
must agree BIT FOR BIT: transforms, iteration counts, the per-iteration trace."""
import numpy as np
import pytest

from locus_amd import synth

pytestmark = pytest.mark.gpu


def _clouds(ctx, capi, seed, rings=16, az=500, k=20, scale=1.0):
    src, tgt, delta = synth.scan_pair(n_rings=rings, n_az=az, scale=scale, noise=0.01, seed=seed)
    cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
    cs.normals_knn(k)
    ct.normals_knn(k)
    return cs, ct, delta


@pytest.mark.parametrize("kw", [dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3),                       # production stopping
                                dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12),  # forced 20 (bench)
                                dict(max_iterations=7, max_inner_iterations=50, corr_dist=0.5, transformation_epsilon=1e-5),    # localization-like, odd count
                                dict(max_iterations=1, corr_dist=1.0)])
def test_device_loop_equals_host_loop_bit_for_bit(ctx, capi, oracle, kw):
    for seed in (301, 302, 303):
        cs, ct, delta = _clouds(ctx, capi, seed)
        res = {}
        for solver in (2, 1):
            g = capi.Gicp(ctx, capi.default_params(cost_mode=1, solver=solver, **kw))
            g.set_source(cs)
            g.set_target(ct)
            guess = oracle.mat_to_T(synth.pose_matrix(0.03, -0.02, 0.0, 0, 0, 0.004).astype(np.float32)) if seed == 303 else None
            res[solver] = g.align(guess=guess)
        a, b = res[2], res[1]
        assert a["status"] == b["status"] == 0
        assert (a["T"] == b["T"]).all() and a["iterations"] == b["iterations"] and a["converged"] == b["converged"]
        assert a["n_corr_last"] == b["n_corr_last"] and a["cost_passes"] == b["cost_passes"]
        ta, tb = a["trace"], b["trace"]
        assert len(ta["n_corr"]) == len(tb["n_corr"]) == a["iterations"]
        for key in ("T", "n_corr", "n_passes", "n_inner", "f_end", "delta"):
            assert (np.asarray(ta[key]) == np.asarray(tb[key])).all(), key
        Tm = oracle.T_to_mat(a["T"])
        if kw["max_iterations"] >= 7:
            assert np.abs(Tm[:3, 3] - delta[:3, 3]).max() < 0.05


def test_device_loop_batches_mixed_sizes_and_early_finishers(ctx, capi, oracle):
    # pairs of different sizes that converge after different numbers of iterations, fewer slots than pairs: the device loop
    # retires and admits pairs between rounds; every result equals the host-driven loop's and the one-at-a-time alignment's
    specs = [(8, 200), (16, 640), (32, 500), (4, 150), (16, 333), (24, 410), (12, 777), (6, 90), (16, 256)]
    S, T = [], []
    for i, (rings, az) in enumerate(specs):
        cs, ct, _ = _clouds(ctx, capi, 400 + 5 * i, rings, az, k=10)
        S.append(cs)
        T.append(ct)
    for kw in (dict(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3), dict(max_iterations=9, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)):
        host = capi.align_batch(ctx, capi.default_params(solver=1, **kw), S, T, max_in_flight=4)
        for in_flight in (1, 3, 4, 9, 32):
            dev = capi.align_batch(ctx, capi.default_params(solver=2, **kw), S, T, max_in_flight=in_flight)
            for k, (a, b) in enumerate(zip(dev, host)):
                assert a["status"] == b["status"] and a["iterations"] == b["iterations"] and a["converged"] == b["converged"], (in_flight, k)
                assert (a["T"] == b["T"]).all() and a["cost_passes"] == b["cost_passes"], (in_flight, k)
        its = [r["iterations"] for r in host]
        if kw["transformation_epsilon"] > 1e-6:
            assert min(its) < max(its) or max(its) < 20   # the pairs really finish at different times


def test_device_loop_error_paths(ctx, capi, oracle):
    three = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    for solver in (2, 1, 0):
        g = capi.Gicp(ctx, capi.default_params(solver=solver))
        g.set_source(capi.make_pointf(three, np.zeros_like(three)))
        g.set_target(capi.make_pointf(three, np.zeros_like(three)))
        res = g.align(raise_on_error=False)
        assert res["status"] == capi.LH_ETOO_FEW_CORR and res["converged"] == 0 and res["iterations"] == 0   # gicp.hpp:225, 542-547
        assert np.allclose(oracle.T_to_mat(res["T"]), np.eye(4))
    # a batch in which one pair fails: the others are untouched
    cs, ct, _ = _clouds(ctx, capi, 77, 8, 300, k=10)
    bad = capi.Cloud(ctx, capi.make_pointf(three, np.zeros_like(three)))
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, solver=2)
    out = capi.align_batch(ctx, P, [cs, bad, cs], [ct, bad, ct], max_in_flight=3)
    assert out[1]["status"] == capi.LH_ETOO_FEW_CORR and out[0]["status"] == 0 and (out[0]["T"] == out[2]["T"]).all()
