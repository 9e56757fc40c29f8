"""lh_gicp_measurement_update (PointCloudLocalization::MeasurementUpdate's device work in one call, PointCloudLocalization.cc:305-336, 398-421,
469-486, 694-750): the call returns the bits of the parity-tested pieces it replaces -- lh_gicp_align, lh_cloud_transform with normals,
lh_nn1, lh_p2plane_information, lh_icp_covariance -- and agrees with the CPU oracle's restatement of the same chain."""
import numpy as np
import pytest

from locus_amd import synth

pytestmark = pytest.mark.gpu


def _pair(capi, ctx, rings, az, seed):
    src, tgt, delta = synth.scan_pair(n_rings=rings, n_az=az, scale=1.0, noise=0.01, seed=seed)
    ns, nt = ctx.normals_knn(src, 20), ctx.normals_knn(tgt, 20)
    return src, ns, tgt, nt, delta


@pytest.mark.parametrize("rings,az", [(16, 200), (32, 900)])   # LOCUS's ~3 000 points, and a 28 800-point scan
def test_measurement_update_equals_the_pieces_bit_for_bit(ctx, capi, oracle, rings, az):
    src, ns, tgt, nt, _ = _pair(capi, ctx, rings, az, 21)
    P = capi.default_params(max_iterations=20, max_inner_iterations=50, corr_dist=0.2 if az > 500 else 1.0, transformation_epsilon=1e-5)
    pf_s, pf_t = capi.make_pointf(src, ns), capi.make_pointf(tgt, nt)
    g = capi.Gicp(ctx, P)
    g.set_source(pf_s)
    g.set_target(pf_t)
    aligned = pf_s.copy()
    m = g.measurement_update(aligned_out=aligned, icp_max_covariance=0.01)
    # the pieces, one at a time
    g2 = capi.Gicp(ctx, P)
    g2.set_source(pf_s)
    g2.set_target(pf_t)
    r = g2.align(want_trace=False)
    assert np.array_equal(np.asarray(r["T"]), np.asarray(m["T"])) and r["iterations"] == m["iterations"] and r["status"] == m["status"] == 0
    cs, ct = capi.Cloud(ctx, pf_s), capi.Cloud(ctx, pf_t)
    al = cs.transform(np.asarray(r["T"], np.float32), with_normals=True).download()
    for f in ("x", "y", "z", "normal_x", "normal_y", "normal_z"):
        assert np.array_equal(al[f], aligned[f]), f
    for f in ("intensity", "curvature"):   # every other field of the input is copied (pcl::transformPointCloudWithNormals)
        assert np.array_equal(aligned[f], pf_s[f])
    idx, _ = g2.nn1(al)
    assert np.array_equal(idx, m["corr"])
    Ap = ctx.p2plane_information(cs, ct, idx)
    assert np.array_equal(Ap, m["Ap"])
    ok, cov, cond = capi.icp_covariance(Ap, 0.01)
    assert ok == m["covariance_ok"] and np.array_equal(cov, m["covariance"]) and cond == m["condition_number"]
    # the device-resident form leaves the same aligned query in HBM
    md = g.measurement_update(aligned_cloud=True)
    ad = md["aligned"].download()
    for f in ("x", "y", "z", "normal_x", "normal_y", "normal_z", "intensity", "curvature"):
        assert np.array_equal(ad[f], aligned[f]), f
    assert np.array_equal(md["Ap"], m["Ap"]) and np.array_equal(md["corr"], m["corr"])
    # ... and the oracle's chain on the same transform
    q4 = oracle.transform(oracle.xyz4(src), np.asarray(r["T"], np.float32))
    io, _ = oracle.nn1_brute(oracle.xyz4(tgt), q4)
    assert np.array_equal(io, idx)
    Ao = oracle.p2plane_Ap(oracle.normalize_cloud(oracle.xyz4(src)), oracle.nrm4(nt), io)
    assert np.allclose(m["Ap"], Ao, rtol=2e-4, atol=2e-4 * np.abs(Ao).max())


def test_measurement_update_without_information_and_without_outputs(ctx, capi):
    src, ns, tgt, nt, delta = _pair(capi, ctx, 16, 300, 5)
    g = capi.Gicp(ctx, capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3))
    g.set_source(capi.make_pointf(src, ns))
    g.set_target(capi.make_pointf(tgt, nt))
    m = g.measurement_update(want_information=False, want_corr=False)
    assert m["status"] == 0 and "Ap" not in m and m["corr"] is None
    T = np.asarray(m["T"], np.float64).reshape(4, 4).T
    assert np.abs(T[:3, 3] - delta[:3, 3]).max() < 0.05
    # a reference without normals cannot give Ap: LH_EINVAL, nothing half-done
    g.set_target(capi.Cloud(ctx, np.ascontiguousarray(tgt, np.float32)))
    with pytest.raises(capi.LocusHipError):
        g.measurement_update(want_information=True)
