"""CPU suite (-m "not gpu"): pins the oracle against every known-answer test the reference's own test-suite holds
for the hot path and against its own brute-force variants."""
import numpy as np
import pytest

from locus_amd import synth


def test_eig3_kat(oracle):
    # doEigenDecomp3x3 KAT (point_cloud_localization/test/test_point_cloud_localization.cpp:550-561); lower triangle is used
    data = np.array([[-2.0, -4.0, 2.0], [-2.0, 1.0, 2.0], [4.0, 2.0, 5.0]])
    ev, V = oracle.eig_sym(data)
    assert np.allclose(ev, [-5.0, 2.0, 7.0], atol=1e-4)
    S = np.tril(data) + np.tril(data, -1).T
    assert np.allclose(S @ V, V * ev, atol=1e-10)


def test_kdtree_equals_bruteforce(oracle):
    rng = np.random.default_rng(0)
    P = oracle.xyz4(np.concatenate([rng.normal(size=(3000, 3)), np.repeat(rng.normal(size=(200, 3)), 3, 0)]))  # with duplicates
    Q = oracle.xyz4(np.concatenate([rng.normal(size=(800, 3)), P[:200, :3]]))
    t = oracle.Tree(P)
    i1, d1 = t.nn1(Q)
    i2, d2 = oracle.nn1_brute(P, Q)
    assert (i1 == i2).all() and (d1 == d2).all()
    k1, e1 = t.knn(Q, 20)
    k2, e2 = oracle.knn_brute(P, Q, 20)
    assert (k1 == k2).all() and (e1 == e2).all()
    i4, d4 = t.nn1(Q, threads=4)
    assert (i4 == i1).all()


def test_empty_and_tiny_clouds(oracle):
    P = oracle.xyz4(np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]]))
    t = oracle.Tree(P)
    idx, d2 = t.knn(oracle.xyz4(np.array([[0.1, 0, 0]])), 5)
    assert list(idx[0][:3]) == [0, 1, 2] and (idx[0][3:] == -1).all()
    p = oracle.default_params()
    r = oracle.gicp_align(P, oracle.nrm4(np.zeros((3, 3))), P, oracle.nrm4(np.zeros((3, 3))), p)
    assert r["status"] == -4  # < 4 correspondences: NotEnoughPointsException (gicp.hpp:225)
    assert r["converged"] == 0


def test_gradient_matches_finite_differences(oracle):
    rng = np.random.default_rng(1)
    n = 400
    src = oracle.xyz4(rng.normal(size=(n, 3)) * 3)
    tgt = oracle.xyz4(src[:, :3] + rng.normal(size=(n, 3)) * 0.05)
    A = rng.normal(size=(n, 3, 3))
    M = np.einsum("nij,nkj->nik", A, A) + np.eye(3)
    idx = np.arange(n, dtype=np.int32)
    x = np.array([0.1, -0.2, 0.05, 0.02, -0.03, 0.04])
    f, g, s = oracle.cost_fdf(src, tgt, idx, idx, M, x)
    for i in range(6):
        h = 1e-3
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        num = (oracle.cost_fdf(src, tgt, idx, idx, M, xp)[0] - oracle.cost_fdf(src, tgt, idx, idx, M, xm)[0]) / (2 * h)
        assert abs(num - g[i]) < 2e-3 * max(1.0, abs(g[i]))


def test_apply_state_is_zyx(oracle):
    x = np.array([0.3, -0.2, 0.1, 0.05, -0.07, 0.2])
    T = oracle.T_to_mat(oracle.apply_state(x))
    R = synth.rot_zyx(x[3], x[4], x[5])
    assert np.allclose(T[:3, :3], R, atol=2e-7)
    assert np.allclose(T[:3, 3], x[:3], atol=1e-7)


def test_cov_from_normals_and_knn_agree_on_a_plane(oracle):
    pts, nrm = synth.plane_grid(12, 12, 0.1)
    P = oracle.xyz4(pts)
    c1 = oracle.cov_from_normals(oracle.nrm4(nrm))
    c2 = oracle.cov_knn(P, k=9)
    expect = np.diag([1.0, 1.0, 1e-3])
    assert np.allclose(c1, expect, atol=1e-12)
    assert np.allclose(c2, expect, atol=1e-9)
    z = oracle.cov_from_normals(oracle.nrm4(np.zeros((2, 3))))
    assert np.allclose(z, np.eye(3))  # zero normal => identity (documented assumption)


def test_hollow_cube_translation_kat(oracle):
    # UpdateEstimateUpdateICP (point_cloud_odometry/test/test_point_cloud_odometry.cpp:280-305), odometry yaml params
    cube = synth.hollow_cube()
    tree = oracle.Tree(oracle.xyz4(cube))
    nrm = oracle.normals_knn(oracle.xyz4(cube), k=5, tree=tree)
    moved = cube + np.array([0.05, 0.05, 0.0], np.float32)
    p = oracle.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, num_threads=2)
    # reference cloud = first scan (with normals); query = translated cloud whose normals are all zero
    r = oracle.gicp_align(oracle.xyz4(moved), oracle.nrm4(np.zeros_like(moved)), oracle.xyz4(cube), nrm, p)
    assert r["status"] == 0 and r["converged"] == 1
    T = oracle.T_to_mat(r["T"])
    Tinv = np.linalg.inv(T)
    assert abs(Tinv[0, 3] - 0.05) < 1e-2 and abs(Tinv[1, 3] - 0.05) < 1e-2 and abs(Tinv[2, 3]) < 1e-2
    fit = oracle.fitness(oracle.xyz4(moved), r["T"], tree)
    assert fit < 0.1


def test_plane_Ap_kat(oracle):
    # ComputeAp KAT (test_point_cloud_localization.cpp:288-394): 10x10 plane, normal (0,0,1), identity correspondences
    pts, nrm = synth.plane_grid(10, 10, 0.1)
    qn = oracle.normalize_cloud(oracle.xyz4(pts))
    Ap = oracle.p2plane_Ap(qn, oracle.nrm4(nrm), np.arange(100))
    assert abs(Ap[0, 0] - 56.7753) < 1e-4 and abs(Ap[1, 1] - 56.7753) < 1e-4 and abs(Ap[5, 5] - 100.0) < 1e-4  # epsilion = 1e-4 (test_point_cloud_localization.cpp:24,337-339)
    # hand-rolled sum (same KAT compares against an explicit loop)
    a = qn[:, :3].astype(np.float64)
    n = np.tile([0.0, 0, 1], (100, 1))
    H = np.concatenate([np.cross(a, n), n], 1)
    assert np.allclose(Ap, H.T @ H, atol=1e-9)
    ev, _ = oracle.eig_sym(Ap)
    assert np.allclose(ev, [0, 0, 0, 56.7753, 56.7753, 100], atol=1e-4)  # observability eigenvalues (:479-507), the reference's epsilion


def test_icp_covariance_kats(oracle):
    # test_point_cloud_localization.cpp:398-476
    pts, nrm = synth.plane_grid(10, 10, 0.1)
    qn = oracle.normalize_cloud(oracle.xyz4(pts))
    Ap = oracle.p2plane_Ap(oracle.normalize_cloud(qn), oracle.nrm4(nrm), np.arange(100))
    ok, cov, cond = oracle.icp_covariance(Ap, 0.01)
    assert np.allclose(cov, np.eye(6) * 0.01, atol=1e-4)  # single plane: clamp
    # three orthogonal planes + yaw-30deg tf
    T1 = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    T2 = np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1.0]])
    n4 = oracle.nrm4(nrm)
    p1, n1 = oracle.transform(qn, oracle.mat_to_T(T1), n4)
    p2, n2 = oracle.transform(qn, oracle.mat_to_T(T2), n4)
    allp = np.concatenate([qn, p1, p2])
    alln = np.concatenate([n4, n1, n2])
    tf = np.array([[0.866, -0.5, 0, 0.001], [0.5, 0.866, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1.0]])
    off, _ = oracle.transform(allp, oracle.mat_to_T(tf), alln)
    Ap = oracle.p2plane_Ap(oracle.normalize_cloud(off), alln, np.arange(300))
    ok, cov, cond = oracle.icp_covariance(Ap, 0.01)
    assert ok
    assert np.allclose(cov, np.eye(6) * 1e-6, atol=1e-4)


def test_frame_transform_kat(oracle):
    # TransformPointsToFixedFrame KAT (test_point_cloud_localization.cpp:243-276): z +/- 3
    pts, nrm = synth.plane_grid(4, 4, 0.1)
    T = np.eye(4)
    T[2, 3] = 3.0
    out, on = oracle.transform(oracle.xyz4(pts), oracle.mat_to_T(T), oracle.nrm4(nrm))
    assert np.allclose(out[:, 2], 3.0) and np.allclose(on[:, :3], nrm)
    back = oracle.transform(out, oracle.mat_to_T(np.linalg.inv(T)))
    assert np.allclose(back[:, :3], pts, atol=1e-6)


def test_voxel_grid_semantics(oracle):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-2, 2, size=(5000, 3)).astype(np.float32)
    xyzi = np.concatenate([pts, rng.uniform(0, 100, size=(5000, 1)).astype(np.float32)], 1)
    out = oracle.voxel_grid(xyzi, 0.25)
    # every output is the centroid of the points that fall in its cell; cells are visited in ascending index
    inv = np.float32(1.0) / np.float32(0.25)
    ijk = np.floor(pts * inv).astype(np.int64)
    ijk -= np.floor(pts.min(0) * inv).astype(np.int64)
    div = ijk.max(0) + 1
    lin = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq = np.unique(lin)
    assert out.shape[0] == uniq.shape[0]
    for r in (0, len(uniq) // 2, len(uniq) - 1):
        sel = xyzi[lin == uniq[r]]
        assert np.allclose(out[r], sel.mean(0), rtol=1e-5, atol=1e-5)
    # pass-through limits drop points, overflow guard returns None
    out2 = oracle.voxel_grid(xyzi, 0.25, limit_axis=2, lo=-1.0, hi=1.0)
    assert out2[:, 2].min() >= -1.0 and out2[:, 2].max() <= 1.0 and out2.shape[0] < out.shape[0]
    assert oracle.voxel_grid(xyzi, 1e-4) is None


def test_voxel_grid_pointf_semantics(oracle):
    """pcl::VoxelGrid<PointXYZINormal> (PointCloudFilter.cc:119-124): the same voxels in the same order as the xyzi flavour,
    and every field averaged -- x, y, z, intensity, curvature are means, the normal is the NORMALISED sum"""
    rng = np.random.default_rng(4)
    pts = rng.uniform(-2, 2, size=(4000, 3)).astype(np.float32)
    xyzi = np.concatenate([pts, rng.uniform(0, 100, size=(4000, 1)).astype(np.float32)], 1)
    nrm = rng.normal(size=(4000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm4 = np.concatenate([nrm, rng.uniform(0, 0.3, size=(4000, 1)).astype(np.float32)], 1)
    out, out_n = oracle.voxel_grid_pointf(xyzi, nrm4, 0.25)
    ref = oracle.voxel_grid(xyzi, 0.25)
    assert (out == ref).all()                      # xyz + intensity centroids: bit-identical to the PCLPointCloud2 flavour
    inv = np.float32(1.0) / np.float32(0.25)
    ijk = np.floor(pts * inv).astype(np.int64) - np.floor(pts.min(0) * inv).astype(np.int64)
    div = ijk.max(0) + 1
    lin = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq = np.unique(lin)
    for r in (0, len(uniq) // 3, len(uniq) - 1):
        sel = nrm4[lin == uniq[r]]
        s = sel[:, :3].astype(np.float64).sum(0)
        assert np.allclose(out_n[r, :3], s / np.linalg.norm(s), atol=1e-5)
        assert np.isclose(out_n[r, 3], sel[:, 3].mean(), rtol=1e-5)
    assert np.allclose(np.linalg.norm(out_n[:, :3], axis=1), 1.0, atol=1e-5)
    # opposite normals in one voxel cancel: the zero sum stays zero (Eigen normalized())
    two = np.array([[0.01, 0.01, 0.01, 1.0], [0.02, 0.02, 0.02, 3.0]], np.float32)
    nn = np.array([[0, 0, 1, 0.1], [0, 0, -1, 0.3]], np.float32)
    o, on = oracle.voxel_grid_pointf(two, nn, 1.0)
    assert o.shape[0] == 1 and (on[0, :3] == 0).all() and np.isclose(on[0, 3], 0.2) and np.isclose(o[0, 3], 2.0)


def test_normals_on_a_plane_and_sphere(oracle):
    pts, nrm = synth.plane_grid(15, 15, 0.1, z=2.0)
    out = oracle.normals_knn(oracle.xyz4(pts), k=10)
    # viewpoint (0,0,0) is below the plane z=2 -> normals flipped to -z
    assert np.allclose(np.abs(out[:, 2]), 1.0, atol=1e-3) and (out[:, 2] < 0).all()
    assert (out[:, 3] < 1e-3).all()


def test_garage_fixture_thread_invariance(oracle, garage):
    # multithreaded_gicp/test/test_same_output_different_num_threads.cpp: identical Matrix4f for 1..8 threads
    q, r = garage
    assert q.shape == (811, 4) and r.shape == (8112, 4)
    base = None
    for th in (1, 2, 4, 8):
        p = oracle.default_params(transformation_epsilon=1e-10, corr_dist=0.2, max_iterations=20, max_inner_iterations=50,
                                  recompute_source_cov=1, recompute_target_cov=1, num_threads=th)
        res = oracle.gicp_align(oracle.xyz4(q), None, oracle.xyz4(r), None, p)
        assert res["status"] == 0
        if base is None:
            base = res
        else:
            assert (res["T"] == base["T"]).all()
    tree = oracle.Tree(oracle.xyz4(r))
    fit = oracle.fitness(oracle.xyz4(q), base["T"], tree)
    fit0 = oracle.fitness(oracle.xyz4(q), oracle.mat_to_T(np.eye(4)), tree)
    assert fit <= fit0 * 1.05  # alignment does not make the fit worse


def test_reference_float_noise_floor(oracle):
    """The reference evaluates T*p in float inside the cost functor (gicp.hpp:382).  Building the SAME source with or
    without FMA contraction changes those roundings, the BFGS trajectory, and the result: this measures how far the
    reference's own output moves, which bounds what any bit-different but equally valid evaluation (our cost_mode 1)
    can be held to."""
    L = oracle.lib()
    worst_t, worst_r = 0.0, 0.0
    for seed in (21, 22, 23):
        src, tgt, _ = synth.scan_pair(n_rings=16, n_az=600, scale=1.0, noise=0.01, seed=seed)
        ns = oracle.normals_knn(oracle.xyz4(src), 20, threads=4)
        nt = oracle.normals_knn(oracle.xyz4(tgt), 20, threads=4)
        res = []
        for v in (0, 1):
            L.lo_set_cost_variant(v)
            res.append(oracle.gicp_align(oracle.xyz4(src), ns, oracle.xyz4(tgt), nt,
                                         oracle.default_params(num_threads=4, max_iterations=20, corr_dist=1.0,
                                                               transformation_epsilon=1e-3)))
        L.lo_set_cost_variant(0)
        A, B = oracle.T_to_mat(res[0]["T"]), oracle.T_to_mat(res[1]["T"])
        worst_t = max(worst_t, np.abs(A[:3, 3] - B[:3, 3]).max())
        worst_r = max(worst_r, np.abs(A[:3, :3] - B[:3, :3]).max())
    print("FMA vs non-FMA reference builds differ by up to |dt| = %.2e m, |dR| = %.2e" % (worst_t, worst_r))
    assert 1e-6 < worst_t < 5e-3  # noise floor is real (not bit-stable) and of order 1e-4..1e-3 m


def test_radius_normals_plane_and_sparse(oracle):
    """radius flavour of the normal filter (normal_computation.cc:71-74): plane normal (0,0,1) like the Ap KATs' planes
    (test_point_cloud_localization.cpp:296); isolated points get NaN (then dropped, normal_computation.cc:52-56)"""
    pts, _ = synth.plane_grid(20, 20, 0.1)
    pts = pts + np.array([0.5, 0.3, -2.0], np.float32)   # below the sensor: viewpoint flip -> +z
    pts = np.concatenate([pts, np.array([[50, 50, 50], [50.2, 50, 50]], np.float32)], 0)
    out = oracle.normals_radius(oracle.xyz4(pts), 0.3, threads=2)
    assert np.isnan(out[-2:]).all()
    assert np.allclose(out[:-2, :3], [0, 0, 1], atol=1e-3)
    # k-NN flavour agrees where the 0.3 m ball holds the same evidence (interior of the grid)
    knn = oracle.normals_knn(oracle.xyz4(pts[:-2]), 20)
    assert np.allclose(knn[:, :3], out[:-2, :3], atol=2e-3)


def test_map_oracle_first_point_per_voxel(oracle):
    """local-map restatement (the mapper is un-vendored, semantics from Locus.cc:464-465 / 531-538 + BLAM):
    one point per octree voxel, the first one offered; Refresh is an inclusive box crop that frees the voxels it drops"""
    m = oracle.MapOracle(0.5)
    pts = np.array([[0.1, 0.1, 0.1], [0.4, 0.2, 0.3], [0.6, 0.1, 0.1], [-0.1, 0.1, 0.1], [np.nan, 0, 0], [0.1, 0.1, 0.1]], np.float32)
    assert m.insert(pts) == [0, 2, 3]            # 1 and 5 share voxel (0,0,0) with 0; 3 is voxel (-1,0,0): floor, not truncation
    assert m.insert(pts) == []
    assert oracle.map_voxel(np.array([-0.1, 0.1, 0.1], np.float32), 0.5) == (-1, 0, 0)
    kept = m.refresh([0.0, 0.0, 0.0], 0.5)
    assert kept == [0, 2] and len(m.pts) == 2    # |x| <= 0.5 keeps 0.1 and -0.1; 0.6 goes
    assert m.insert(pts) == [2]                  # its voxel is free again


def test_ndt_oracle_self_consistency(oracle):
    """NDT restatement (oracle/locus_oracle_ndt.c; the reference holds no NDT test: "parity unpinned").  Checks that do not need
    the reference: the analytic gradient against central differences of the score (the score is only piecewise smooth -- cells
    enter and leave the radius-search neighbourhood -- hence a loose bound on the dominant components), float and double
    hessian paths agree, the SVD solve, pose <-> matrix round trip, and registration of known motions."""
    delta = synth.pose_matrix(0.04, -0.03, 0.01, 0.002, -0.001, 0.006)
    src, tgt, delta = synth.scan_pair(n_rings=16, n_az=600, scale=2.0, noise=0.02, seed=10, delta=delta)
    s4, t4 = oracle.xyz4(src), oracle.xyz4(tgt)
    P = oracle.ndt_default_params(transformation_epsilon=1e-3, max_iterations=30)
    g = oracle.NdtGrid(t4, P)
    mean, icov, cen = g.cells()
    assert len(mean) > 100 and np.allclose(mean, cen[:, :3], atol=1e-4)            # double mean vs float centroid of the same points
    assert (np.linalg.eigvalsh(0.5 * (icov + icov.transpose(0, 2, 1))) > 0).all()  # inflated covariances are positive definite
    p0 = np.array([0.05, -0.02, 0.01, 0.004, -0.003, 0.01])
    s, gr, H = g.derivatives(s4, p0)
    fd = np.zeros(6)
    for k in range(6):
        a, b = p0.copy(), p0.copy()
        a[k] += 1e-3
        b[k] -= 1e-3
        fd[k] = (g.derivatives(s4, a, False)[0] - g.derivatives(s4, b, False)[0]) / 2e-3
    big = np.abs(gr) > 0.2 * np.abs(gr).max()
    assert np.allclose(fd[big], gr[big], rtol=0.1), (fd, gr)
    Hd = g.hessian(s4, p0)
    assert np.abs(H - Hd).max() < 1e-5 * np.abs(Hd).max() and np.abs(Hd - Hd.T).max() < 1e-12 * np.abs(Hd).max()
    rng = np.random.default_rng(0)
    A, b = rng.normal(size=(6, 6)), rng.normal(size=6)
    assert np.allclose(oracle.svd_solve6(A, b), np.linalg.solve(A, b), atol=1e-12)
    A[:, 5] = A[:, 4]                                                              # rank-deficient: minimum-norm least squares
    assert np.allclose(oracle.svd_solve6(A, b), np.linalg.pinv(A) @ b, atol=1e-9)
    for pose in (p0, np.array([0.3, -0.2, 0.1, -0.05, 0.02, 2.5])):                # eulerAngles(0,1,2) may pick the flipped branch:
        T = oracle.ndt_pose_to_matrix(pose)                                        # compare matrices, not angles
        assert np.allclose(oracle.ndt_pose_to_matrix(oracle.ndt_matrix_to_pose(T)), T, atol=2e-6)
    srcd, tgtd, _ = synth.scan_pair(n_rings=32, n_az=900, scale=2.0, noise=0.02, seed=10, delta=delta)   # 1-m voxels need dense walls
    r = oracle.ndt_align(oracle.xyz4(srcd), oracle.xyz4(tgtd), P)
    T = oracle.T_to_mat(r["T"])
    assert r["converged"] == 1 and np.abs(T[:3, 3] - delta[:3, 3]).max() < 0.02 and np.abs(T[:3, :3] - delta[:3, :3]).max() < 2e-3
    r = oracle.ndt_align(t4, t4, P, oracle.mat_to_T(synth.pose_matrix(0.05, -0.03, 0.0, 0, 0, 0.005).astype(np.float32)))
    assert np.abs(oracle.T_to_mat(r["T"]) - np.eye(4)).max() < 1e-3                # a cloud against itself, from a small offset


def test_config1_plumbing_chain_on_oracle(oracle):
    """BASELINE configs[0]: single ~5 k-pt synthetic pair, voxel grid + GICP on the CPU path.  The chain
    runs on the restatement, recovers the simulated motion, and reproduces the committed golden pose
    (tests/golden/config1_chain.json, generated by tests/golden/make_config1_golden.py)."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_config1_golden", os.path.join(here, "make_config1_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    c = mod.chain(threads=4)
    gold = json.load(open(os.path.join(here, "config1_chain.json")))
    r = c["result"]
    assert [c["src"].shape[0], c["tgt"].shape[0]] == gold["n_raw"] == [28800, 28800]
    assert [c["vs"].shape[0], c["vt"].shape[0]] == gold["n_voxel"]
    assert 3000 < c["vs"].shape[0] < 8000 and 3000 < c["vt"].shape[0] < 8000          # "N_out ~ 5 k"
    assert np.isclose(c["vs"][:, :3].astype(np.float64).sum(), gold["voxel_checksum"][0], rtol=0, atol=1e-6)
    assert r["status"] == 0 and r["converged"] == gold["converged"] == 1 and r["iterations"] == gold["iterations"]
    assert np.abs(np.array(r["T"], np.float64) - np.array(gold["T_colmajor"])).max() < 1e-6
    Tm = oracle.T_to_mat(r["T"])
    assert np.abs(Tm[:3, 3] - c["delta"][:3, 3]).max() < 0.01 and np.abs(Tm[:3, :3] - c["delta"][:3, :3]).max() < 2e-3
    # thread-count invariance on this chain too (the reference's only GICP test asserts exactly this)
    c1 = mod.chain(threads=1)
    assert (np.array(c1["result"]["T"]) == np.array(r["T"])).all()



def test_oracle_no_neighbour_failure_path(oracle):
    """gicp.hpp:471-478, 504-506: a query without a nearest neighbour (here: a non-finite source point) sets `failure`;
    computeTransformation returns before the solve, so final_transformation_ is still the identity align() reset it to."""
    from locus_amd import synth
    src, tgt, _ = synth.scan_pair(n_rings=8, n_az=200, scale=1.0, noise=0.01, seed=5)
    ns = oracle.normals_knn(oracle.xyz4(src), 10, threads=2)
    nt = oracle.normals_knn(oracle.xyz4(tgt), 10, threads=2)
    P = oracle.default_params(num_threads=2, max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    ok = oracle.gicp_align(oracle.xyz4(src), ns, oracle.xyz4(tgt), nt, P)
    assert ok["status"] == oracle.LO_OK and ok["iterations"] >= 1
    for bad in (np.nan, np.inf, -np.inf):
        s2 = src.copy()
        s2[37, 2] = bad
        r = oracle.gicp_align(oracle.xyz4(s2), ns, oracle.xyz4(tgt), nt, P)
        assert r["status"] == oracle.LO_ENO_NN and r["converged"] == 0 and r["iterations"] == 0
        assert (oracle.T_to_mat(r["T"]) == np.eye(4)).all()
    # the kd-tree and the exhaustive search agree that such a query has no neighbour
    q = oracle.xyz4(np.array([[np.nan, 0, 0], [np.inf, 0, 0], [0.1, 0.2, 0.3]], np.float32))
    it, _ = oracle.Tree(oracle.xyz4(tgt)).nn1(q, threads=1)
    ib, _ = oracle.nn1_brute(oracle.xyz4(tgt), q)
    assert list(it[:2]) == [-1, -1] and list(ib[:2]) == [-1, -1] and it[2] == ib[2] >= 0
