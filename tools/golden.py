#!/usr/bin/env python3
"""SECOND, independent restatement of the GICP hot path in numpy + scipy -- written from the reference's source
(/root/reference/multithreaded_gicp/include/multithreaded_gicp/gicp.hpp, cited per function), sharing NO code with oracle/locus_oracle.c:
neighbours from scipy.spatial.cKDTree instead of the oracle's own kd-tree, 3x3 inverses / SVDs from LAPACK instead of hand-written
cofactors / Jacobi sweeps, and -- the part the C oracle restates "as recalled" -- NO pcl::BFGS at all: every outer iteration minimises
the frozen-correspondence cost with scipy.optimize (BFGS with an analytic gradient, tight tolerance), i.e. it goes to the minimiser the
reference's inner loop is heading for instead of imitating where that loop stops.

It writes tests/golden/second_restatement.npz (inputs + expected outputs); tests/test_second_restatement.py then holds the C oracle to
it on the CPU:  first-sweep correspondences and Mahalanobis matrices, k-NN covariances, f and g of the cost functor, the
per-outer-iteration minimisers and the final pose -- on the reference's own garage scans (multithreaded_gicp/test/*.pcd, parameters of
test_same_output_different_num_threads.cpp:31-36) and on BASELINE configs[0]'s 5 k-point pair.  Runs in the build container
(numpy + scipy only; nothing of it ships): python tools/golden.py"""
import os
import sys

import numpy as np
from scipy.optimize import minimize
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from locus_amd import synth  # noqa: E402  (the seeded scene generator: numpy only)

F32 = np.float32


def read_pcd_xyz(path):
    """PCD v0.7, binary, fields x y z intensity (the garage fixtures)"""
    raw = open(path, "rb").read()
    head, _, body = raw.partition(b"DATA binary\n")
    h = {l.split()[0]: l.split()[1:] for l in head.decode().splitlines() if l and not l.startswith("#")}
    n = int(h["POINTS"][0])
    stride = sum(int(s) * int(c) for s, c in zip(h["SIZE"], h["COUNT"]))
    a = np.frombuffer(body[: n * stride], np.uint8).reshape(n, stride)
    return np.ascontiguousarray(a[:, :12]).view(F32).reshape(n, 3).copy()


def voxel_grid_xyz(p, leaf):
    """pcl::VoxelGrid::applyFilter on x, y, z (custom_voxel_grid.cc:82-85): bounding box, floor(p / leaf) cells, centroid per cell, cells in
    ascending (z, y, x)-major index order.  float32 sums in input order like PCL's accumulation."""
    inv = F32(1.0) / F32(leaf)
    mn, mx = p.min(0), p.max(0)
    lo = np.floor(mn * inv).astype(np.int64)
    hi = np.floor(mx * inv).astype(np.int64)
    div = hi - lo + 1
    ijk = np.floor(p * inv).astype(np.int64) - lo
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    s = 0
    si = idx[order]
    while s < len(order):
        e = s
        while e < len(order) and si[e] == si[s]:
            e += 1
        acc = np.zeros(3, F32)
        for j in order[s:e]:
            acc = acc + p[j]
        out.append(acc / F32(e - s))
        s = e
    return np.asarray(out, F32)


def knn_covariances(p, k, eps):
    """computeCovariances, k-NN branch (gicp.hpp:85-154): double moments over the k nearest (the point itself included), SVD, singular
    values (1, 1, eps)"""
    tree = cKDTree(p.astype(np.float64))
    _, nn = tree.query(p.astype(np.float64), k=k)
    out = np.empty((len(p), 3, 3))
    for i in range(len(p)):
        q = p[nn[i]].astype(np.float64)
        mean = q.sum(0) / k
        cov = (q.T @ q) / k - np.outer(mean, mean)
        cov = np.tril(cov) + np.tril(cov, -1).T          # the reference fills the lower triangle and mirrors it (:127-133)
        U, _, _ = np.linalg.svd(cov)
        out[i] = U[:, 0:1] @ U[:, 0:1].T + U[:, 1:2] @ U[:, 1:2].T + eps * (U[:, 2:3] @ U[:, 2:3].T)
    return out, nn


def xform_f32(T, p):
    """Eigen Matrix4f * Vector4f(x, y, z, 1) in float32 (gicp.hpp:469, :382): column-by-column accumulation ((m0 x + m1 y) + m2 z) + m3"""
    T = T.astype(F32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    return np.stack([((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3] for r in range(3)], 1).astype(F32)


def first_sweep(src, tgt, T, guess, C1, C2, corr_dist):
    """the NN + Mahalanobis loop of computeTransformation (gicp.hpp:446-498)"""
    q = xform_f32(T, src)
    tree = cKDTree(tgt.astype(np.float64))
    _, nn = tree.query(q.astype(np.float64), k=1)
    d = q - tgt[nn]
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(F32)     # FLANN L2_Simple, float
    ok = d2.astype(np.float64) < corr_dist * corr_dist                                   # nn_dists[0] < dist_threshold (:483)
    R = (T.astype(np.float64) @ guess.astype(np.float64))[:3, :3]                        # transform_R (:450-460)
    M = np.zeros((len(src), 3, 3))
    for i in np.nonzero(ok)[0]:
        M[i] = np.linalg.inv(R @ C1[i] @ R.T + C2[nn[i]])                                # :487-493
    return nn.astype(np.int32), d2, ok, M


def euler_zyx(x):
    """applyState's rotation (gicp.hpp:619-634): Rz(x5) Ry(x4) Rx(x3)"""
    cph, sph, cth, sth, cps, sps = np.cos(x[3]), np.sin(x[3]), np.cos(x[4]), np.sin(x[4]), np.cos(x[5]), np.sin(x[5])
    Rz = np.array([[cps, -sps, 0], [sps, cps, 0], [0, 0, 1.0]])
    Ry = np.array([[cth, 0, sth], [0, 1, 0], [-sth, 0, cth]])
    Rx = np.array([[1, 0, 0], [0, cph, -sph], [0, sph, cph]])
    return Rz @ Ry @ Rx


def apply_state(x):
    """transformation_matrix = identity; applyState(transformation_matrix, x) (gicp.hpp:277-278): a float matrix"""
    T = np.eye(4)
    T[:3, :3] = euler_zyx(x)
    T[:3, 3] = x[:3]
    return T.astype(F32)


def state_of(T):
    """gicp.hpp:233-239"""
    return np.array([T[0, 3], T[1, 3], T[2, 3], np.arctan2(T[2, 1], T[2, 2]), np.arcsin(-T[2, 0]), np.arctan2(T[1, 0], T[0, 0])], np.float64)


def d_euler(x):
    """computeRDerivative's three matrices (gicp.hpp:160-214), here by differentiating Rz Ry Rx instead of copying its tables"""
    cph, sph, cth, sth, cps, sps = np.cos(x[3]), np.sin(x[3]), np.cos(x[4]), np.sin(x[4]), np.cos(x[5]), np.sin(x[5])
    Rz = np.array([[cps, -sps, 0], [sps, cps, 0], [0, 0, 1.0]])
    Ry = np.array([[cth, 0, sth], [0, 1, 0], [-sth, 0, cth]])
    Rx = np.array([[1, 0, 0], [0, cph, -sph], [0, sph, cph]])
    dRz = np.array([[-sps, -cps, 0], [cps, -sps, 0], [0, 0, 0.0]])
    dRy = np.array([[-sth, 0, cth], [0, 0, 0], [-cth, 0, -sth]])
    dRx = np.array([[0, 0, 0], [0, -sph, -cph], [0, cph, -sph]])
    return Rz @ Ry @ dRx, Rz @ dRy @ Rx, dRz @ Ry @ Rx


def cost_fdf(x, src, tgt, nn, ok, M, float_product=True):
    """OptimizationFunctorWithIndices::fdf (gicp.hpp:362-402), base_transformation_ = identity.  float_product: T * p evaluated in float32
    like the reference (:382); False = in double (the smooth function the optimiser is given)"""
    ii = np.nonzero(ok)[0]
    p, t, Mi = src[ii], tgt[nn[ii]], M[ii]
    if float_product:
        res = (xform_f32(apply_state(x), p) - t).astype(np.float64)     # pp[0] - p_tgt[0]: two floats, a FLOAT subtraction (:384)
    else:
        res = p.astype(np.float64) @ euler_zyx(x).T + x[:3] - t.astype(np.float64)
    temp = np.einsum("nij,nj->ni", Mi, res)
    m = len(ii)
    f = float(np.einsum("ni,ni->", res, temp)) / m
    g = np.zeros(6)
    g[:3] = temp.sum(0) * (2.0 / m)
    Rm = (p.astype(np.float64).T @ temp) * (2.0 / m)                    # R += p_src3 * temp^T (:394-397)
    dphi, dth, dpsi = d_euler(x)
    g[3], g[4], g[5] = (dphi * Rm.T).sum(), (dth * Rm.T).sum(), (dpsi * Rm.T).sum()   # matricesInnerProd = trace(dR * R) (gicp.h)
    return f, g


def gicp(src, tgt, C1, C2, corr_dist, max_iterations, tf_eps, rot_eps=2e-3):
    """computeTransformation (gicp.hpp:406-617) with every inner solve taken to the minimiser of the frozen-correspondence cost"""
    guess = np.eye(4, dtype=F32)
    T = np.eye(4, dtype=F32)
    prev = T.copy()
    first = None
    trace = []
    for it in range(max_iterations):
        nn, d2, ok, M = first_sweep(src, tgt, T, guess, C1, C2, corr_dist)
        if first is None:
            first = (nn, d2, ok, M)
        prev = T.copy()
        x0 = state_of(T.astype(np.float64))
        r = minimize(lambda x: cost_fdf(x, src, tgt, nn, ok, M, float_product=False), x0, jac=True, method="BFGS", options={"gtol": 1e-10, "maxiter": 500})
        T = apply_state(r.x)
        trace.append({"x": r.x.copy(), "n_corr": int(ok.sum()), "f": float(r.fun)})
        delta = 0.0
        for k in range(4):
            for l in range(4):
                ratio = 1.0 / rot_eps if (k < 3 and l < 3) else 1.0 / tf_eps
                delta = max(delta, ratio * abs(float(F32(prev[k, l]) - F32(T[k, l]))))
        if delta < 1:
            break
    return T, first, trace


def case(name, src, tgt, k, eps, corr_dist, max_iterations, tf_eps):
    C1, nn1 = knn_covariances(src, k, eps)
    C2, nn2 = knn_covariances(tgt, k, eps)
    T, (nn, d2, ok, M), trace = gicp(src, tgt, C1, C2, corr_dist, max_iterations, tf_eps)
    x_probe = np.array([0.013, -0.021, 0.006, 0.0012, -0.0023, 0.0031])
    f0, g0 = cost_fdf(np.zeros(6), src, tgt, nn, ok, M)
    f1, g1 = cost_fdf(x_probe, src, tgt, nn, ok, M)
    print("%s: %d -> %d points, %d correspondences in the first sweep, %d outer iterations, final t = %s" % (name, len(src), len(tgt), int(ok.sum()), len(trace), T[:3, 3]))
    return {name + "_src": src, name + "_tgt": tgt, name + "_params": np.array([k, eps, corr_dist, max_iterations, tf_eps]),
            name + "_cov_src": C1, name + "_cov_tgt_sample": C2[:: max(1, len(C2) // 512)], name + "_knn_src": nn1.astype(np.int32),
            name + "_nn": nn, name + "_d2": d2, name + "_ok": ok, name + "_maha": M,
            name + "_x_probe": x_probe, name + "_f0": f0, name + "_g0": g0, name + "_f1": f1, name + "_g1": g1,
            name + "_iter_x": np.array([t["x"] for t in trace]), name + "_iter_ncorr": np.array([t["n_corr"] for t in trace]), name + "_T": T}


def main():
    gold = os.path.join(ROOT, "tests", "golden")
    out = {}
    q, r = read_pcd_xyz(os.path.join(gold, "query_82_garage.pcd")), read_pcd_xyz(os.path.join(gold, "reference_82_garage.pcd"))
    # test_same_output_different_num_threads.cpp:31-36: tf_eps 1e-10, corr_dist 0.2, 50 iterations, k-NN covariance branch (PointXYZI clouds)
    out.update(case("garage", q, r, 20, 1e-3, 0.2, 50, 1e-10))
    # 1 721 of the reference scan's 8 112 points repeat another point's coordinates (up to 75 times): which of several coincident points is "the"
    # neighbour is FLANN's unpinned tie rule, and it decides which (degenerate) covariance enters.  The same pair with every coordinate kept
    # once (first occurrence, input order) has no such freedom.
    _, keep = np.unique(r, axis=0, return_index=True)
    out.update(case("garage_unique", q, r[np.sort(keep)], 20, 1e-3, 0.2, 50, 1e-10))
    src, tgt, _ = synth.config1_pair()
    vs, vt = voxel_grid_xyz(src, 0.25), voxel_grid_xyz(tgt, 0.25)
    # BASELINE configs[0]: the 5 k-point plumbing pair, odometry parameters (parameters.yaml: corr_dist 1.0, tf_eps 1e-3, 20 iterations), recompute branch
    out.update(case("config1", vs, vt, 20, 1e-3, 1.0, 20, 1e-3))
    path = os.environ.get("GOLDEN_OUT") or os.path.join(gold, "second_restatement.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
