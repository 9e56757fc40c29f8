"""ctypes binding of the CPU oracle (oracle/liblocus_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package (locus_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblocus_oracle.so")
LO_MAX_TRACE = 256
LO_OK, LO_EINVAL, LO_ENOMEM, LO_ETOO_FEW, LO_ESOLVER, LO_ENO_NN = 0, -1, -2, -4, -5, -6   # locus_oracle.h


class LoParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("max_inner_iterations", C.c_int),
        ("corr_dist", C.c_double),
        ("transformation_epsilon", C.c_double),
        ("rotation_epsilon", C.c_double),
        ("gicp_epsilon", C.c_double),
        ("k_correspondences", C.c_int),
        ("recompute_source_cov", C.c_int),
        ("recompute_target_cov", C.c_int),
        ("num_threads", C.c_int),
        ("parallel_cost", C.c_int),
    ]


class LoTrace(C.Structure):
    _fields_ = [
        ("n_iters", C.c_int),
        ("T", (C.c_float * 16) * LO_MAX_TRACE),
        ("n_corr", C.c_int * LO_MAX_TRACE),
        ("n_passes", C.c_int * LO_MAX_TRACE),
        ("n_inner", C.c_int * LO_MAX_TRACE),
        ("f_end", C.c_double * LO_MAX_TRACE),
        ("delta", C.c_double * LO_MAX_TRACE),
    ]


class LoResult(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16),
        ("converged", C.c_int),
        ("iterations", C.c_int),
        ("n_corr_last", C.c_int),
        ("status", C.c_int),
        ("total_passes", C.c_long),
        ("t_index", C.c_double),
        ("t_cov", C.c_double),
        ("t_nn", C.c_double),
        ("t_opt", C.c_double),
        ("t_total", C.c_double),
    ]


def build(force=False):
    src = os.path.join(_HERE, "locus_oracle.c")
    if force or not os.path.exists(_LIB) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class LoNdtParams(C.Structure):
    _fields_ = [("resolution", C.c_float), ("step_size", C.c_double), ("outlier_ratio", C.c_double), ("transformation_epsilon", C.c_double),
                ("max_iterations", C.c_int), ("min_points_per_voxel", C.c_int), ("min_covar_eigvalue_mult", C.c_double), ("num_threads", C.c_int)]


class LoNdtResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("converged", C.c_int), ("iterations", C.c_int), ("evaluations", C.c_int), ("n_cells", C.c_int),
                ("trans_probability", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        L = _lib
        L.lo_tree_build.restype = C.c_void_p
        L.lo_tree_build.argtypes = [C.c_void_p, C.c_int]
        L.lo_tree_free.argtypes = [C.c_void_p]
        L.lo_fitness.restype = C.c_double
        L.lo_fitness.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_nn1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_nn1_brute.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.lo_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_knn_brute.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.lo_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_cov_from_normals.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        L.lo_cov_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int]
        L.lo_nn_mahalanobis.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_cost_fdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_apply_state.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_gicp_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(LoParams), C.c_void_p, C.POINTER(LoResult), C.POINTER(LoTrace),
                                    C.c_void_p]
        L.lo_normalize_cloud.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lo_p2plane_Ap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_icp_covariance.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.lo_eig_sym.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.lo_voxel_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int]
        L.lo_voxel_grid_pointf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_normals_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lo_normals_radius.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int]
        L.lo_ndt_default_params.argtypes = [C.POINTER(LoNdtParams)]
        L.lo_ndt_grid_build.argtypes = [C.c_void_p, C.c_int, C.POINTER(LoNdtParams)]
        L.lo_ndt_grid_build.restype = C.c_void_p
        L.lo_ndt_grid_free.argtypes = [C.c_void_p]
        L.lo_ndt_grid_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_ndt_pose_to_matrix.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_ndt_matrix_to_pose.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_ndt_derivatives.argtypes = [C.c_void_p, C.POINTER(LoNdtParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_ndt_derivatives.restype = C.c_double
        L.lo_ndt_hessian.argtypes = [C.c_void_p, C.POINTER(LoNdtParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.lo_svd_solve6.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_ndt_align.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(LoNdtParams), C.c_void_p, C.POINTER(LoNdtResult)]
        L.lo_read_pcd_xyzi.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
        L.lo_default_params.argtypes = [C.POINTER(LoParams)]
    return _lib


def _f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4, a.shape
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def xyz4(xyz):
    """(n,3|4) -> float32 (n,4) with w = 1 (pcl PointXYZ data[3] = 1)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    out = np.ones((xyz.shape[0], 4), dtype=np.float32)
    out[:, :3] = xyz[:, :3]
    return out


def nrm4(nrm):
    nrm = np.asarray(nrm, dtype=np.float32)
    out = np.zeros((nrm.shape[0], 4), dtype=np.float32)
    out[:, : nrm.shape[1]] = nrm
    return out


def default_params(**kw):
    p = LoParams()
    lib().lo_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


class Tree:
    def __init__(self, pts4):
        self.pts = _f4(pts4)
        self.h = lib().lo_tree_build(_p(self.pts), self.pts.shape[0])

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_tree_free(self.h)
            self.h = None

    def nn1(self, q4, threads=1):
        q4 = _f4(q4)
        idx = np.empty(q4.shape[0], np.int32)
        d2 = np.empty(q4.shape[0], np.float32)
        lib().lo_nn1(self.h, _p(q4), q4.shape[0], _p(idx), _p(d2), threads)
        return idx, d2

    def knn(self, q4, k, threads=1):
        q4 = _f4(q4)
        idx = np.empty((q4.shape[0], k), np.int32)
        d2 = np.empty((q4.shape[0], k), np.float32)
        lib().lo_knn(self.h, _p(q4), q4.shape[0], k, _p(idx), _p(d2), threads)
        return idx, d2


def nn1_brute(pts4, q4):
    pts4, q4 = _f4(pts4), _f4(q4)
    idx = np.empty(q4.shape[0], np.int32)
    d2 = np.empty(q4.shape[0], np.float32)
    lib().lo_nn1_brute(_p(pts4), pts4.shape[0], _p(q4), q4.shape[0], _p(idx), _p(d2))
    return idx, d2


def knn_brute(pts4, q4, k):
    pts4, q4 = _f4(pts4), _f4(q4)
    idx = np.empty((q4.shape[0], k), np.int32)
    d2 = np.empty((q4.shape[0], k), np.float32)
    lib().lo_knn_brute(_p(pts4), pts4.shape[0], _p(q4), q4.shape[0], k, _p(idx), _p(d2))
    return idx, d2


def transform(pts4, T_colmajor16, nrm=None):
    pts4 = _f4(pts4)
    T = np.ascontiguousarray(T_colmajor16, np.float32).reshape(16)
    out = np.empty_like(pts4)
    if nrm is not None:
        nrm = _f4(nrm)
        on = np.empty_like(nrm)
        lib().lo_transform(_p(pts4), _p(nrm), pts4.shape[0], _p(T), _p(out), _p(on))
        return out, on
    lib().lo_transform(_p(pts4), None, pts4.shape[0], _p(T), _p(out), None)
    return out


def cov_from_normals(nrm, eps=1e-3):
    nrm = _f4(nrm)
    cov = np.empty((nrm.shape[0], 3, 3), np.float64)
    lib().lo_cov_from_normals(_p(nrm), nrm.shape[0], eps, _p(cov))
    return cov


def cov_knn(pts4, k=20, eps=1e-3, threads=1, tree=None):
    pts4 = _f4(pts4)
    tree = tree or Tree(pts4)
    cov = np.empty((pts4.shape[0], 3, 3), np.float64)
    rc = lib().lo_cov_knn(_p(pts4), pts4.shape[0], tree.h, k, eps, _p(cov), threads)
    assert rc == 0, rc
    return cov


def nn_mahalanobis(out4, tree, cov_src, cov_tgt, T16, R9, corr_dist, threads=1):
    out4 = _f4(out4)
    n = out4.shape[0]
    cov_src = np.ascontiguousarray(cov_src, np.float64)
    cov_tgt = np.ascontiguousarray(cov_tgt, np.float64)
    T16 = np.ascontiguousarray(T16, np.float32).reshape(16)
    R9 = np.ascontiguousarray(R9, np.float64).reshape(9)
    idx = np.empty(n, np.int32)
    maha = np.zeros((n, 3, 3), np.float64)
    lib().lo_nn_mahalanobis(_p(out4), n, tree.h, _p(cov_src), _p(cov_tgt), _p(T16), _p(R9), corr_dist,
                            _p(idx), _p(maha), threads)
    return idx, maha


def cost_fdf(out4, tgt4, src_idx, tgt_idx, maha, x6):
    out4, tgt4 = _f4(out4), _f4(tgt4)
    src_idx = np.ascontiguousarray(src_idx, np.int32)
    tgt_idx = np.ascontiguousarray(tgt_idx, np.int32)
    maha = np.ascontiguousarray(maha, np.float64)
    x6 = np.ascontiguousarray(x6, np.float64)
    f = C.c_double()
    g = np.empty(6, np.float64)
    sums = np.empty(13, np.float64)
    lib().lo_cost_fdf(_p(out4), _p(tgt4), _p(src_idx), _p(tgt_idx), src_idx.shape[0], _p(maha), _p(x6),
                      C.byref(f), _p(g), _p(sums))
    return f.value, g, sums


def bfgs_trace(out4, tgt4, src_idx, tgt_idx, maha, x_start, max_inner=20, cap=256):
    """estimateRigidTransformationBFGS's solve (gicp.hpp:249-271 + the restated pcl::BFGS) from state x_start with its inner steps recorded:
    row 0 after minimizeInit, then one row per successful minimizeOneStep"""
    out4, tgt4 = _f4(out4), _f4(tgt4)
    src_idx = np.ascontiguousarray(src_idx, np.int32)
    tgt_idx = np.ascontiguousarray(tgt_idx, np.int32)
    maha = np.ascontiguousarray(maha, np.float64)
    x0 = np.ascontiguousarray(x_start, np.float64)
    xs, fs, gn, ev = np.zeros((cap, 6)), np.zeros(cap), np.zeros(cap), np.zeros(cap, np.int32)
    res = C.c_int()
    L = lib()
    L.lo_estimate_rigid_bfgs_trace.restype = C.c_int
    rows = L.lo_estimate_rigid_bfgs_trace(_p(out4), _p(tgt4), _p(src_idx), _p(tgt_idx), C.c_int(src_idx.shape[0]), _p(maha), C.c_int(max_inner), _p(x0),
                                          _p(xs), _p(fs), _p(gn), _p(ev), C.c_int(cap), C.byref(res))
    return {"x": xs[:rows], "f": fs[:rows], "gnorm": gn[:rows], "evals": ev[:rows], "result": res.value}


def apply_state(x6):
    x6 = np.ascontiguousarray(x6, np.float64)
    T = np.empty(16, np.float32)
    lib().lo_apply_state(_p(x6), _p(T))
    return T


def T_to_mat(T16):
    """column-major 16 floats -> (4,4) matrix"""
    return np.asarray(T16, dtype=np.float64).reshape(4, 4).T.copy()


def mat_to_T(M):
    return np.ascontiguousarray(np.asarray(M, np.float32).T).reshape(16)


def gicp_align(src4, src_n, tgt4, tgt_n, params, guess=None, want_trace=True, want_aligned=False):
    src4, tgt4 = _f4(src4), _f4(tgt4)
    src_n = _f4(src_n) if src_n is not None else None
    tgt_n = _f4(tgt_n) if tgt_n is not None else None
    res = LoResult()
    trace = LoTrace() if want_trace else None
    g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
    aligned = np.empty_like(src4) if want_aligned else None
    lib().lo_gicp_align(_p(src4), _p(src_n), src4.shape[0], _p(tgt4), _p(tgt_n), tgt4.shape[0], C.byref(params),
                        _p(g), C.byref(res), C.byref(trace) if trace is not None else None, _p(aligned))
    out = {
        "T": np.array(res.T[:], np.float32),
        "converged": res.converged,
        "iterations": res.iterations,
        "n_corr_last": res.n_corr_last,
        "status": res.status,
        "total_passes": res.total_passes,
        "t_index": res.t_index, "t_cov": res.t_cov, "t_nn": res.t_nn, "t_opt": res.t_opt, "t_total": res.t_total,
    }
    if trace is not None:
        k = trace.n_iters
        out["trace"] = {
            "T": np.array([trace.T[i][:] for i in range(k)], np.float32).reshape(k, 16),
            "n_corr": np.array(trace.n_corr[:k]),
            "n_passes": np.array(trace.n_passes[:k]),
            "n_inner": np.array(trace.n_inner[:k]),
            "f_end": np.array(trace.f_end[:k]),
            "delta": np.array(trace.delta[:k]),
        }
    if aligned is not None:
        out["aligned"] = aligned
    return out


def fitness(src4, T16, tree, threads=1):
    src4 = _f4(src4)
    T16 = np.ascontiguousarray(T16, np.float32).reshape(16)
    return lib().lo_fitness(_p(src4), src4.shape[0], _p(T16), tree.h, threads)


def normalize_cloud(pts4):
    pts4 = _f4(pts4)
    out = np.empty_like(pts4)
    lib().lo_normalize_cloud(_p(pts4), pts4.shape[0], _p(out))
    return out


def p2plane_Ap(qnorm4, ref_nrm4, corr):
    qnorm4, ref_nrm4 = _f4(qnorm4), _f4(ref_nrm4)
    corr = np.ascontiguousarray(corr, np.int64)
    Ap = np.empty((6, 6), np.float64)
    lib().lo_p2plane_Ap(_p(qnorm4), qnorm4.shape[0], _p(ref_nrm4), _p(corr), _p(Ap))
    return Ap


def icp_covariance(Ap, upper_bound=0.01):
    Ap = np.ascontiguousarray(Ap, np.float64)
    cov = np.empty((6, 6), np.float64)
    cond = C.c_double()
    ok = lib().lo_icp_covariance(_p(Ap), upper_bound, _p(cov), C.byref(cond))
    return bool(ok), cov, cond.value


def eig_sym(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    ev = np.empty(n, np.float64)
    V = np.empty((n, n), np.float64)
    lib().lo_eig_sym(_p(A), n, _p(ev), _p(V))
    return ev, V


def voxel_grid(xyzi, leaf, limit_axis=-1, lo=-np.inf, hi=np.inf):
    xyzi = _f4(xyzi)
    out = np.empty_like(xyzi)
    n = lib().lo_voxel_grid(_p(xyzi), xyzi.shape[0], leaf, limit_axis, float(max(lo, -3.0e38)), float(min(hi, 3.0e38)),
                            _p(out), xyzi.shape[0])
    if n < 0:
        return None
    return out[:n].copy()


def voxel_grid_pointf(xyzi, nrm4, leaf):
    """pcl::VoxelGrid<PointXYZINormal> (PointCloudFilter.cc:119-124): returns (xyzi centroids, (nx, ny, nz, curvature))"""
    xyzi, nrm4 = _f4(xyzi), _f4(nrm4)
    out, out_n = np.empty_like(xyzi), np.empty_like(nrm4)
    n = lib().lo_voxel_grid_pointf(_p(xyzi), _p(nrm4), xyzi.shape[0], leaf, _p(out), _p(out_n), xyzi.shape[0])
    if n < 0:
        return None
    return out[:n].copy(), out_n[:n].copy()


def normals_knn(pts4, k=20, threads=1, tree=None):
    pts4 = _f4(pts4)
    tree = tree or Tree(pts4)
    out = np.empty_like(pts4)
    lib().lo_normals_knn(_p(pts4), pts4.shape[0], tree.h, k, _p(out), threads)
    return out


def normals_radius(pts4, radius=0.3, threads=1, tree=None):
    pts4 = _f4(pts4)
    tree = tree or Tree(pts4)
    out = np.empty_like(pts4)
    lib().lo_normals_radius(_p(pts4), pts4.shape[0], tree.h, radius, _p(out), threads)
    return out


def read_pcd_xyzi(path, cap=1 << 22):
    buf = np.empty((cap, 4), np.float32)
    n = lib().lo_read_pcd_xyzi(path.encode(), _p(buf), cap)
    if n < 0:
        raise IOError("cannot read %s (%d)" % (path, n))
    return buf[:n].copy()


# ---- local map, restated sequentially.  point_cloud_mapper is un-vendored: "parity unpinned"; semantics from its
# BLAM lineage (InsertPoints: a point is added iff its octree voxel is still empty; Refresh: box crop around the pose).
def map_voxel(p, resolution):
    """voxel of a float32 point: floor(double(p) / resolution) per axis"""
    return tuple(int(np.floor(float(c) * (1.0 / float(resolution)))) for c in p[:3])


class MapOracle:
    """pure-Python restatement (small cases only): occupancy set + points in insertion order"""

    def __init__(self, resolution):
        self.res = float(resolution)
        self.occ = set()
        self.pts = []      # rows: index into the concatenation of everything ever offered (for order checks) is not kept; xyz only
        self.extra = []    # per-point payload (e.g. intensity)

    def insert(self, pts, payload=None):
        added = []
        for i, p in enumerate(np.asarray(pts, np.float32)):
            if not np.all(np.isfinite(p[:3])):
                continue
            v = map_voxel(p, self.res)
            if v in self.occ:
                continue
            self.occ.add(v)
            self.pts.append(p[:3].copy())
            self.extra.append(None if payload is None else payload[i])
            added.append(i)
        return added

    def refresh(self, center, half):
        c = np.asarray(center, np.float32)
        h = np.float32(half)
        keep = [k for k, p in enumerate(self.pts) if np.all(p >= c - h) and np.all(p <= c + h)]
        self.pts = [self.pts[k] for k in keep]
        self.extra = [self.extra[k] for k in keep]
        self.occ = {map_voxel(p, self.res) for p in self.pts}
        return keep


# ---- NDT -----------------------------------------------------------------------------------------------------
def ndt_default_params(**kw):
    p = LoNdtParams()
    lib().lo_ndt_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class NdtGrid:
    def __init__(self, tgt4, params):
        self.params = params
        tgt4 = _f4(tgt4)
        self.h = lib().lo_ndt_grid_build(_p(tgt4), tgt4.shape[0], C.byref(params))

    def __del__(self):
        try:
            lib().lo_ndt_grid_free(self.h)
        except Exception:
            pass

    def cells(self):
        n = lib().lo_ndt_grid_cells(self.h, None, None, None, 0)
        mean, icov, cen = np.empty((n, 3)), np.empty((n, 3, 3)), np.empty((n, 4), np.float32)
        lib().lo_ndt_grid_cells(self.h, _p(mean), _p(icov), _p(cen), n)
        return mean, icov, cen

    def derivatives(self, src4, p6, want_h=True):
        src4 = _f4(src4)
        p6 = np.ascontiguousarray(p6, np.float64)
        T = ndt_pose_to_matrix(p6)
        trans = transform(src4, T)
        g, H = np.empty(6), np.empty((6, 6))
        s = lib().lo_ndt_derivatives(self.h, C.byref(self.params), _p(src4), _p(trans), src4.shape[0], _p(p6), _p(g), _p(H), 1 if want_h else 0)
        return s, g, H

    def hessian(self, src4, p6):
        src4 = _f4(src4)
        p6 = np.ascontiguousarray(p6, np.float64)
        trans = transform(src4, ndt_pose_to_matrix(p6))
        H = np.empty((6, 6))
        lib().lo_ndt_hessian(self.h, C.byref(self.params), _p(src4), _p(trans), src4.shape[0], _p(p6), _p(H))
        return H


def ndt_pose_to_matrix(p6):
    p6 = np.ascontiguousarray(p6, np.float64)
    T = np.empty(16, np.float32)
    lib().lo_ndt_pose_to_matrix(_p(p6), _p(T))
    return T


def ndt_matrix_to_pose(T16):
    T16 = np.ascontiguousarray(T16, np.float32).reshape(16)
    p = np.empty(6)
    lib().lo_ndt_matrix_to_pose(_p(T16), _p(p))
    return p


def svd_solve6(A, b):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.empty(6)
    lib().lo_svd_solve6(_p(A), _p(b), _p(x))
    return x


def ndt_align(src4, tgt4, params, guess=None):
    src4, tgt4 = _f4(src4), _f4(tgt4)
    g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
    r = LoNdtResult()
    lib().lo_ndt_align(_p(src4), src4.shape[0], _p(tgt4), tgt4.shape[0], C.byref(params), _p(g) if g is not None else None, C.byref(r))
    return {"T": np.array(r.T[:], np.float32), "converged": r.converged, "iterations": r.iterations, "evaluations": r.evaluations,
            "n_cells": r.n_cells, "trans_probability": r.trans_probability}
