"""ctypes binding of the C ABI in include/locus_hip.h (locus_amd/csrc/liblocus_hip.so).

The library is the product; this module only marshals numpy arrays into lh_cloud_view structs.
There is no CPU fallback: if the shared library is missing, import fails loudly; if no HIP device
is present, lh_create() returns LH_EDEVICE and Context() raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LH_LIB") or os.path.join(_HERE, "csrc", "liblocus_hip.so")  # LH_LIB: A/B a second build of the same ABI
LH_MAX_TRACE = 256
UINT32_MAX = 0xFFFFFFFF

LH_OK, LH_EINVAL, LH_ENOMEM, LH_EDEVICE, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN = 0, -1, -2, -3, -4, -5, -6


class LocusHipError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().lh_status_string(status).decode() if _lib is not None else str(status)
        super().__init__("locus_hip: %s failed: %s (%d)" % (what, msg, status))


class CloudView(C.Structure):
    _fields_ = [
        ("base", C.c_void_p),
        ("count", C.c_uint32),
        ("stride", C.c_uint32),
        ("off_xyz", C.c_uint32),
        ("off_normal", C.c_uint32),
        ("off_intensity", C.c_uint32),
        ("off_curvature", C.c_uint32),
    ]


class GicpParams(C.Structure):
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
        ("enable_timing", C.c_int),
        ("cost_mode", C.c_int),
        ("solver", C.c_int),
        ("bfgs_quad_curv", C.c_int),
    ]


class NdtParams(C.Structure):
    _fields_ = [
        ("resolution", C.c_float),
        ("max_iterations", C.c_int),
        ("step_size", C.c_double),
        ("outlier_ratio", C.c_double),
        ("transformation_epsilon", C.c_double),
        ("min_covar_eigvalue_mult", C.c_double),
        ("min_points_per_voxel", C.c_int),
        ("reserved0", C.c_int),
    ]


class GicpResult(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16),
        ("converged", C.c_int),
        ("iterations", C.c_int),
        ("n_correspondences_last", C.c_int),
        ("status", C.c_int),
        ("fitness", C.c_double),
        ("cost_passes", C.c_int),
        ("reserved", C.c_int),
    ]


class Measurement(C.Structure):   # lh_measurement
    _fields_ = [
        ("result", GicpResult),
        ("Ap", C.c_double * 36),
        ("covariance", C.c_double * 36),
        ("condition_number", C.c_double),
        ("have_information", C.c_int32),
        ("covariance_ok", C.c_int32),
    ]


class GicpTrace(C.Structure):
    _fields_ = [
        ("n_iters", C.c_int),
        ("T", (C.c_float * 16) * LH_MAX_TRACE),
        ("n_corr", C.c_int * LH_MAX_TRACE),
        ("n_passes", C.c_int * LH_MAX_TRACE),
        ("n_inner", C.c_int * LH_MAX_TRACE),
        ("f_end", C.c_double * LH_MAX_TRACE),
        ("delta", C.c_double * LH_MAX_TRACE),
    ]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double), ("bytes", C.c_double)]


# lh_allreduce_fn: int (*)(double* sums, int n, void* user)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)

# every symbol include/locus_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "lh_abi_version", "lh_status_string", "lh_create", "lh_destroy", "lh_synchronize", "lh_default_gicp_params",
    "lh_cloud_create", "lh_cloud_destroy", "lh_cloud_size", "lh_cloud_build_index", "lh_cloud_drop_index",
    "lh_cloud_download", "lh_cloud_transform", "lh_cloud_slice", "lh_cloud_concat", "lh_gicp_create", "lh_gicp_destroy", "lh_gicp_set_params",
    "lh_gicp_set_source", "lh_gicp_set_target", "lh_gicp_set_source_cloud", "lh_gicp_set_target_cloud",
    "lh_gicp_promote_source_to_target", "lh_gicp_align", "lh_gicp_fitness", "lh_set_allreduce", "lh_set_device_allreduce", "lh_nn1", "lh_nn1_cloud", "lh_knn_cloud",
    "lh_gicp_align_batch", "lh_gicp_align_batch_out", "lh_gicp_align_stream", "lh_device_count", "lh_gicp_align_batch_multi", "lh_gicp_align_batch_multi_views", "lh_cov_knn", "lh_gicp_debug_sweep", "lh_gicp_debug_sweep_fused", "lh_gicp_debug_stats", "lh_debug_traversal_stats", "lh_debug_index_dump", "lh_debug_small_index", "lh_gicp_debug_cost", "lh_p2plane_information",
    "lh_icp_covariance", "lh_gicp_measurement_update", "lh_gicp_measurement_update_cloud", "lh_voxel_grid", "lh_cloud_voxel_grid", "lh_cloud_voxel_grid_pointf", "lh_cloud_nearest_neighbors", "lh_cloud_crop_box", "lh_default_ndt_params", "lh_ndt_create", "lh_ndt_destroy", "lh_ndt_set_params",
    "lh_ndt_set_source", "lh_ndt_set_target", "lh_ndt_set_source_cloud", "lh_ndt_set_target_cloud", "lh_ndt_align", "lh_ndt_debug_cells",
    "lh_ndt_debug_derivatives", "lh_map_create", "lh_map_destroy", "lh_map_insert", "lh_map_refresh", "lh_map_cloud", "lh_map_size", "lh_normals_knn", "lh_normals_knn_cloud", "lh_normals_knn_batch", "lh_cov_knn_batch",
    "lh_normals_radius", "lh_normals_radius_cloud", "lh_cloud_remove_nan_normals", "lh_profile_enable",
    "lh_profile_reset", "lh_profile_get", "lh_runtime_init", "lh_runtime_info",
]

class RuntimeInfo(C.Structure):
    _fields_ = [("hw_queues_env", C.c_int), ("streams_probed", C.c_int), ("stream_concurrency", C.c_double), ("adequate", C.c_int), ("reserved", C.c_int)]


_lib = None


def lib_sha256():
    """sha256 of the library file the binding loads: the stamp counter files (profiles/pmc_latest.json) carry to say which build they describe"""
    import hashlib
    return hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "locus_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_double
        L.lh_abi_version.restype = i32
        L.lh_status_string.restype = C.c_char_p
        L.lh_status_string.argtypes = [i32]
        L.lh_create.argtypes = [C.POINTER(vp), i32]
        L.lh_destroy.argtypes = [vp]
        L.lh_destroy.restype = None
        L.lh_synchronize.argtypes = [vp]
        L.lh_runtime_init.argtypes = [i32]
        L.lh_runtime_info.argtypes = [vp, C.POINTER(RuntimeInfo)]
        L.lh_default_gicp_params.argtypes = [C.POINTER(GicpParams)]
        L.lh_default_gicp_params.restype = None
        L.lh_cloud_create.argtypes = [vp, C.POINTER(CloudView), C.POINTER(vp)]
        L.lh_cloud_destroy.argtypes = [vp]
        L.lh_cloud_destroy.restype = None
        L.lh_cloud_size.argtypes = [vp]
        L.lh_cloud_size.restype = u32
        L.lh_cloud_build_index.argtypes = [vp]
        L.lh_cloud_drop_index.argtypes = [vp]
        L.lh_cloud_download.argtypes = [vp, vp, u32, u32, u32, u32, u32]
        L.lh_cloud_transform.argtypes = [vp, vp, i32, C.POINTER(vp)]
        L.lh_gicp_create.argtypes = [vp, C.POINTER(GicpParams), C.POINTER(vp)]
        L.lh_gicp_destroy.argtypes = [vp]
        L.lh_gicp_destroy.restype = None
        L.lh_gicp_set_params.argtypes = [vp, C.POINTER(GicpParams)]
        L.lh_gicp_set_source.argtypes = [vp, C.POINTER(CloudView)]
        L.lh_gicp_set_target.argtypes = [vp, C.POINTER(CloudView)]
        L.lh_gicp_set_source_cloud.argtypes = [vp, vp]
        L.lh_gicp_set_target_cloud.argtypes = [vp, vp]
        L.lh_gicp_promote_source_to_target.argtypes = [vp]
        L.lh_gicp_align.argtypes = [vp, vp, C.POINTER(GicpResult), C.POINTER(GicpTrace), vp, u32, u32]
        L.lh_gicp_fitness.argtypes = [vp, C.POINTER(dbl)]
        L.lh_gicp_measurement_update.argtypes = [vp, vp, C.c_int, dbl, vp, vp, vp, u32, u32, u32]
        L.lh_gicp_measurement_update_cloud.argtypes = [vp, vp, C.c_int, dbl, vp, vp, C.POINTER(vp)]
        L.lh_nn1.argtypes = [vp, C.POINTER(CloudView), vp, vp]
        L.lh_nn1_cloud.argtypes = [vp, vp, vp, vp]
        L.lh_knn_cloud.argtypes = [vp, vp, i32, vp, vp]
        L.lh_gicp_align_batch.argtypes = [vp, C.POINTER(GicpParams), i32, C.POINTER(vp), C.POINTER(vp), vp,
                                          C.POINTER(GicpResult), i32]
        L.lh_gicp_align_batch_out.argtypes = [vp, C.POINTER(GicpParams), i32, C.POINTER(vp), C.POINTER(vp), vp,
                                              C.POINTER(GicpResult), C.POINTER(vp), i32]
        L.lh_gicp_align_stream.argtypes = [vp, C.POINTER(GicpParams), i32, C.POINTER(vp), vp, C.POINTER(GicpResult), i32]
        L.lh_device_count.restype = i32
        L.lh_gicp_align_batch_multi.argtypes = [i32, C.POINTER(vp), C.POINTER(GicpParams), i32, C.POINTER(vp), C.POINTER(vp), vp,
                                                C.POINTER(GicpResult), C.POINTER(vp), i32]
        L.lh_gicp_align_batch_multi_views.argtypes = [i32, C.POINTER(vp), C.POINTER(GicpParams), i32, C.POINTER(CloudView), C.POINTER(CloudView), vp,
                                                      C.POINTER(GicpResult), i32]
        L.lh_cov_knn.argtypes = [vp, i32, dbl, vp]
        L.lh_gicp_debug_sweep.argtypes = [vp, vp, vp, vp, vp]
        L.lh_gicp_debug_sweep_fused.argtypes = [vp, vp, C.c_int, vp, vp, vp]
        L.lh_gicp_debug_stats.argtypes = [vp, vp, i32]
        L.lh_debug_traversal_stats.argtypes = [vp, vp, vp, vp, i32, vp]
        L.lh_gicp_debug_cost.argtypes = [vp, vp, C.POINTER(dbl), vp, vp, C.POINTER(i32)]
        L.lh_p2plane_information.argtypes = [vp, vp, vp, vp, vp]
        L.lh_icp_covariance.argtypes = [vp, dbl, vp, C.POINTER(dbl)]
        L.lh_voxel_grid.argtypes = [vp, C.POINTER(CloudView), C.c_float, i32, dbl, dbl, vp, u32, C.POINTER(u32)]
        L.lh_cloud_voxel_grid.argtypes = [vp, C.c_float, i32, dbl, dbl, C.POINTER(vp)]
        L.lh_cloud_voxel_grid_pointf.argtypes = [vp, C.c_float, C.POINTER(vp)]
        L.lh_cloud_nearest_neighbors.argtypes = [vp, vp, C.POINTER(vp)]
        L.lh_normals_knn.argtypes = [vp, C.POINTER(CloudView), i32, vp]
        L.lh_normals_knn_cloud.argtypes = [vp, i32]
        L.lh_debug_index_dump.argtypes = [vp, vp, vp, u32, vp]
        L.lh_debug_small_index.argtypes = [i32]
        L.lh_debug_small_index.restype = i32
        L.lh_normals_knn_batch.argtypes = [C.POINTER(vp), i32, i32]
        L.lh_cov_knn_batch.argtypes = [C.POINTER(vp), i32, i32, dbl]
        L.lh_cloud_slice.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.lh_cloud_concat.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
        L.lh_default_ndt_params.argtypes = [C.POINTER(NdtParams)]
        L.lh_default_ndt_params.restype = None
        L.lh_ndt_create.argtypes = [vp, C.POINTER(NdtParams), C.POINTER(vp)]
        L.lh_ndt_destroy.argtypes = [vp]
        L.lh_ndt_destroy.restype = None
        L.lh_ndt_set_params.argtypes = [vp, C.POINTER(NdtParams)]
        L.lh_ndt_set_source.argtypes = [vp, C.POINTER(CloudView)]
        L.lh_ndt_set_target.argtypes = [vp, C.POINTER(CloudView)]
        L.lh_ndt_set_source_cloud.argtypes = [vp, vp]
        L.lh_ndt_set_target_cloud.argtypes = [vp, vp]
        L.lh_ndt_align.argtypes = [vp, vp, C.POINTER(GicpResult), vp, u32, u32]
        L.lh_ndt_debug_cells.argtypes = [vp, C.POINTER(i32), vp, vp, vp, i32]
        L.lh_ndt_debug_derivatives.argtypes = [vp, vp, i32, i32, C.POINTER(dbl), vp, vp]
        L.lh_cloud_crop_box.argtypes = [vp, vp, vp, C.c_float, i32, C.POINTER(vp)]
        L.lh_map_create.argtypes = [vp, dbl, C.POINTER(vp)]
        L.lh_map_destroy.argtypes = [vp]
        L.lh_map_destroy.restype = None
        L.lh_map_insert.argtypes = [vp, vp, C.POINTER(u32)]
        L.lh_map_refresh.argtypes = [vp, vp, C.c_float]
        L.lh_map_cloud.argtypes = [vp]
        L.lh_map_cloud.restype = vp
        L.lh_map_size.argtypes = [vp]
        L.lh_map_size.restype = u32
        L.lh_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
        L.lh_normals_radius.argtypes = [vp, C.POINTER(CloudView), C.c_float, vp]
        L.lh_normals_radius_cloud.argtypes = [vp, C.c_float]
        L.lh_cloud_remove_nan_normals.argtypes = [vp, C.POINTER(vp)]
        L.lh_profile_enable.argtypes = [vp, i32]
        L.lh_profile_reset.argtypes = [vp]
        L.lh_profile_get.argtypes = [vp, C.POINTER(KernelStat), i32]
        _lib = L
    return _lib


def _check(st, what):
    if st != LH_OK:
        raise LocusHipError(st, what)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---- point-type layouts (pcl::PointXYZI = 32 B, pcl::PointXYZINormal = PointF = 48 B) --------------------
POINT_XYZI = np.dtype({"names": ["x", "y", "z", "intensity"], "formats": ["f4"] * 4, "offsets": [0, 4, 8, 16],
                       "itemsize": 32})
POINT_XYZINORMAL = np.dtype({"names": ["x", "y", "z", "normal_x", "normal_y", "normal_z", "intensity", "curvature"],
                             "formats": ["f4"] * 8, "offsets": [0, 4, 8, 16, 20, 24, 32, 36], "itemsize": 48})


def make_pointf(xyz, normals=None, intensity=None, curvature=None):
    """numpy structured array with pcl::PointXYZINormal layout (PointF, the type GICP consumes)."""
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    a = np.zeros(n, POINT_XYZINORMAL)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if normals is not None:
        normals = np.asarray(normals, np.float32)
        a["normal_x"], a["normal_y"], a["normal_z"] = normals[:, 0], normals[:, 1], normals[:, 2]
        if curvature is None and normals.shape[1] > 3:
            curvature = normals[:, 3]
    if intensity is not None:
        a["intensity"] = np.asarray(intensity, np.float32)
    if curvature is not None:
        a["curvature"] = np.asarray(curvature, np.float32)
    return a


def make_pointxyzi(xyz, intensity=None):
    xyz = np.asarray(xyz, np.float32)
    a = np.zeros(xyz.shape[0], POINT_XYZI)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if intensity is not None:
        a["intensity"] = np.asarray(intensity, np.float32)
    return a


def view_of(points, with_normals=None):
    """lh_cloud_view of a structured array (PointXYZI / PointXYZINormal) or of a float32 (n,3|4) xyz array.
    Returns (view, keepalive)."""
    if isinstance(points, np.ndarray) and points.dtype.names:
        a = np.ascontiguousarray(points)
        f = a.dtype.fields
        has_n = "normal_x" in f if with_normals is None else (with_normals and "normal_x" in f)
        v = CloudView(_ptr(a), a.shape[0], a.dtype.itemsize, f["x"][1],
                      f["normal_x"][1] if has_n else UINT32_MAX,
                      f["intensity"][1] if "intensity" in f else UINT32_MAX,
                      f["curvature"][1] if (has_n and "curvature" in f) else UINT32_MAX)
        return v, a
    a = np.ascontiguousarray(points, np.float32)
    assert a.ndim == 2 and a.shape[1] in (3, 4)
    v = CloudView(_ptr(a), a.shape[0], a.shape[1] * 4, 0, UINT32_MAX, UINT32_MAX, UINT32_MAX)
    return v, a


def small_index(enable):
    """lh_debug_small_index: switch the one-launch build of small clouds off (0) / on (1); returns the previous setting"""
    return lib().lh_debug_small_index(int(enable))


def normals_knn_batch(clouds, k=20):
    """lh_normals_knn_batch: the normal filter for a queue of device clouds, one index build + one k-NN launch"""
    arr = (C.c_void_p * len(clouds))(*[c.h for c in clouds])
    _check(lib().lh_normals_knn_batch(arr, len(clouds), k), "lh_normals_knn_batch")


def cov_knn_batch(clouds, k=20, eps=1e-3):
    arr = (C.c_void_p * len(clouds))(*[c.h for c in clouds])
    _check(lib().lh_cov_knn_batch(arr, len(clouds), k, eps), "lh_cov_knn_batch")


def default_params(**kw):
    p = GicpParams()
    lib().lh_default_gicp_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


class Context:
    """lh_ctx: one per GPU / rank."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().lh_create(C.byref(self.h), device), "lh_create")
        self.device = device

    def close(self):
        if self.h:
            lib().lh_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().lh_synchronize(self.h), "lh_synchronize")

    def runtime_info(self):
        """lh_runtime_info: GPU_MAX_HW_QUEUES as the process sees it and the MEASURED concurrency of sixteen streams"""
        ri = RuntimeInfo()
        _check(lib().lh_runtime_info(self.h, C.byref(ri)), "lh_runtime_info")
        return {"hw_queues_env": ri.hw_queues_env, "streams_probed": ri.streams_probed, "stream_concurrency": ri.stream_concurrency, "adequate": bool(ri.adequate)}

    def set_allreduce(self, fn):
        """source-sharded single pair: fn(numpy float64 view) sums in place over the ranks; None removes it"""
        if fn is None:
            self._reduce_cb = C.cast(None, ALLREDUCE_FN)
        else:
            def _cb(ptr, n, _user):
                try:
                    fn(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except Exception:  # an exception must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            self._reduce_cb = ALLREDUCE_FN(_cb)  # keep alive as long as the context uses it
        _check(lib().lh_set_allreduce(self.h, self._reduce_cb, None), "lh_set_allreduce")

    def profile(self, on=True):
        _check(lib().lh_profile_enable(self.h, 1 if on else 0), "lh_profile_enable")

    def profile_reset(self):
        _check(lib().lh_profile_reset(self.h), "lh_profile_reset")

    def profile_get(self):
        buf = (KernelStat * 64)()
        n = lib().lh_profile_get(self.h, buf, 64)
        return {buf[i].name.decode(): {"launches": int(buf[i].launches), "ms": buf[i].total_ms, "bytes": buf[i].bytes}
                for i in range(min(n, 64))}

    def voxel_grid(self, points, leaf, limit_axis=-1, lo=-np.inf, hi=np.inf, capacity=None):
        v, keep = view_of(points)
        cap = capacity if capacity is not None else v.count
        out = np.empty((max(cap, 1), 4), np.float32)
        cnt = C.c_uint32()
        _check(lib().lh_voxel_grid(self.h, C.byref(v), leaf, limit_axis, float(max(lo, -3e38)), float(min(hi, 3e38)),
                                   _ptr(out), cap, C.byref(cnt)), "lh_voxel_grid")
        return out[: min(cnt.value, cap)].copy(), cnt.value

    def normals_knn(self, points, k=20):
        v, keep = view_of(points)
        out = np.empty((v.count, 4), np.float32)
        _check(lib().lh_normals_knn(self.h, C.byref(v), k, _ptr(out)), "lh_normals_knn")
        return out

    def normals_radius(self, points, radius=0.3):
        v, keep = view_of(points)
        out = np.empty((v.count, 4), np.float32)
        _check(lib().lh_normals_radius(self.h, C.byref(v), radius, _ptr(out)), "lh_normals_radius")
        return out

    def p2plane_information(self, query, reference, corr):
        corr = np.ascontiguousarray(corr, np.int64)
        Ap = np.empty((6, 6), np.float64)
        _check(lib().lh_p2plane_information(self.h, query.h, reference.h, _ptr(corr), _ptr(Ap)), "lh_p2plane_information")
        return Ap


def icp_covariance(Ap, icp_max_covariance=0.01):
    Ap = np.ascontiguousarray(Ap, np.float64)
    cov = np.empty((6, 6), np.float64)
    cond = C.c_double()
    st = lib().lh_icp_covariance(_ptr(Ap), icp_max_covariance, _ptr(cov), C.byref(cond))
    return st == LH_OK, cov, cond.value


class Cloud:
    """lh_cloud: device-resident cloud."""

    def __init__(self, ctx, points, _handle=None, _borrowed=False, _owner=None):
        self.ctx = ctx
        self._borrowed = _borrowed   # a handle owned by another object (lh_map_cloud): never destroyed from here
        self._owner = _owner
        if _handle is not None:
            self.h = _handle
            return
        v, keep = view_of(points)
        self.h = C.c_void_p()
        _check(lib().lh_cloud_create(ctx.h, C.byref(v), C.byref(self.h)), "lh_cloud_create")

    def __len__(self):
        return lib().lh_cloud_size(self.h)

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                lib().lh_cloud_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass

    def build_index(self):
        _check(lib().lh_cloud_build_index(self.h), "lh_cloud_build_index")

    def drop_index(self):
        _check(lib().lh_cloud_drop_index(self.h), "lh_cloud_drop_index")

    def download(self):
        a = np.zeros(len(self), POINT_XYZINORMAL)
        f = a.dtype.fields
        _check(lib().lh_cloud_download(self.h, _ptr(a), a.dtype.itemsize, f["x"][1], f["normal_x"][1], f["intensity"][1],
                                       f["curvature"][1]), "lh_cloud_download")
        return a

    def transform(self, T16_colmajor, with_normals=False):
        T = np.ascontiguousarray(T16_colmajor, np.float32).reshape(16)
        out = C.c_void_p()
        _check(lib().lh_cloud_transform(self.h, _ptr(T), 1 if with_normals else 0, C.byref(out)), "lh_cloud_transform")
        return Cloud(self.ctx, None, _handle=out)

    def slice(self, first, count):
        out = C.c_void_p()
        _check(lib().lh_cloud_slice(self.h, first, count, C.byref(out)), "lh_cloud_slice")
        return Cloud(self.ctx, None, _handle=out)

    @staticmethod
    def concat(parts):
        """PointCloudMerger.cc:158-159 (`*merged = *a + *b`) on the device"""
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        out = C.c_void_p()
        _check(lib().lh_cloud_concat(arr, len(parts), C.byref(out)), "lh_cloud_concat")
        return Cloud(parts[0].ctx, None, _handle=out)

    def voxel_grid(self, leaf, limit_axis=-1, lo=-np.inf, hi=np.inf):
        out = C.c_void_p()
        _check(lib().lh_cloud_voxel_grid(self.h, leaf, limit_axis, float(max(lo, -3e38)), float(min(hi, 3e38)), C.byref(out)),
               "lh_cloud_voxel_grid")
        return Cloud(self.ctx, None, _handle=out)

    def voxel_grid_pointf(self, leaf):
        """pcl::VoxelGrid<PointXYZINormal> (PointCloudFilter.cc:119-124): every field averaged, normals re-normalised"""
        out = C.c_void_p()
        _check(lib().lh_cloud_voxel_grid_pointf(self.h, leaf, C.byref(out)), "lh_cloud_voxel_grid_pointf")
        return Cloud(self.ctx, None, _handle=out)

    def crop_box(self, min_pt, max_pt, yaw=0.0, negative=True):
        """BodyFilter (body_filter.cc:27-52): pcl::CropBox, negative=True removes the inside of the box"""
        mn = np.ascontiguousarray(min_pt, np.float32).reshape(3)
        mx = np.ascontiguousarray(max_pt, np.float32).reshape(3)
        out = C.c_void_p()
        _check(lib().lh_cloud_crop_box(self.h, _ptr(mn), _ptr(mx), float(yaw), 1 if negative else 0, C.byref(out)), "lh_cloud_crop_box")
        return Cloud(self.ctx, None, _handle=out)

    def nearest_neighbors(self, query_cloud):
        """mapper ApproxNearestNeighbors analogue: cloud of this (map) cloud's points nearest to each query point"""
        out = C.c_void_p()
        _check(lib().lh_cloud_nearest_neighbors(self.h, query_cloud.h, C.byref(out)), "lh_cloud_nearest_neighbors")
        return Cloud(self.ctx, None, _handle=out)

    def nn1(self, query_cloud):
        n = len(query_cloud)
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        _check(lib().lh_nn1_cloud(self.h, query_cloud.h, _ptr(idx), _ptr(d2)), "lh_nn1_cloud")
        return idx, d2

    def knn(self, query_cloud, k):
        n = len(query_cloud)
        idx = np.empty((n, k), np.int32)
        d2 = np.empty((n, k), np.float32)
        _check(lib().lh_knn_cloud(self.h, query_cloud.h, k, _ptr(idx), _ptr(d2)), "lh_knn_cloud")
        return idx, d2

    def traversal_stats(self, query_cloud, T16=None, cand=None, leaf_prescan=False):
        out = np.zeros(5, np.uint64)
        T = np.ascontiguousarray(T16, np.float32).reshape(16) if T16 is not None else None
        cd = np.ascontiguousarray(cand, np.int32) if cand is not None else None
        _check(lib().lh_debug_traversal_stats(self.h, query_cloud.h, _ptr(T), _ptr(cd), 1 if leaf_prescan else 0, _ptr(out)),
               "lh_debug_traversal_stats")
        n, w = len(query_cloud), int(out[3])
        return {"nodes_per_query": out[0] / n, "leaves_per_query": out[1] / n, "wave_max_visits_avg": out[2] / max(w, 1),
                "max_visits": int(out[4])}

    def index_dump(self):
        """(sorted (n + 8, 4) float32, nodes (n, 16) uint32, header (16,) uint32) of the cloud's NN index (built if missing)"""
        n = len(self)
        srt = np.empty((n + 8, 4), np.float32)
        nodes = np.zeros((max(n, 1), 16), np.uint32)
        hdr = np.zeros(16, np.uint32)
        _check(lib().lh_debug_index_dump(self.h, _ptr(srt), _ptr(nodes), n, _ptr(hdr)), "lh_debug_index_dump")
        return srt, nodes, hdr

    def cov_knn(self, k=20, eps=1e-3):
        cov = np.empty((len(self), 3, 3), np.float64)
        _check(lib().lh_cov_knn(self.h, k, eps, _ptr(cov)), "lh_cov_knn")
        return cov

    def normals_knn(self, k=20):
        _check(lib().lh_normals_knn_cloud(self.h, k), "lh_normals_knn_cloud")

    def normals_radius(self, radius=0.3):
        _check(lib().lh_normals_radius_cloud(self.h, radius), "lh_normals_radius_cloud")

    def cov_knn_planes(self, k=20, eps=1e-3):
        """the covariances the cloud already holds on the device (after cov_knn_batch / an alignment in recompute mode), as lh_cov_knn returns them"""
        return self.cov_knn(k, eps)

    def remove_nan_normals(self):
        """pcl::removeNaNNormalsFromPointCloud (normal_computation.cc:52-56): new cloud without the NaN-normal points"""
        out = C.c_void_p()
        _check(lib().lh_cloud_remove_nan_normals(self.h, C.byref(out)), "lh_cloud_remove_nan_normals")
        return Cloud(self.ctx, None, _handle=out)


def default_ndt_params(**kw):
    p = NdtParams()
    lib().lh_default_ndt_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Ndt:
    """lh_ndt: the registration object of registration_method "ndt" (pclomp::NormalDistributionsTransform)."""

    def __init__(self, ctx, params=None):
        self.ctx = ctx
        self.params = params or default_ndt_params()
        self.h = C.c_void_p()
        _check(lib().lh_ndt_create(ctx.h, C.byref(self.params), C.byref(self.h)), "lh_ndt_create")
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            lib().lh_ndt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass

    def set_params(self, params):
        self.params = params
        _check(lib().lh_ndt_set_params(self.h, C.byref(params)), "lh_ndt_set_params")

    def set_source(self, cloud):
        if isinstance(cloud, Cloud):
            self._keep.append(cloud)
            _check(lib().lh_ndt_set_source_cloud(self.h, cloud.h), "lh_ndt_set_source_cloud")
        else:
            v, keep = view_of(cloud)
            _check(lib().lh_ndt_set_source(self.h, C.byref(v)), "lh_ndt_set_source")

    def set_target(self, cloud):
        if isinstance(cloud, Cloud):
            self._keep.append(cloud)
            _check(lib().lh_ndt_set_target_cloud(self.h, cloud.h), "lh_ndt_set_target_cloud")
        else:
            v, keep = view_of(cloud)
            _check(lib().lh_ndt_set_target(self.h, C.byref(v)), "lh_ndt_set_target")

    def align(self, guess=None):
        r = GicpResult()
        g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
        _check(lib().lh_ndt_align(self.h, _ptr(g), C.byref(r), None, 0, 0), "lh_ndt_align")
        d = _result_dict(r)
        d["trans_probability"] = d["fitness"]
        d["evaluations"] = d["cost_passes"]
        d["n_cells"] = d["n_corr_last"]
        return d

    def cells(self):
        n = C.c_int()
        _check(lib().lh_ndt_debug_cells(self.h, C.byref(n), None, None, None, 0), "lh_ndt_debug_cells")
        mean, icov, cen = np.empty((n.value, 3)), np.empty((n.value, 3, 3)), np.empty((n.value, 4), np.float32)
        _check(lib().lh_ndt_debug_cells(self.h, C.byref(n), _ptr(mean), _ptr(icov), _ptr(cen), n.value), "lh_ndt_debug_cells")
        return mean, icov, cen

    def derivatives(self, p6, want_h=True, hessian_only=False):
        p = np.ascontiguousarray(p6, np.float64)
        s = C.c_double()
        g, H = np.empty(6), np.empty((6, 6))
        _check(lib().lh_ndt_debug_derivatives(self.h, _ptr(p), 1 if want_h else 0, 1 if hessian_only else 0, C.byref(s), _ptr(g), _ptr(H)),
               "lh_ndt_debug_derivatives")
        return s.value, g, H


class Map:
    """lh_map: the device-resident local map behind mapper_->InsertPoints / ApproxNearestNeighbors / Refresh."""

    def __init__(self, ctx, octree_resolution):
        self.ctx = ctx
        self.h = C.c_void_p()
        _check(lib().lh_map_create(ctx.h, float(octree_resolution), C.byref(self.h)), "lh_map_create")

    def close(self):
        if getattr(self, "h", None):
            lib().lh_map_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass

    def __len__(self):
        return int(lib().lh_map_size(self.h))

    def insert(self, cloud):
        n = C.c_uint32()
        _check(lib().lh_map_insert(self.h, cloud.h, C.byref(n)), "lh_map_insert")
        return n.value

    def refresh(self, center, half_extent):
        c = np.ascontiguousarray(center, np.float32).reshape(3)
        _check(lib().lh_map_refresh(self.h, _ptr(c), float(half_extent)), "lh_map_refresh")

    def cloud(self):
        """the map as a (borrowed) Cloud: GICP target / nearest_neighbors source; None while the map is empty"""
        h = lib().lh_map_cloud(self.h)
        if not h:
            return None
        return Cloud(self.ctx, None, _handle=C.c_void_p(h), _borrowed=True, _owner=self)


def _result_dict(r, trace=None):
    out = {
        "T": np.array(r.T[:], np.float32), "converged": r.converged, "iterations": r.iterations,
        "n_corr_last": r.n_correspondences_last, "status": r.status, "fitness": r.fitness, "cost_passes": r.cost_passes,
    }
    if trace is not None:
        k = trace.n_iters
        out["trace"] = {
            "T": np.array([trace.T[i][:] for i in range(k)], np.float32).reshape(k, 16),
            "n_corr": np.array(trace.n_corr[:k]), "n_passes": np.array(trace.n_passes[:k]),
            "n_inner": np.array(trace.n_inner[:k]), "f_end": np.array(trace.f_end[:k]), "delta": np.array(trace.delta[:k]),
        }
    return out


class Gicp:
    """lh_gicp: one registration object (the `icp_` member of PointCloudOdometry / PointCloudLocalization)."""

    def __init__(self, ctx, params=None):
        self.ctx = ctx
        self.params = params or default_params()
        self.h = C.c_void_p()
        _check(lib().lh_gicp_create(ctx.h, C.byref(self.params), C.byref(self.h)), "lh_gicp_create")
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            lib().lh_gicp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass

    def set_params(self, params):
        self.params = params
        _check(lib().lh_gicp_set_params(self.h, C.byref(params)), "lh_gicp_set_params")

    def set_source(self, points):
        self._n_src = len(points)
        if isinstance(points, Cloud):
            self._src = points
            _check(lib().lh_gicp_set_source_cloud(self.h, points.h), "lh_gicp_set_source_cloud")
        else:
            v, keep = view_of(points)
            _check(lib().lh_gicp_set_source(self.h, C.byref(v)), "lh_gicp_set_source")

    def set_target(self, points):
        if isinstance(points, Cloud):
            self._tgt = points
            _check(lib().lh_gicp_set_target_cloud(self.h, points.h), "lh_gicp_set_target_cloud")
        else:
            v, keep = view_of(points)
            _check(lib().lh_gicp_set_target(self.h, C.byref(v)), "lh_gicp_set_target")

    def promote_source_to_target(self):
        _check(lib().lh_gicp_promote_source_to_target(self.h), "lh_gicp_promote_source_to_target")

    def align(self, guess=None, want_trace=True, aligned_out=None, raise_on_error=True):
        r = GicpResult()
        tr = GicpTrace() if want_trace else None
        g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
        if aligned_out is not None:
            f = aligned_out.dtype.fields
            st = lib().lh_gicp_align(self.h, _ptr(g), C.byref(r), C.byref(tr) if tr is not None else None, _ptr(aligned_out),
                                     aligned_out.dtype.itemsize, f["x"][1])
        else:
            st = lib().lh_gicp_align(self.h, _ptr(g), C.byref(r), C.byref(tr) if tr is not None else None, None, 0, 0)
        if raise_on_error and st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
            raise LocusHipError(st, "lh_gicp_align")
        return _result_dict(r, tr)

    def fitness(self):
        f = C.c_double()
        _check(lib().lh_gicp_fitness(self.h, C.byref(f)), "lh_gicp_fitness")
        return f.value

    def measurement_update(self, guess=None, want_information=True, icp_max_covariance=0.01, aligned_out=None, want_corr=True, aligned_cloud=False):
        """lh_gicp_measurement_update(_cloud): align -> aligned query (points + normals) -> ungated 1-NN -> Ap -> covariance, in one call;
        aligned_cloud: the aligned query stays on the device (out["aligned"] = a Cloud)"""
        m = Measurement()
        g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
        corr = np.empty(self._n_src, np.int32) if want_corr else None
        ac = C.c_void_p()
        if aligned_cloud:
            st = lib().lh_gicp_measurement_update_cloud(self.h, _ptr(g), 1 if want_information else 0, icp_max_covariance, C.byref(m), _ptr(corr), C.byref(ac))
        elif aligned_out is not None:
            f = aligned_out.dtype.fields
            off_n = f["normal_x"][1] if "normal_x" in f else 0xFFFFFFFF
            st = lib().lh_gicp_measurement_update(self.h, _ptr(g), 1 if want_information else 0, icp_max_covariance, C.byref(m), _ptr(corr),
                                                  _ptr(aligned_out), aligned_out.dtype.itemsize, f["x"][1], off_n)
        else:
            st = lib().lh_gicp_measurement_update(self.h, _ptr(g), 1 if want_information else 0, icp_max_covariance, C.byref(m), _ptr(corr), None, 0, 0, 0xFFFFFFFF)
        if st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
            raise LocusHipError(st, "lh_gicp_measurement_update")
        out = _result_dict(m.result)
        out["corr"] = corr
        if aligned_cloud and ac:
            out["aligned"] = Cloud(self.ctx, None, _handle=ac)
        if m.have_information:
            out["Ap"] = np.array(m.Ap, np.float64).reshape(6, 6)
            out["covariance"] = np.array(m.covariance, np.float64).reshape(6, 6)
            out["condition_number"] = m.condition_number
            out["covariance_ok"] = bool(m.covariance_ok)
        return out

    def nn1(self, points):
        v, keep = view_of(points)
        idx = np.empty(v.count, np.int32)
        d2 = np.empty(v.count, np.float32)
        _check(lib().lh_nn1(self.h, C.byref(v), _ptr(idx), _ptr(d2)), "lh_nn1")
        return idx, d2

    def debug_sweep(self, T16, n_src, guess=None):
        T = np.ascontiguousarray(T16, np.float32).reshape(16)
        g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
        idx = np.empty(n_src, np.int32)
        maha = np.zeros((n_src, 3, 3), np.float64)
        _check(lib().lh_gicp_debug_sweep(self.h, _ptr(T), _ptr(g), _ptr(idx), _ptr(maha)), "lh_gicp_debug_sweep")
        return idx, maha

    def debug_sweep_fused(self, T16, n_src, sweep_index):
        """one sweep of the cost_mode 1 kernels (fused / k_late + k_walk by sweep_index): neighbour indices, tree walks, 74 sums"""
        T = np.ascontiguousarray(T16, np.float32).reshape(16)
        idx = np.empty(n_src, np.int32)
        walks = np.zeros(1, np.uint64)
        sums = np.zeros(74, np.float64)
        _check(lib().lh_gicp_debug_sweep_fused(self.h, _ptr(T), int(sweep_index), _ptr(idx), _ptr(walks), _ptr(sums)), "lh_gicp_debug_sweep_fused")
        return idx, int(walks[0]), sums

    def debug_stats(self, reset=True):
        out = np.zeros(2, np.uint64)
        _check(lib().lh_gicp_debug_stats(self.h, _ptr(out), 1 if reset else 0), "lh_gicp_debug_stats")
        return int(out[0]), int(out[1])

    def debug_cost(self, x6):
        x = np.ascontiguousarray(x6, np.float64)
        f = C.c_double()
        g = np.empty(6, np.float64)
        sums = np.empty(13, np.float64)
        m = C.c_int()
        _check(lib().lh_gicp_debug_cost(self.h, _ptr(x), C.byref(f), _ptr(g), _ptr(sums), C.byref(m)), "lh_gicp_debug_cost")
        return f.value, g, sums, m.value


def align_batch(ctx, params, src_clouds, tgt_clouds, guesses=None, max_in_flight=0):
    n = len(src_clouds)
    assert n == len(tgt_clouds)
    S = (C.c_void_p * n)(*[c.h for c in src_clouds])
    T = (C.c_void_p * n)(*[c.h for c in tgt_clouds])
    out = (GicpResult * n)()
    g = np.ascontiguousarray(guesses, np.float32).reshape(n * 16) if guesses is not None else None
    st = lib().lh_gicp_align_batch(ctx.h, C.byref(params), n, S, T, _ptr(g), out, max_in_flight)
    if st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
        raise LocusHipError(st, "lh_gicp_align_batch")
    return [_result_dict(out[i]) for i in range(n)]


def align_stream(ctx, params, scans, guesses=None, max_in_flight=0, raw=False):
    """lh_gicp_align_stream: pair i = scans[i + 1] -> scans[i]; a scan's index is built once and kept (PointCloudOdometry over a queue)"""
    n = len(scans)
    S = (C.c_void_p * n)(*[c.h for c in scans])
    out = (GicpResult * (n - 1))()
    g = np.ascontiguousarray(guesses, np.float32).reshape((n - 1) * 16) if guesses is not None else None
    st = lib().lh_gicp_align_stream(ctx.h, C.byref(params), n, S, _ptr(g), out, max_in_flight)
    if st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
        raise LocusHipError(st, "lh_gicp_align_stream")
    return out if raw else [_result_dict(out[i]) for i in range(n - 1)]


def align_batch_out(ctx, params, src_clouds, tgt_clouds, guesses=None, max_in_flight=0, aligned=None, raw=False):
    """lh_gicp_align_batch_out: results plus align()'s output clouds.  aligned = list of Cloud / None per pair (None: created);
    returns (results, aligned clouds).  raw=True returns the ctypes result array instead of dicts (bench.py gathers it)."""
    n = len(src_clouds)
    assert n == len(tgt_clouds)
    S = (C.c_void_p * n)(*[c.h for c in src_clouds])
    T = (C.c_void_p * n)(*[c.h for c in tgt_clouds])
    A = (C.c_void_p * n)(*[(a.h if a is not None else None) for a in (aligned or [None] * n)])
    out = (GicpResult * n)()
    g = np.ascontiguousarray(guesses, np.float32).reshape(n * 16) if guesses is not None else None
    st = lib().lh_gicp_align_batch_out(ctx.h, C.byref(params), n, S, T, _ptr(g), out, A, max_in_flight)
    if st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
        raise LocusHipError(st, "lh_gicp_align_batch_out")
    clouds = []
    for i in range(n):
        if aligned is not None and aligned[i] is not None:
            clouds.append(aligned[i])
        else:
            clouds.append(Cloud(ctx, None, _handle=C.c_void_p(A[i])))
    return (out if raw else [_result_dict(out[i]) for i in range(n)]), clouds


def device_count():
    return lib().lh_device_count()


def align_batch_multi(ctxs, params, src_clouds, tgt_clouds, guesses=None, max_in_flight=0):
    """lh_gicp_align_batch_multi: every pair runs on the context (GPU) that owns its clouds, one host thread per device"""
    n = len(src_clouds)
    assert n == len(tgt_clouds)
    X = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    S = (C.c_void_p * n)(*[c.h for c in src_clouds])
    T = (C.c_void_p * n)(*[c.h for c in tgt_clouds])
    out = (GicpResult * n)()
    g = np.ascontiguousarray(guesses, np.float32).reshape(n * 16) if guesses is not None else None
    st = lib().lh_gicp_align_batch_multi(len(ctxs), X, C.byref(params), n, S, T, _ptr(g), out, None, max_in_flight)
    if st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
        raise LocusHipError(st, "lh_gicp_align_batch_multi")
    return [_result_dict(out[i]) for i in range(n)]


def align_batch_multi_views(ctxs, params, src_points, tgt_points, guesses=None, max_in_flight=0):
    """lh_gicp_align_batch_multi_views: host arrays (PointXYZINormal records) in, results out; contiguous pair blocks per context"""
    n = len(src_points)
    assert n == len(tgt_points)
    X = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    keep, SV, TV = [], (CloudView * n)(), (CloudView * n)()
    for i in range(n):
        sv, k1 = view_of(src_points[i])
        tv, k2 = view_of(tgt_points[i])
        SV[i], TV[i] = sv, tv
        keep += [k1, k2]
    out = (GicpResult * n)()
    g = np.ascontiguousarray(guesses, np.float32).reshape(n * 16) if guesses is not None else None
    st = lib().lh_gicp_align_batch_multi_views(len(ctxs), X, C.byref(params), n, SV, TV, _ptr(g), out, max_in_flight)
    if st not in (LH_OK, LH_ETOO_FEW_CORR, LH_ESOLVER, LH_ENO_NN):
        raise LocusHipError(st, "lh_gicp_align_batch_multi_views")
    return [_result_dict(out[i]) for i in range(n)]
