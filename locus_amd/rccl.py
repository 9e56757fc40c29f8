"""ctypes binding of liblocus_hip_rccl.so (include/locus_hip_rccl.h): the exchange steps of the multi-GPU path on RCCL for
hosts that run one rank per GPU without torch.distributed.  Test / example binding only -- a C++ caller links the library."""
import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblocus_hip_rccl.so")
EXPORTS = ["lh_rccl_get_unique_id", "lh_rccl_create", "lh_rccl_destroy", "lh_rccl_rank", "lh_rccl_world", "lh_rccl_sum_hook", "lh_rccl_device_sum_hook",
           "lh_rccl_install_sum_hook", "lh_rccl_allgather_results", "lh_rccl_max_double", "lh_rccl_barrier"]
ID_BYTES = 128
_lib = None


def lib():
    global _lib
    if _lib is None:
        capi.lib()   # liblocus_hip.so first (the RCCL library links it)
        if not os.path.exists(LIB_PATH):
            raise ImportError("locus_amd: %s is missing -- build it with __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.lh_rccl_get_unique_id.argtypes = [C.c_char_p]
        L.lh_rccl_create.argtypes = [i32, C.c_char_p, i32, i32, C.POINTER(vp)]
        L.lh_rccl_destroy.argtypes = [vp]
        L.lh_rccl_destroy.restype = None
        L.lh_rccl_rank.argtypes = [vp]
        L.lh_rccl_world.argtypes = [vp]
        L.lh_rccl_install_sum_hook.argtypes = [vp, vp]
        L.lh_rccl_allgather_results.argtypes = [vp, C.POINTER(capi.GicpResult), i32, C.POINTER(capi.GicpResult), i32, C.POINTER(i32)]
        L.lh_rccl_max_double.argtypes = [vp, C.POINTER(C.c_double)]
        L.lh_rccl_barrier.argtypes = [vp]
        _lib = L
    return _lib


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    capi._check(lib().lh_rccl_get_unique_id(buf), "lh_rccl_get_unique_id")
    return buf.raw


class Comm:
    def __init__(self, device, uid, rank, world):
        self.h = C.c_void_p()
        capi._check(lib().lh_rccl_create(device, uid, rank, world, C.byref(self.h)), "lh_rccl_create")
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "h", None):
            lib().lh_rccl_destroy(self.h)
            self.h = C.c_void_p()

    def install_sum_hook(self, ctx):
        capi._check(lib().lh_rccl_install_sum_hook(ctx.h, self.h), "lh_rccl_install_sum_hook")

    def remove_sum_hook(self, ctx):
        capi._check(lib().lh_rccl_install_sum_hook(ctx.h, None), "lh_rccl_install_sum_hook")

    def allgather_results(self, local, cap):
        """local: ctypes array of GicpResult (or a list of them); returns (ctypes array of all results, per-rank counts)"""
        n = len(local)
        arr = local if isinstance(local, C.Array) else (capi.GicpResult * max(n, 1))(*local)
        out = (capi.GicpResult * cap)()
        counts = (C.c_int * self.world)()
        capi._check(lib().lh_rccl_allgather_results(self.h, arr, n, out, cap, counts), "lh_rccl_allgather_results")
        return out, list(counts)

    def max_double(self, v):
        x = C.c_double(float(v))
        capi._check(lib().lh_rccl_max_double(self.h, C.byref(x)), "lh_rccl_max_double")
        return x.value

    def barrier(self):
        capi._check(lib().lh_rccl_barrier(self.h), "lh_rccl_barrier")
