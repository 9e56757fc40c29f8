"""Does the library's own default for GPU_MAX_HW_QUEUES (lh_api.hip lh_runtime_defaults) reach the HIP runtime in a process that loads
liblocus_hip.so before anything else touched HIP (a C++ host, or Python without torch)?

  python tools/probe_queues.py ctor     # variable removed from the environment before the library is loaded: the constructor sets it
  python tools/probe_queues.py 4        # explicit value (the runtime's default)
  python tools/probe_queues.py 24

Prints pairs/s of a 256-pair forced-20 queue (the timed step of bench.py, smaller)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the variable: undone below)
from locus_amd import capi  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "ctor"
host = bench.gen_pairs_host(256, 0, 64, 1563, 2.0)   # (worker processes: before the GPU runtime exists)
libc = ctypes.CDLL(None)
if mode == "ctor":
    os.environ.pop("GPU_MAX_HW_QUEUES", None)   # (unsetenv)
else:
    os.environ["GPU_MAX_HW_QUEUES"] = mode
libc.getenv.restype = ctypes.c_char_p
before = libc.getenv(b"GPU_MAX_HW_QUEUES")
capi.lib()   # dlopen: the constructor runs here
after = libc.getenv(b"GPU_MAX_HW_QUEUES")
ctx = capi.Context(0)
P = capi.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12)
S, T, _ = bench.make_pairs(ctx, host)
_, A = capi.align_batch_out(ctx, P, S, T, max_in_flight=512)
best = 0.0
for _ in range(4):
    for t in T:
        t.drop_index()
    ctx.synchronize()
    t0 = time.perf_counter()
    capi.align_batch_out(ctx, P, S, T, max_in_flight=512, aligned=A, raw=True)
    ctx.synchronize()
    best = max(best, len(S) / (time.perf_counter() - t0))
print("mode=%s env before load=%s after load=%s pairs/s=%.0f" % (mode, before, after, best))
