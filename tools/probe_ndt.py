import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from locus_amd import capi, synth
ctx = capi.Context(0)
delta = synth.pose_matrix(0.04, -0.03, 0.01, 0.002, -0.001, 0.006)
src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10, delta=delta)
P = capi.default_ndt_params(resolution=1.0, transformation_epsilon=1e-3, max_iterations=30)
ndt = capi.Ndt(ctx, P)
cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
ndt.set_source(cs); ndt.set_target(ct); r = ndt.align()
for name, fn in (("grid build (set_target + cells)", lambda: (ndt.set_target(ct), ndt.cells())), ("align without rebuild", lambda: ndt.align()),
                 ("one derivative evaluation (hessian)", lambda: ndt.derivatives(np.zeros(6))), ("one evaluation (gradient only)", lambda: ndt.derivatives(np.zeros(6), want_h=False))):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    ctx.synchronize()
    print("%-40s %.3f ms" % (name, 1e2 * (time.perf_counter() - t0)))
print(r["iterations"], r["evaluations"])
