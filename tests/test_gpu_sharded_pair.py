"""GPU suite: ONE pair sharded by source points over ranks through lh_set_allreduce.  The box has a
single GPU, so both ranks share cuda:0 and the exchange step runs over gloo; on an 8-GPU node the same hook runs over RCCL
(locus_amd/dist.py::make_sum_hook(device="cuda")).  Every rank must return the same transform, and it must equal the
unsharded alignment up to the summation order of the 74 (or 14) double sums."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from locus_amd import capi, synth
    from locus_amd import dist as ldist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ctx = capi.Context(0)
    src, tgt, delta = synth.scan_pair(n_rings=32, n_az=1500, scale=2.0, noise=0.02, seed=77)
    ns, nt = ctx.normals_knn(src, 20), ctx.normals_knn(tgt, 20)     # normals BEFORE sharding: they need the whole cloud
    lo, hi = ldist.shard_range(len(src), rank, world)
    res = {}
    for mode in (1, 0):
        P = capi.default_params(max_iterations=12, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12, cost_mode=mode)
        g = capi.Gicp(ctx, P)
        g.set_target(capi.make_pointf(tgt, nt))
        g.set_source(capi.make_pointf(src[lo:hi], ns[lo:hi]))
        ctx.set_allreduce(ldist.make_sum_hook(world))
        r = g.align(want_trace=False)
        fit = g.fitness()
        ctx.set_allreduce(None)
        whole = None
        if rank == 0:   # the unsharded answer, same process, hook removed
            g.set_source(capi.make_pointf(src, ns))
            w = g.align(want_trace=False)
            whole = (np.asarray(w["T"]), w["iterations"], w["n_corr_last"], g.fitness())
        res[mode] = (np.asarray(r["T"]), r["iterations"], r["n_corr_last"], fit, r["status"], whole)
        g.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res, delta))


def test_source_sharded_pair_equals_whole():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in procs:
        rank, r, delta = q.get(timeout=600)
        res[rank] = r
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for mode in (1, 0):
        T0, it0, nc0, f0, st0, whole = res[0][mode]
        T1, it1, nc1, f1, st1, _ = res[1][mode]
        assert st0 == 0 and st1 == 0
        assert np.array_equal(T0, T1) and it0 == it1 and nc0 == nc1 and f0 == f1   # same sums -> same BFGS -> same bits
        Tw, itw, ncw, fw = whole
        assert nc0 == ncw and it0 == itw                # the correspondence count is an integer sum: exact
        assert np.abs(T0 - Tw).max() < 1e-5, (mode, np.abs(T0 - Tw).max())
        assert abs(f0 - fw) <= 1e-6 * abs(fw)            # the fitness of two transforms that differ by < 1e-5
        Tm = T0.reshape(4, 4).T
        assert np.abs(Tm[:3, 3] - delta[:3, 3]).max() < 0.05
