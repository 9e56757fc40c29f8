"""GPU suite: liblocus_hip_rccl.so (include/locus_hip_rccl.h) -- the exchange steps of the multi-GPU path on RCCL itself, without
torch.distributed.  A 1-GPU box can only form a communicator of one rank (RCCL refuses two ranks on one device), which still
runs every call through librccl on the device; with >= 2 visible GPUs the same test body runs as two ranks (spawned)."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rank_body(rank, world, uid, device, q=None):
    from locus_amd import capi, rccl, synth
    from locus_amd import dist as ldist
    comm = rccl.Comm(device, uid, rank, world)
    ctx = capi.Context(device)
    out = {}
    # (1) independent pairs sharded over ranks: no data-path collective, results gathered with ONE all-gather of the records
    n_pairs = 5                                   # does not divide by 2: the ranks hold 3 and 2 pairs
    lo, hi = ldist.shard_range(n_pairs, rank, world)
    P = capi.default_params(max_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3)
    S, T = [], []
    for p in range(lo, hi):
        src, tgt, _ = synth.scan_pair(n_rings=16, n_az=300, scale=1.0, noise=0.01, seed=900 + 2 * p)
        cs, ct = capi.Cloud(ctx, src), capi.Cloud(ctx, tgt)
        cs.normals_knn(10)
        ct.normals_knn(10)
        S.append(cs)
        T.append(ct)
    raw, _ = capi.align_batch_out(ctx, P, S, T, raw=True) if S else ((capi.GicpResult * 1)(), [])
    local = (capi.GicpResult * max(len(S), 1))(*[raw[i] for i in range(len(S))])
    allr, counts = comm.allgather_results(local if S else [], n_pairs)
    out["counts"] = counts
    out["poses"] = np.array([allr[i].T[:] for i in range(n_pairs)], np.float32)
    out["mine"] = (lo, hi, np.array([raw[i].T[:] for i in range(len(S))], np.float32))
    # (2) one pair sharded by source points: the SUM hook on RCCL
    src, tgt, _ = synth.scan_pair(n_rings=16, n_az=600, scale=1.0, noise=0.01, seed=990)
    ns, nt = ctx.normals_knn(src, 20), ctx.normals_knn(tgt, 20)
    a, b = ldist.shard_range(len(src), rank, world)
    g = capi.Gicp(ctx, capi.default_params(max_iterations=10, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12))
    g.set_target(capi.make_pointf(tgt, nt))
    g.set_source(capi.make_pointf(src[a:b], ns[a:b]))
    comm.install_sum_hook(ctx)   # the host hook AND the device hook: the sharded pair runs the device-driven loop, its exchange stays in HBM
    ctx.profile(True)
    ctx.profile_reset()
    r = g.align(want_trace=False)
    stats = ctx.profile_get()
    ctx.profile(False)
    out["device_allreduces"] = (stats.get("moments_allreduce", {}).get("launches", 0), stats.get("bfgs_solve", {}).get("launches", 0), r["iterations"])
    fit = g.fitness()
    comm.remove_sum_hook(ctx)
    g.set_source(capi.make_pointf(src, ns))
    w = g.align(want_trace=False)
    out["sharded"] = (np.asarray(r["T"]), r["iterations"], fit, np.asarray(w["T"]), w["iterations"], g.fitness())
    out["tmax"] = comm.max_double(1.0 + rank)
    comm.barrier()
    comm.close()
    if q is not None:
        q.put((rank, out))
    return out


def _check(outs, world):
    n_pairs = 5
    for rank, o in outs.items():
        assert sum(o["counts"]) == n_pairs and len(o["counts"]) == world
        assert (o["poses"] == outs[0]["poses"]).all()                  # every rank holds the same gathered table
        lo, hi, mine = o["mine"]
        assert (o["poses"][lo:hi] == mine).all()                        # ... with its own block in rank order
        Ts, its, fs, Tw, itw, fw = o["sharded"]
        assert (Ts == outs[0]["sharded"][0]).all() and its == itw
        assert np.abs(Ts - Tw).max() < 1e-5 and abs(fs - fw) <= 1e-6 * abs(fw)
        if world == 1:
            assert (Ts == Tw).all() and fs == fw                        # a world of one: the hook is the identity
        n_red, n_solve, its_ = o["device_allreduces"]
        assert n_red >= its_ and n_red == n_solve        # one device-side all-reduce in front of every k_solve launch: no host copy in the loop
        assert o["tmax"] == float(world)


def test_rccl_exchange_steps():
    from locus_amd import capi, rccl
    ndev = capi.device_count()
    uid = rccl.unique_id()
    assert len(uid) == 128
    if ndev < 2:
        _check({0: _rank_body(0, 1, uid, 0)}, 1)
        return
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_body, args=(r, 2, uid, r, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _check(outs, 2)
