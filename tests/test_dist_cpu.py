"""CPU suite: the N>1 path of bench.py (pair sharding, pose all_gather, max-over-ranks timing, trajectory chaining)
under torch.distributed with the gloo backend, world_size 2."""
import os
import socket

import numpy as np
import pytest

from locus_amd import dist as ldist


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 16, 512, 513):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = ldist.shard_range(n, r, world)
                covered += list(range(lo, hi))
                assert hi - lo in (n // world, n // world + 1)
            assert covered == list(range(n))


def _fake_pose(i):
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.1 * i, -0.01 * i, 0.0]
    c, s = np.cos(0.01 * i), np.sin(0.01 * i)
    T[:2, :2] = [[c, -s], [s, c]]
    return np.ascontiguousarray(T.T).reshape(16)  # column-major like lh_gicp_result.T


def _worker(rank, world, port, n_pairs, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lo, hi = ldist.shard_range(n_pairs, rank, world)
    local = np.stack([_fake_pose(i) for i in range(lo, hi)]) if hi > lo else np.zeros((0, 16), np.float32)
    gathered = ldist.gather_poses(local, world)
    t = ldist.max_over_ranks(1.0 + rank, world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, np.concatenate(gathered), t))


@pytest.mark.parametrize("n_pairs", [5, 8])
def test_two_rank_gather_over_gloo(n_pairs):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([_fake_pose(i) for i in range(n_pairs)])
    for rank, allposes, t in res:
        assert np.array_equal(allposes, expect)  # every rank sees every pair, in pair order
        assert t == 2.0                          # max over ranks
    traj = ldist.chain_poses(res[0][1])
    assert traj.shape == (n_pairs + 1, 4, 4)
    assert np.allclose(traj[-1][:3, :3] @ traj[-1][:3, :3].T, np.eye(3), atol=1e-5)


def test_four_rank_gather_with_an_uneven_shard_over_gloo():
    """world size 4, 10 pairs: blocks of 3 / 3 / 2 / 2 -- the padded all_gather trims every rank's block to its own count, the table is in pair
    order on every rank, the timing is the slowest rank's"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_pairs, world = 10, 4
    assert [ldist.shard_range(n_pairs, r, world) for r in range(world)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([_fake_pose(i) for i in range(n_pairs)])
    assert sorted(r[0] for r in res) == [0, 1, 2, 3]
    for rank, allposes, t in res:
        assert np.array_equal(allposes, expect)
        assert t == 4.0


def test_strong_split_of_a_fixed_queue():
    """bench.py --strong (BASELINE configs[3]: 512 queued pairs over 8 GPUs): --pairs is the TOTAL, every rank takes shard_range's block, the
    blocks cover the queue exactly once whatever the rank count, and the per-GPU load the line reports is the largest block"""
    for total, world in ((512, 8), (512, 3), (10, 4), (7, 8)):
        blocks = [ldist.shard_range(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[k][1] == blocks[k + 1][0] for k in range(world - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
    assert [hi - lo for lo, hi in (ldist.shard_range(512, r, 8) for r in range(8))] == [64] * 8   # configs[3]'s per-GPU load
    # the script's own use of it (bench.py main(): pairs_here under --strong)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert "ldist.shard_range(args.pairs" in src and "args.strong" in src


def _records_worker(rank, world, port, q):
    import torch.distributed as dist
    from locus_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n = 3
    raw = (capi.GicpResult * n)()
    for i in range(n):
        for k in range(16):
            raw[i].T[k] = float(100 * rank + 10 * i + k)
        raw[i].iterations = 20 - i
        raw[i].n_correspondences_last = 1000 * rank + i
        raw[i].fitness = 0.25 * (rank + 1)
    allr = ldist.gather_records(raw, world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bytes(allr.numpy().tobytes()), bytes(bytearray(raw))))


def test_two_rank_result_record_gather_over_gloo():
    """bench.py --gpus N: the lh_gicp_result records of every rank in ONE all_gather (device tensors under RCCL, host tensors under
    gloo here): every rank ends with the same table, each rank's block at its own offset, bit for bit"""
    import ctypes as C
    import torch.multiprocessing as mp
    from locus_amd import capi
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_records_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in procs:
        rank, table, mine = q.get(timeout=120)
        res[rank] = (table, mine)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == res[0][1] + res[1][1]
    rec = (capi.GicpResult * 6).from_buffer_copy(res[0][0])
    assert C.sizeof(capi.GicpResult) == 96 and rec[4].T[3] == 113.0 and rec[4].iterations == 19 and rec[5].n_correspondences_last == 1002


def _hook_worker(rank, world, port, q):
    import ctypes as C
    import torch.distributed as dist
    from locus_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    hook = ldist.make_sum_hook(world)
    # drive it exactly the way the C side does: through the lh_allreduce_fn C callback on a raw double buffer
    def _cb(ptr, n, _user):
        hook(np.ctypeslib.as_array(ptr, shape=(n,)))
        return 0
    cb = capi.ALLREDUCE_FN(_cb)
    out = []
    for n in (74, 14, 2):   # moment sums, cost sums + count, fitness (sum, n)
        buf = (C.c_double * n)(*[(rank + 1) * (i + 0.5) for i in range(n)])
        assert cb(C.cast(buf, C.POINTER(C.c_double)), n, None) == 0
        out.append(np.array(buf[:]))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_source_shard_sum_hook_over_gloo():
    """the exchange step of the source-sharded single pair (lh_set_allreduce): every rank ends with the same sums"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hook_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        for n, got in zip((74, 14, 2), out):
            assert np.array_equal(got, np.array([3.0 * (i + 0.5) for i in range(n)]))  # (1 + 2) * (i + 0.5), exact


def test_bench_rank0_only_block_calls_no_collective():
    """bench.py's roofline / extra legs run on rank 0 only while the other ranks already wait at the final barrier: a
    collective inside that block (it once reached the pose all_gather through step()) deadlocks every N > 1 run."""
    import ast
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    def is_rank0(t):   # `rank == 0` or `rank == 0 and ...`
        if isinstance(t, ast.BoolOp) and isinstance(t.op, ast.And):
            t = t.values[0]
        return isinstance(t, ast.Compare) and isinstance(t.left, ast.Name) and t.left.id == "rank" and isinstance(t.ops[0], ast.Eq)
    blocks = [n for n in ast.walk(main) if isinstance(n, ast.If) and is_rank0(n.test)]
    assert blocks, "rank == 0 block not found"
    big = max(blocks, key=lambda n: n.end_lineno - n.lineno)
    assert big.end_lineno - big.lineno > 30  # the roofline leg, not the final print
    for call in [n for n in ast.walk(big) if isinstance(n, ast.Call)]:
        f = call.func
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name):
            assert f.value.id not in ("dist", "ldist"), "collective %s.%s inside the rank-0-only block (line %d)" % (f.value.id, f.attr, call.lineno)
        if isinstance(f, ast.Name) and f.id in ("step", "barrier"):
            assert f.id != "barrier", "barrier() inside the rank-0-only block"
            kw = {k.arg: k.value for k in call.keywords}
            assert "exchange" in kw and isinstance(kw["exchange"], ast.Constant) and kw["exchange"].value is False, \
                "step() inside the rank-0-only block must pass exchange=False (line %d)" % call.lineno
