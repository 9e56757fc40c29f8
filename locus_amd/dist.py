"""Multi-GPU plumbing of the hot path: independent scan pairs are sharded over ranks (one process per GPU) with NO
data-path collective; the only exchanges are an all_gather of the 16-float poses and a MAX all-reduce of the elapsed time
(SURVEY.md 8e).  The single-huge-pair case (config 5) shards SOURCE points instead and has one real exchange step, the
SUM all-reduce of make_sum_hook.  Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs under "gloo" on CPU in the tests."""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block of items for `rank` (blocks differ by at most one item): scan i serves as query then as
    reference on the same device, only one boundary scan is duplicated per block."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_poses(local_poses, world, device=None):
    """all_gather of per-pair 4x4 poses: local (k,16) float32 -> list of per-rank arrays, identical on every rank.
    Ranks may hold different k: padded to the max and trimmed after the gather."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local_poses, np.float32).reshape(-1, 16)
    if world == 1:
        return [local]
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    kmax = int(max(int(c.item()) for c in cnts))
    buf = torch.zeros((kmax, 16), dtype=torch.float32, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(local).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o[: int(c.item())].cpu().numpy() for o, c in zip(out, cnts)]


def gather_records(raw, world, device=None):
    """ONE all_gather of fixed-size result records (a ctypes array of lh_gicp_result, 96 B each; every rank holds the same
    number) on device tensors: the only exchange of the pair-sharded path.  Returns the uint8 tensor of all
    ranks' records in rank order (left on `device`: nobody needs them on the host inside the timed step)."""
    import torch
    import torch.distributed as dist
    loc = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if device is not None:
        loc = loc.to(device)
    if world == 1:
        return loc
    allr = torch.empty(world * loc.numel(), dtype=torch.uint8, device=loc.device)
    dist.all_gather_into_tensor(allr, loc)
    return allr


def max_over_ranks(value, world, device=None):
    import torch
    import torch.distributed as dist
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def make_sum_hook(world, device=None):
    """The one real exchange step of the path (one huge pair sharded by source points): in-place SUM
    all-reduce of the 74 moment sums (or 14 cost sums) every rank computed over its own slice of the source cloud.
    Returns fn(float64 numpy view) for Context.set_allreduce.  gloo reduces the host buffer directly; under nccl (RCCL)
    the 592 bytes take a round trip through a small device tensor."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return lambda buf: None
    staging = {}

    def hook(buf):
        if device is None:
            t = torch.from_numpy(buf)     # shares memory with the C buffer
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            n = buf.shape[0]
            t = staging.get(n)
            if t is None:
                t = staging[n] = torch.zeros(n, dtype=torch.float64, device=device)
            t.copy_(torch.from_numpy(buf))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            buf[:] = t.cpu().numpy()
    return hook


def chain_poses(poses16):
    """SE(3) prefix product of incremental poses (PoseUpdate chaining, PointCloudOdometry.cc:308-309): returns (k+1,4,4)."""
    T = np.eye(4)
    out = [T.copy()]
    for p in np.asarray(poses16, np.float64).reshape(-1, 16):
        T = T @ p.reshape(4, 4).T  # column-major 16 floats -> matrix
        out.append(T.copy())
    return np.stack(out)
