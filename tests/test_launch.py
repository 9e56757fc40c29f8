"""`python bench.py --gpus N` must come up as N ranks or fail: the launcher decision on the CPU, and the
real self-launch through torch.distributed.run with two gloo ranks sharing the one GPU of the test box."""
import json
import os
import subprocess
import sys

import pytest

from locus_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_gpu_needs_no_launcher():
    assert launch.self_launch_command("bench.py", ["--gpus", "1"], 1, {}) is None
    assert launch.self_launch_command("bench.py", [], 1, {"WORLD_SIZE": "1"}) is None


def test_plain_command_launches_its_ranks():
    cmd = launch.self_launch_command("/x/bench.py", ["--gpus", "4", "--steps", "2"], 4, {}, executable="/usr/bin/python3", port=29555)
    assert cmd == ["/usr/bin/python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "29555", "/x/bench.py", "--gpus", "4", "--steps", "2"]
    # a free port is chosen when none is given
    cmd = launch.self_launch_command("bench.py", ["--gpus", "2"], 2, {})
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536


def test_rank_of_a_matching_launch_runs_in_place():
    assert launch.self_launch_command("bench.py", ["--gpus", "8"], 8, {"WORLD_SIZE": "8", "RANK": "3"}) is None


@pytest.mark.parametrize("gpus,world", [(8, "1"), (8, "4"), (1, "2"), (2, "8")])
def test_rank_count_mismatch_is_a_hard_failure(gpus, world):
    with pytest.raises(launch.LaunchError):
        launch.self_launch_command("bench.py", [], gpus, {"WORLD_SIZE": world})


def test_check_world():
    launch.check_world(2, 2, 2)
    launch.check_world(1, 1, 1)
    for bad in ((2, 1, 1), (2, 2, 1), (1, 2, 2)):
        with pytest.raises(launch.LaunchError):
            launch.check_world(*bad)


def test_mismatch_fails_before_any_gpu_work():
    """the real script: --gpus 8 under a WORLD_SIZE=1 environment exits non-zero with the reason, needing neither torch nor a GPU"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--quick"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_plain_command_gpus_2_yields_a_two_rank_line():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE: two ranks come up (gloo, both on cuda:0: the test box has one GPU), the
    records all_gather and the max-over-ranks timing run, the line says n_gpus 2 and carries the strong-scaling reading"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--quick",
                        "--pairs", "8", "--in-flight", "8", "--steps", "1", "--warmup", "1", "--rings", "16", "--azimuths", "450"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_world"] == 2 and line["self_launched"] is True
    assert line["strong_scaling_same_pairs"]["total_pairs"] == 8 and line["strong_scaling_same_pairs"]["pairs_per_gpu"] == 4
    assert line["value"] > 0
    # configs[4]'s multi-GPU form rides along at N > 1: the source-sharded dense pair, every rank ending with the same transform
    sp = line["config5_sharded_pair"]
    assert "error" not in sp, sp
    assert sp["status"] == 0 and sp["all_ranks_same_transform_bit_for_bit"] is True and sp["translation_err_vs_truth_m"] < 0.05
    assert 0 < sp["source_points_this_rank"] < sp["points_after_voxel_grid"]
