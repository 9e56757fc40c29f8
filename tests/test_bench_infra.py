"""CPU suite: the checker-side plumbing of bench.py -- the CPU farm that runs the oracle's untimed concurrent alignments in spawned worker
processes (unbound, with the affinity mask the bench process started with) gives the same bits as the same oracle call in this process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _download_like(oracle, pts):
    n4 = oracle.normals_knn(oracle.xyz4(pts), 20, threads=2)
    return {"x": pts[:, 0], "y": pts[:, 1], "z": pts[:, 2], "normal_x": n4[:, 0], "normal_y": n4[:, 1], "normal_z": n4[:, 2]}


def test_cpu_farm_matches_the_in_process_oracle(oracle, monkeypatch):
    for k in ("OMP_PROC_BIND", "OMP_PLACES", "GOMP_SPINCOUNT", "GPU_MAX_HW_QUEUES"):   # importing bench.py sets its defaults: not for the rest of this test session
        monkeypatch.setenv(k, os.environ[k]) if k in os.environ else monkeypatch.delenv(k, raising=False)
    saved = {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "GOMP_SPINCOUNT", "GPU_MAX_HW_QUEUES")}
    import bench
    from locus_amd import synth
    src, tgt, _ = synth.scan_pair(n_rings=16, n_az=300, scale=1.0, noise=0.01, seed=5)
    a, b = _download_like(oracle, src), _download_like(oracle, tgt)
    kw = dict(max_iterations=6, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-3, rotation_epsilon=2e-3, gicp_epsilon=1e-3)
    ref = bench.InProcessFarm(2, threads=2)
    rs, rt = ref.put("s", a), ref.put("t", b)
    want = ref.align([(rs, rt)], kw)[0]
    want_fit = ref.fitness([(rs, rt, want["T"])])[0]
    farm = bench.CpuFarm(2, threads=2)
    try:
        ks, kt = farm.put("s", a), farm.put("t", b)
        assert os.path.exists(ks + "_xyz.npy")
        got = farm.align([(ks, kt), (ks, kt), (ks, kt)], kw)
        for g in got:
            assert g["status"] == 0 and g["iterations"] == want["iterations"] and (g["T"] == want["T"]).all()
        assert farm.fitness([(ks, kt, want["T"])])[0] == want_fit
        farm.drop(ks)
        assert not os.path.exists(ks + "_xyz.npy")
    finally:
        farm.close()
    assert not [f for f in os.listdir(farm.dir) if f.startswith("lhbench_%d_" % os.getpid())]
    # the environment of this process is what it was: the binding variables are only withheld from the children
    assert os.environ.get("LH_BENCH_WORKER") is None
    for k, v in saved.items():   # (what bench.py's import added goes away with the test)
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
