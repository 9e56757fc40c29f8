"""The cooperative all-walk sweep (k_sweep_coop, lh_kernels.hip; opt-in with LH_SWEEP_COOP=1) is held to the same bars as the default
k_sweep_fused: the sweep / alignment tests are run once more in a child process with the variable set (the library reads it once)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sweep_and_alignment_tests_pass_with_the_cooperative_sweep():
    env = dict(os.environ, LH_SWEEP_COOP="1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", "sweep or neighbour or align or trace or batch",
           os.path.join(ROOT, "tests", "test_gpu_kernels.py"), os.path.join(ROOT, "tests", "test_gpu_align.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail
