"""GPU suite: the reference's own gtest cases replayed against the C++ host mirror (locus_amd/host) on the C ABI."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_passes_the_reference_gtests():
    exe = os.path.join(ROOT, "locus_amd", "host", "host_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "locus_amd", "host")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "HOST_CHECK_OK" in out.stdout
