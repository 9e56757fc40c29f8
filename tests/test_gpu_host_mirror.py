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


def test_locus_stream_device_resident_flow_equals_the_host_surface(tmp_path):
    """locus_amd/host/locus_stream: LOCUS's whole per-scan work (Locus.cc:450-520: odometry update, frame transforms, map neighbours, MeasurementUpdate,
    keyframe insertion) over a 12-scan stream, once with host clouds in and out of every call and once with everything between the two
    registrations resident in HBM: same poses, covariance, keyframes and map size, bit for bit (the tool's exit code says so)."""
    import json
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    from locus_amd import capi, synth
    exe = os.path.join(ROOT, "locus_amd", "host", "locus_stream")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "locus_amd", "host")])
    ctx = capi.Context(0)
    path = str(tmp_path / "scans.bin")
    with open(path, "wb") as f:
        n_scans = 12
        f.write(np.int32(n_scans).tobytes())
        for i in range(n_scans):
            pose = synth.pose_matrix(0.15 * i, 0.03 * np.sin(0.5 * i), 0.0, 0.0, 0.0, 0.01 * i)
            c = capi.Cloud(ctx, capi.make_pointxyzi(synth.scan(pose, 16, 900, (-15.0, 15.0), 1.0, 0.02, seed=700 + i))).voxel_grid(0.3)
            c.normals_knn(20)
            d = c.download()
            a = capi.make_pointf(np.stack([d["x"], d["y"], d["z"]], 1), np.stack([d["normal_x"], d["normal_y"], d["normal_z"]], 1))
            f.write(np.int32(len(a)).tobytes())
            f.write(a.tobytes())
    out = subprocess.run([exe, path, "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["device_equals_host_bit_for_bit"] is True
    assert r["host_surface"]["updates"] == n_scans - 1 and r["host_surface"]["keyframes"] == r["device_resident"]["keyframes"] >= 1
    assert abs(r["host_surface"]["integrated_translation"][0] - 0.15 * (n_scans - 1)) < 0.1   # the stream's simulated motion along x
