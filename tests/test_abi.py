"""CPU suite: the C-ABI library loads, exports every symbol include/locus_hip.h declares, and refuses to run
without a GPU (no CPU fallback).  No device compute is attempted here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "locus_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    L = capi.lib()
    syms = _header_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == syms
    assert L.lh_abi_version() == 1


def test_library_exports_no_function_but_the_c_abi(capi):
    """the TUs are compiled -fvisibility=hidden: the only FUNCTIONS liblocus_hip.so exports are the header's (kernel handles are data symbols)"""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    funcs = [ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T"]
    assert sorted(funcs) == _header_symbols(), sorted(set(funcs) ^ set(_header_symbols()))


def test_rccl_library_exports_every_declared_symbol(capi):
    """liblocus_hip_rccl.so (include/locus_hip_rccl.h): loads next to librccl and exports what its header declares"""
    from locus_amd import rccl
    txt = open(os.path.join(ROOT, "include", "locus_hip_rccl.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    syms = sorted(set(re.findall(r"\b(lh_rccl_[a-z0-9_]+)\s*\(", txt)))
    L = rccl.lib()
    assert len(syms) == 11 and sorted(rccl.EXPORTS) == syms
    assert not [s for s in syms if not hasattr(L, s)]
    assert rccl.ID_BYTES == 128
    if not _has_gpu():   # argument validation happens before any device or RCCL call
        h = C.c_void_p()
        assert L.lh_rccl_create(0, b"\0" * 128, 2, 2, C.byref(h)) == capi.LH_EINVAL      # rank outside the world
        assert L.lh_rccl_create(0, b"\0" * 128, 0, 1, C.byref(h)) == capi.LH_EDEVICE      # no GPU here
        assert L.lh_rccl_install_sum_hook(None, None) == capi.LH_EINVAL


def test_struct_layouts_match_header(capi):
    assert C.sizeof(capi.CloudView) == 32
    assert C.sizeof(capi.GicpParams) == 72
    assert C.sizeof(capi.GicpResult) == 96
    assert capi.POINT_XYZI.itemsize == 32 and capi.POINT_XYZINORMAL.itemsize == 48


def test_default_params_are_the_class_defaults(capi):
    p = capi.default_params()  # gicp.h:111-132
    assert (p.max_iterations, p.max_inner_iterations, p.k_correspondences) == (200, 20, 20)
    assert (p.corr_dist, p.transformation_epsilon, p.rotation_epsilon, p.gicp_epsilon) == (5.0, 5e-4, 2e-3, 1e-3)
    assert p.recompute_source_cov == 0 and p.recompute_target_cov == 0


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(capi):
    with pytest.raises(capi.LocusHipError) as e:
        capi.Context(0)
    assert e.value.status == capi.LH_EDEVICE


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_multi_device_entry_points_without_a_gpu(capi):
    """the multi-GPU entry points load, validate their arguments and report "no device" without touching a GPU"""
    L = capi.lib()
    assert capi.device_count() == 0
    P = capi.default_params()
    assert L.lh_gicp_align_batch_multi(0, None, C.byref(P), 0, None, None, None, None, None, 0) == capi.LH_EINVAL
    assert L.lh_gicp_align_batch_multi_views(1, None, C.byref(P), 0, None, None, None, None, 0) == capi.LH_EINVAL
    assert L.lh_gicp_align_batch_out(None, C.byref(P), 0, None, None, None, None, None, 0) == capi.LH_EINVAL


def test_missing_library_fails_loudly():
    code = ("import sys; sys.path.insert(0, %r); from locus_amd import capi; capi.LIB_PATH = '/nonexistent/liblocus_hip.so'; "
            "capi.lib()" % ROOT)
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr.replace("\n", " ")


def test_runtime_init_is_explicit_and_keeps_a_chosen_value():
    """The scheduler's sixteen streams need more than the HIP runtime's default of four hardware queues, and the variable is read at the
    process's first HIP call.  LOADING the library must not touch the environment (round 4's constructor did); lh_runtime_init() is the
    explicit form and leaves a value the deployment chose alone; the Python package sets the default before anything can initialise HIP."""
    lib = os.path.join(ROOT, "locus_amd", "csrc", "liblocus_hip.so")
    code = ("import ctypes, os, sys\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "L = ctypes.CDLL(%r)\n"
            "a = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "rc = L.lh_runtime_init(int(sys.argv[1]))\n"
            "b = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "print(a.decode() if a else 'unset', rc, b.decode() if b else 'unset')" % lib)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run(["python", "-c", code, "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.split() == ["unset", "0", "24"], (r.stdout, r.stderr)       # loading alone: untouched; init(0): 24
    r = subprocess.run(["python", "-c", code, "16"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.split() == ["unset", "0", "16"], (r.stdout, r.stderr)
    r = subprocess.run(["python", "-c", code, "0"], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="6"))
    assert r.returncode == 0 and r.stdout.split() == ["6", "0", "6"], (r.stdout, r.stderr)            # the deployment's value stays
    r = subprocess.run(["python", "-c", code, "1000"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.split() == ["unset", "-1", "unset"], (r.stdout, r.stderr)   # LH_EINVAL
    # the Python package sets the default before anything can have initialised HIP
    code2 = "import os, sys; sys.path.insert(0, %r); import locus_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % ROOT
    r = subprocess.run(["python", "-c", code2], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.strip() == "24", (r.stdout, r.stderr)


@pytest.mark.gpu
def test_runtime_info_measures_the_stream_concurrency():
    """lh_runtime_info in fresh processes: with the package's default of 24 hardware queues the sixteen probe kernels overlap (adequate); with
    the runtime's default of 4 they run about four deep and the report says so -- the knob is observable, not an article of faith."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from locus_amd import capi\n"
            "import json; print(json.dumps(capi.Context(0).runtime_info()))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    import json
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    hi = json.loads(r.stdout.strip().splitlines()[-1])
    assert hi["hw_queues_env"] == 24 and hi["streams_probed"] == 16 and hi["adequate"] and hi["stream_concurrency"] >= 12.0, hi
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="4"), timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    lo = json.loads(r.stdout.strip().splitlines()[-1])
    assert lo["hw_queues_env"] == 4 and not lo["adequate"] and lo["stream_concurrency"] < 8.0, lo
    print("stream concurrency: 24 queues %.1f, 4 queues %.1f" % (hi["stream_concurrency"], lo["stream_concurrency"]))


def test_host_side_covariance_conditioning_matches_oracle(capi, oracle):
    # lh_icp_covariance is 6x6 host arithmetic (H2) -- callable without a device
    rng = np.random.default_rng(5)
    for trial in range(20):
        A = rng.normal(size=(40, 6))
        Ap = A.T @ A * rng.choice([1.0, 1e3, 1e6])
        ok1, c1, k1 = capi.icp_covariance(Ap, 0.01)
        ok2, c2, k2 = oracle.icp_covariance(Ap, 0.01)
        assert ok1 == ok2
        assert np.allclose(c1, c2, rtol=1e-9, atol=1e-18)
        assert abs(k1 - k2) <= 1e-6 * abs(k2)
    ok1, c1, _ = capi.icp_covariance(np.zeros((6, 6)), 0.01)  # singular: NaN guard -> diag(upper)
    ok2, c2, _ = oracle.icp_covariance(np.zeros((6, 6)), 0.01)
    assert not ok1 and not ok2 and np.allclose(c1, np.eye(6) * 0.01) and np.allclose(c2, c1)


def test_bfgs_host_solver_matches_oracle_on_a_quadratic():
    # the product's host BFGS (lh_bfgs.hpp) and the oracle's are separate restatements of pcl::BFGS; a tiny C++
    # driver minimises the same GICP cost through both and must agree (see tests/host_emu/bfgs_check.cpp)
    exe = "/tmp/lh_bfgs_check"
    src = os.path.join(ROOT, "tests", "host_emu", "bfgs_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", src, os.path.join(ROOT, "oracle", "locus_oracle.c"),
                           "-x", "none", "-fopenmp", "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BFGS_CHECK_OK" in out.stdout


def test_traversal_device_functions_on_host():
    exe = "/tmp/lh_traversal_check"
    src = os.path.join(ROOT, "tests", "host_emu", "traversal_check.cpp")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "TRAVERSAL_CHECK_OK" in out.stdout


def test_per_point_terms_match_the_reference_formula_on_oracle_covariances():
    """lh_terms.hpp (cost_mode 1: rank-one Mahalanobis matrix with one reciprocal, residual terms, the pose's doubles -- shared by the fused
    sweep, k_late and k_walk) against M = (C2 + R C1 R^T)^-1 (gicp.hpp:488-493) on the oracle's covariances from normals: unit,
    unnormalised, zero and non-finite normals, float rotations (tests/host_emu/terms_check.cpp)."""
    obj, oo, exe = "/tmp/lh_terms_check.o", "/tmp/lh_terms_oracle.o", "/tmp/lh_terms_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-c", os.path.join(ROOT, "oracle", "locus_oracle.c"), "-o", oo])
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-c",
                           os.path.join(ROOT, "tests", "host_emu", "terms_check.cpp"), "-o", obj])
    subprocess.check_call(["hipcc", obj, oo, "-fopenmp", "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "TERMS_CHECK_OK" in out.stdout, out.stdout + out.stderr


def test_block_knn_search_on_host():
    """The block k-NN search of K3 (lh_knn_block.hpp: one wave per 64 Morton-consecutive queries, sorted-key lists, window + one shared
    tree walk, second pass, redo list) stepped lane by lane on the host with the product's own per-lane functions and networks: every
    query's k-NN list (indices and float distances) against an exhaustive search -- tiny clouds, k < K, duplicates, runs of 50
    identical points, a lattice (ties everywhere: the redo path), 20 000 points"""
    exe = "/tmp/lh_knn_block_model"
    src = os.path.join(ROOT, "tools", "model", "knn_block_model.cpp")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "KNN_BLOCK_MODEL_OK" in out.stdout


def test_knn_networks_are_the_generators_output():
    """lh_knn_net.hpp is generated (tools/gen_knn_net.py checks every network before writing it): the committed header is what the
    generator writes today"""
    hdr = os.path.join(ROOT, "locus_amd", "csrc", "lh_knn_net.hpp")
    before = open(hdr).read()
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "gen_knn_net.py")], stdout=subprocess.DEVNULL)
    assert open(hdr).read() == before


def test_host_pool_runs_every_index_exactly_once():
    """HostPool (lh_runtime.hpp) resumes the alignment coroutines of a scheduler group on a few host threads: over 100 000 back-to-back
    parallel_for calls of changing size every index must run exactly once (a worker still leaving the previous call must never take
    an index of the next one)"""
    exe = "/tmp/lh_hostpool_check"
    src = os.path.join(ROOT, "tests", "host_emu", "hostpool_check.cpp")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-pthread", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "HOSTPOOL_CHECK_OK" in out.stdout


def test_start_grid_walks_on_a_scan_pair_host_model(tmp_path):
    """tools/model/grid_start_model.cpp drives the product's own start-grid functions (grid_fill_child / grid_start / node_visit /
    scan_leaf, lh_device.hpp) on the host over a lidar-like scan pair at four poses from the cold guess to the true motion: every
    neighbour must equal the walk from the root, and every certificate bound must be a bound (never below the winner, never above the
    exhaustive runner-up on the sampled queries).  The GPU runs the same functions; this is their CPU check on scan geometry (surfaces,
    queries outside the target's box, stale candidates)."""
    import numpy as np
    from scipy.spatial.transform import Rotation as R
    from locus_amd import synth
    src, tgt, delta = synth.scan_pair(n_rings=24, n_az=500, scale=2.0, noise=0.02, seed=77)
    src.astype(np.float32).tofile(tmp_path / "src.f32")
    tgt.astype(np.float32).tofile(tmp_path / "tgt.f32")
    D = np.asarray(delta, np.float64)
    poses = []
    for f in (0.0, 0.7, 0.95, 1.0):
        T = np.eye(4)
        T[:3, 3] = f * D[:3, 3]
        T[:3, :3] = R.from_rotvec(R.from_matrix(D[:3, :3]).as_rotvec() * f).as_matrix()
        poses.append(T[:3, :4])
    np.stack(poses).astype(np.float32).tofile(tmp_path / "poses.f32")
    exe = "/tmp/lh_grid_start_model"
    src_cpp = os.path.join(ROOT, "tools", "model", "grid_start_model.cpp")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", src_cpp, "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    sweeps = [l for l in out.stdout.splitlines() if l.startswith("sweep ")]
    assert len(sweeps) == 4, out.stdout
    for l in sweeps:
        assert "mismatches 0, bad certificate bounds 0" in l, l


def test_work_item_searches_host_model_is_exact(tmp_path):
    """tools/model/item_stack_model.cpp (round 6): the four restructurings of the sweeps' exact 1-NN search that were priced before any kernel was
    written -- the wave-shared walk, item stacks, per-lane walks over a shared stack, the lockstep first descent + items that became k_sweep_coop --
    each run on the host with the product's own grid_start / boxd2_q / leaf arithmetic over a scan pair at three poses: whatever the ORDER of the
    visits, every scheme must end with the neighbours of the product's walk (the model's own step counts are only as good as that)."""
    import numpy as np
    from scipy.spatial.transform import Rotation as R
    from locus_amd import synth
    src, tgt, delta = synth.scan_pair(n_rings=16, n_az=400, scale=2.0, noise=0.02, seed=31)
    src.astype(np.float32).tofile(tmp_path / "src.f32")
    tgt.astype(np.float32).tofile(tmp_path / "tgt.f32")
    D = np.asarray(delta, np.float64)
    poses = []
    for fr in (0.0, 0.8, 1.0):
        T = np.eye(4)
        T[:3, 3] = fr * D[:3, 3]
        T[:3, :3] = R.from_rotvec(R.from_matrix(D[:3, :3]).as_rotvec() * fr).as_matrix()
        poses.append(T[:3, :4])
    np.stack(poses).astype(np.float32).tofile(tmp_path / "poses.f32")
    exe = "/tmp/lh_item_stack_model"
    src_cpp = os.path.join(ROOT, "tools", "model", "item_stack_model.cpp")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", src_cpp, "-o", exe])
    out = subprocess.run([exe, str(tmp_path), "128"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("sweep ")]
    assert len(lines) >= 3 * 5, out.stdout   # (A), (B), (C) x 2, (D) per sweep (+ the cold variant of (D) in sweep 0)
    for l in lines:
        assert l.rstrip().endswith("mismatches 0"), l


def test_ndt_host_algebra_matches_oracle(oracle):
    """lh_ndt_host.hpp (pose <-> matrix, the 6x6 SVD solve of the Newton step) against the oracle's restatement of the same
    pclomp pieces; the `oracle` fixture makes sure oracle/liblocus_oracle.so is built"""
    exe = "/tmp/lh_ndt_host_check"
    src = os.path.join(ROOT, "tests", "host_emu", "ndt_host_check.cpp")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", src, "-L", odir, "-llocus_oracle",
                           "-Wl,-rpath," + odir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "NDT_HOST_CHECK_OK" in out.stdout, out.stdout + out.stderr


def test_adaptive_voxelization_controller_on_host():
    """Locus::ApplyAdaptiveInputVoxelization (Locus.cc:780-810) restated in locus_amd/host/AdaptiveVoxelization.hpp: plain C++"""
    exe = "/tmp/lh_controller_check"
    src = os.path.join(ROOT, "tests", "host_emu", "controller_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "CONTROLLER_CHECK_OK" in out.stdout, out.stdout + out.stderr


def test_pcd_io_and_pointcloud2_mapping_on_host(oracle):
    """PCD v0.7 reader/writer (Locus.cc:749-751; the GICP test's fixtures) and the sensor_msgs/PointCloud2 <->
    lh_cloud_view mapping of the filter nodelets' boundary; the reader must agree with the oracle's independent reader"""
    import numpy as np
    exe = "/tmp/lh_io_check"
    src = os.path.join(ROOT, "tests", "host_emu", "io_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe])
    for name, n in (("query_82_garage.pcd", 811), ("reference_82_garage.pcd", 8112)):
        fixture = os.path.join(ROOT, "tests", "golden", name)
        out = subprocess.run([exe, fixture, "/tmp/lh_io_check_tmp.pcd"], capture_output=True, text=True)
        assert out.returncode == 0 and "IO_CHECK_OK" in out.stdout, out.stdout + out.stderr
        line = [l for l in out.stdout.splitlines() if l.startswith("POINTS")][0].split()
        assert int(line[1]) == n
        ref = oracle.read_pcd_xyzi(fixture).astype(np.float64).sum(0)
        assert np.allclose([float(v) for v in line[3:7]], ref, rtol=1e-8)
