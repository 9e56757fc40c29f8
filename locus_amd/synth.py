"""Synthetic Velodyne-style scans for tests and bench (SURVEY.md 8d input table).

Scene S0: a room 20 x 12 x 4 m (six planes) with four vertical cylinders (r = 0.4 m), optionally
scaled; a spinning lidar (rings x azimuths) ray-casts it analytically from a given sensor pose and
returns points in the SENSOR frame with Gaussian range noise.  Pure numpy, seeded, no I/O.
"""
import numpy as np

ROOM_HALF = np.array([10.0, 6.0])        # x, y half extents
ROOM_Z = np.array([-1.5, 2.5])           # floor / ceiling (sensor ~1.5 m above the floor)
CYLS = np.array([[4.0, 2.5], [-5.0, 3.0], [6.0, -3.5], [-3.0, -4.0]])
CYL_R = 0.4


def rot_zyx(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def pose_matrix(tx=0.0, ty=0.0, tz=0.0, roll=0.0, pitch=0.0, yaw=0.0):
    T = np.eye(4)
    T[:3, :3] = rot_zyx(roll, pitch, yaw)
    T[:3, 3] = [tx, ty, tz]
    return T


def _raycast(o, d, scale):
    """o (3,), d (n,3) unit, world frame.  Returns range t (n,) and world normal (n,3)."""
    n = d.shape[0]
    best = np.full(n, np.inf)
    nrm = np.zeros((n, 3))
    hx, hy = ROOM_HALF * scale
    z0, z1 = ROOM_Z * scale
    planes = [(0, hx), (0, -hx), (1, hy), (1, -hy), (2, z1), (2, z0)]
    with np.errstate(divide="ignore", invalid="ignore"):
        for ax, val in planes:
            t = (val - o[ax]) / d[:, ax]
            ok = (t > 1e-6) & (t < best)
            best = np.where(ok, t, best)
            nv = np.zeros(3)
            nv[ax] = -np.sign(val - o[ax])
            nrm[ok] = nv
        for c in CYLS * scale:
            r = CYL_R * scale
            oc = o[:2] - c
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            b = 2 * (oc[0] * d[:, 0] + oc[1] * d[:, 1])
            cc = oc @ oc - r * r
            disc = b * b - 4 * a * cc
            t = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
            ok = (disc > 0) & (t > 1e-6) & (t < best)
            best = np.where(ok, t, best)
            hit = o[None, :] + t[:, None] * d
            nn = np.zeros((n, 3))
            nn[:, :2] = (hit[:, :2] - c) / r
            nrm[ok] = nn[ok]
    return best, nrm


def scan(pose=None, rings=64, azimuths=1563, elev_deg=(-25.0, 15.0), scale=2.0, noise=0.02, seed=0,
         with_normals=False):
    """One lidar sweep from sensor pose `pose` (4x4, sensor->world).  Returns float32 (n,3) points in the
    sensor frame (and analytic unit normals in the sensor frame if with_normals)."""
    pose = np.eye(4) if pose is None else np.asarray(pose, float)
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], rings))
    az = np.linspace(0.0, 2 * np.pi, azimuths, endpoint=False)
    E, A = np.meshgrid(el, az, indexing="ij")
    d_s = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    R, o = pose[:3, :3], pose[:3, 3]
    d_w = d_s @ R.T
    t, n_w = _raycast(o, d_w, scale)
    t = t + rng.normal(0.0, noise, t.shape)
    pts = (d_s * t[:, None]).astype(np.float32)
    if with_normals:
        return pts, (n_w @ R).astype(np.float32)
    return pts


def scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10, delta=None, elev_deg=(-25.0, 15.0)):
    """Config-2 style pair: target = sweep at the origin pose (seed), source = sweep from pose `delta`
    (seed+1).  GICP(source -> target) should recover ~delta.  delta=None draws the SURVEY 8d perturbation."""
    rng = np.random.default_rng(seed + 7919)
    if delta is None:
        delta = pose_matrix(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.05, 0.05),
                            np.deg2rad(rng.uniform(-0.5, 0.5)), np.deg2rad(rng.uniform(-0.5, 0.5)),
                            np.deg2rad(rng.uniform(-3, 3)))
    tgt = scan(np.eye(4), n_rings, n_az, elev_deg, scale, noise, seed)
    src = scan(delta, n_rings, n_az, elev_deg, scale, noise, seed + 1)
    return src, tgt, delta


def config1_pair():
    """BASELINE configs[0] / SURVEY 8d row 1: VLP-16 pattern (16 rings -15..+15 deg x 1800 azimuths = 28.8 k rays) in the
    unscaled room, range noise 0.02 m, seeds 1 / 2; the source is the same scene seen from
    delta = (0.20, -0.10, 0.02 m; yaw 2 deg, pitch 0.3 deg).  Voxelised at 0.25 m it is the ~5 k-point plumbing case."""
    delta = pose_matrix(0.20, -0.10, 0.02, 0.0, np.deg2rad(0.3), np.deg2rad(2.0))
    return scan_pair(n_rings=16, n_az=1800, scale=1.0, noise=0.02, seed=1, delta=delta, elev_deg=(-15.0, 15.0))


def hollow_cube(nx=10, ny=10, nz=10, step=0.1):
    """GenerateHollowCubic of the reference's odometry test (test_point_cloud_odometry.cpp:60-79): the four
    side walls of a 10x10x10 lattice (no top/bottom)."""
    pts = []
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                if ix == 0 or iy == 0 or ix == nx - 1 or iy == ny - 1:
                    pts.append((np.float32(ix) * np.float32(step), np.float32(iy) * np.float32(step),
                                np.float32(iz) * np.float32(step)))
    return np.array(pts, np.float32)


def plane_grid(nx=10, ny=10, step=0.1, z=0.0):
    """GeneratePlane-style fixture of the reference's localization tests: nx x ny lattice, normal (0,0,1)."""
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    pts = np.stack([ix.ravel() * step, iy.ravel() * step, np.full(ix.size, z)], -1).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pts.shape[0], 1))
    return pts, nrm


def husky_extrinsics():
    """body -> sensor extrinsics of a three-lidar platform (top / front tilted down / rear tilted, husky4_sensors.yaml style)"""
    return [pose_matrix(0, 0, 0.3), pose_matrix(0.4, 0, 0.0, pitch=0.35), pose_matrix(-0.4, 0, 0.0, pitch=-0.35, yaw=np.pi)]


def multi_lidar_parts(body_pose, extrinsics, rings=128, azimuths=2604, seed=0, scale=2.0, noise=0.02):
    """one scan per sensor, each expressed in the BODY frame (what point_cloud_merger receives);
    3 x 128 x 2604 ~ 1.0 M points"""
    parts = []
    for k, e in enumerate(extrinsics):
        pts = scan(body_pose @ e, rings, azimuths, (-25.0, 15.0), scale, noise, seed=seed + k)
        parts.append((pts.astype(np.float64) @ e[:3, :3].T + e[:3, 3]).astype(np.float32))
    return parts
