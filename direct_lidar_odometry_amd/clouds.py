"""Deterministic synthetic LiDAR clouds for the NanoGICP hot path.

Scene: closed box room 40 x 30 x 6 m centred at the origin in x/y, sensor height
1.5 m above the floor, plus 6 axis-aligned box obstacles (seeded).  Clouds are
ray-cast from a sensor pose with Gaussian range noise (sigma = 1 cm; the noise is
mandatory — it removes exact distance ties and exactly-planar neighbourhoods,
SURVEY.md §7 "hard parts").

Sensor models
  * VLP-16 shaped : 16 rings, elevations -15..+15 deg step 2 deg, `cols` azimuth columns
                    (6250 -> 100 000 points, 625 -> 10 000 points)
  * OS1-128 shaped: 128 rings over +-22.5 deg, 1954 columns, truncated to 250 000

Everything is float32 xyz in metres, in the *sensor* frame unless stated.
Points are emitted azimuth-major (column by column, rings inside a column), the
order a spinning LiDAR driver delivers them.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

ROOM_HALF = np.array([20.0, 15.0], dtype=np.float64)  # 40 x 30 m footprint
FLOOR_Z = -1.5  # sensor origin is 1.5 m above the floor
CEIL_Z = 4.5    # 6 m room height
SCENE_SEED = 1234

# Ground-truth motion source -> target
GT_TRANSLATION = (0.30, 0.10, 0.02)
GT_RPY_DEG = (0.5, -0.3, 2.0)


def rpy_to_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def make_pose(t=(0.0, 0.0, 0.0), rpy_deg=(0.0, 0.0, 0.0)) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rpy_to_matrix(*[math.radians(a) for a in rpy_deg])
    T[:3, 3] = t
    return T


def gt_transform() -> np.ndarray:
    """4x4 float64 that maps source-frame points into the target frame."""
    return make_pose(GT_TRANSLATION, GT_RPY_DEG)


@dataclasses.dataclass(frozen=True)
class Scene:
    boxes_lo: np.ndarray  # (6,3)
    boxes_hi: np.ndarray  # (6,3)


def make_scene(seed: int = SCENE_SEED) -> Scene:
    rng = np.random.default_rng(seed)
    lo, hi = [], []
    for _ in range(6):
        edge = rng.uniform(1.0, 3.0, size=3)
        while True:
            cx = rng.uniform(-ROOM_HALF[0] + 3.0, ROOM_HALF[0] - 3.0)
            cy = rng.uniform(-ROOM_HALF[1] + 3.0, ROOM_HALF[1] - 3.0)
            if math.hypot(cx, cy) > 4.0:  # keep the sensor track clear
                break
        lo.append([cx - edge[0] / 2, cy - edge[1] / 2, FLOOR_Z])
        hi.append([cx + edge[0] / 2, cy + edge[1] / 2, FLOOR_Z + edge[2]])
    return Scene(np.array(lo), np.array(hi))


def _ray_dirs(rings: int, cols: int, elev_lo_deg: float, elev_hi_deg: float) -> np.ndarray:
    elev = np.radians(np.linspace(elev_lo_deg, elev_hi_deg, rings))
    azim = np.linspace(0.0, 2.0 * math.pi, cols, endpoint=False)
    az, el = np.meshgrid(azim, elev, indexing="ij")  # azimuth-major
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1)
    return d.reshape(-1, 3)


def _raycast(scene: Scene, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """Range along each unit ray from `origin` (inside the room) to the first surface."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs
        room_lo = np.array([-ROOM_HALF[0], -ROOM_HALF[1], FLOOR_Z])
        room_hi = np.array([ROOM_HALF[0], ROOM_HALF[1], CEIL_Z])
        t1 = (room_lo - origin) * inv
        t2 = (room_hi - origin) * inv
        t_exit = np.nanmin(np.maximum(t1, t2), axis=1)
        best = t_exit
        for lo, hi in zip(scene.boxes_lo, scene.boxes_hi):
            a = (lo - origin) * inv
            b = (hi - origin) * inv
            tn = np.nanmax(np.minimum(a, b), axis=1)
            tf = np.nanmin(np.maximum(a, b), axis=1)
            hit = (tn < tf) & (tn > 0.0)
            best = np.where(hit & (tn < best), tn, best)
    return best


def scan(scene: Scene, pose: np.ndarray, *, rings: int, cols: int, elev_deg: tuple[float, float],
         noise_seed: int, sigma: float = 0.01, limit: int | None = None) -> np.ndarray:
    """One LiDAR scan from world pose `pose` (4x4).  Returns float32 (N,3) in the SENSOR frame."""
    dirs_local = _ray_dirs(rings, cols, *elev_deg)
    if limit is not None:
        dirs_local = dirs_local[:limit]
    dirs_world = dirs_local @ pose[:3, :3].T
    rng_ = _raycast(scene, pose[:3, 3], dirs_world)
    noise = np.random.default_rng(noise_seed).normal(0.0, sigma, size=rng_.shape)
    pts = dirs_local * (rng_ + noise)[:, None]
    return np.ascontiguousarray(pts, dtype=np.float32)


def vlp16(scene: Scene, pose: np.ndarray, noise_seed: int, cols: int = 6250) -> np.ndarray:
    return scan(scene, pose, rings=16, cols=cols, elev_deg=(-15.0, 15.0), noise_seed=noise_seed)


def os1_128(scene: Scene, pose: np.ndarray, noise_seed: int, n: int = 250_000) -> np.ndarray:
    return scan(scene, pose, rings=128, cols=1954, elev_deg=(-22.5, 22.5), noise_seed=noise_seed, limit=n)


def transform_points(T: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """float32 rigid transform the way pcl::transformPointCloud does it (float matrix)."""
    Tf = T.astype(np.float32)
    return np.ascontiguousarray(pts @ Tf[:3, :3].T + Tf[:3, 3], dtype=np.float32)


def to_xyzi(pts: np.ndarray) -> np.ndarray:
    """Pack (N,3) float32 into the 32-byte pcl::PointXYZI layout: x y z 1 | intensity 0 0 0."""
    out = np.zeros((pts.shape[0], 8), dtype=np.float32)
    out[:, :3] = pts
    out[:, 3] = 1.0
    return out


@dataclasses.dataclass
class Workload:
    name: str
    source: np.ndarray           # (Ns,3) float32, source sensor frame
    target: np.ndarray           # (Nt,3) float32, target (world) frame
    guess: np.ndarray            # 4x4 float32 initial guess
    gt: np.ndarray               # 4x4 float64 ground truth source->target
    keyframe_sizes: list[int]    # target is a concatenation of keyframes with these sizes
    max_corr_dist: float


def _sensor(shape: str):
    if shape == "vlp16":
        return lambda sc, pose, seed, n: vlp16(sc, pose, seed, cols=n // 16)
    if shape == "os1":
        return lambda sc, pose, seed, n: os1_128(sc, pose, seed, n=n)
    raise ValueError(shape)


def scan_to_scan(n: int = 100_000, shape: str = "vlp16", seed_offset: int = 0) -> Workload:
    """BASELINE config 1 (n=10k) / config 2 (n=100k): target at the world origin, source displaced by GT."""
    sc = make_scene()
    mk = _sensor(shape)
    target_pose = np.eye(4)
    gt = gt_transform()
    source_pose = target_pose @ gt
    src = mk(sc, source_pose, 1 + seed_offset, n)
    tgt = mk(sc, target_pose, 100 + seed_offset, n)
    return Workload(f"s2s_{shape}_{n}", src, tgt, np.eye(4, dtype=np.float32), gt, [tgt.shape[0]], 1.0)


def scan_to_submap(n_src: int = 100_000, n_keyframes: int = 5, shape: str = "vlp16", seed_offset: int = 0,
                   n_per_keyframe: int | None = None) -> Workload:
    """BASELINE config 3 (100k vs 5x100k) / config 5 (shape='os1', 250k vs 8x250k).

    Keyframes are scanned from poses 2 m apart along x, transformed to the world
    frame and concatenated in order — DLO's submap assembly
    (/root/reference/src/dlo/odom.cc:1166-1174,1318-1325).  The source is scanned
    from world pose GT (relative to keyframe 0); the guess is GT perturbed by
    (0.05 m, 0.5 deg), mimicking T_s2s (odom.cc:837).
    """
    sc = make_scene()
    mk = _sensor(shape)
    npk = n_per_keyframe or n_src
    kfs = []
    for i in range(n_keyframes):
        pose = make_pose(((i - (n_keyframes - 1) / 2.0) * 2.0, 0.0, 0.0))
        local = mk(sc, pose, 100 + i + seed_offset, npk)
        kfs.append(transform_points(pose, local))
    target = np.ascontiguousarray(np.concatenate(kfs, axis=0))
    gt = gt_transform()
    src = mk(sc, gt, 1 + seed_offset, n_src)
    guess = (gt @ make_pose((0.05, -0.03, 0.02), (0.2, -0.3, 0.5))).astype(np.float32)
    return Workload(f"s2m_{shape}_{n_src}_{target.shape[0]}", src, target, guess, gt, [k.shape[0] for k in kfs], 0.5)


def pose_error(T: np.ndarray, T_ref: np.ndarray) -> tuple[float, float]:
    """(translation error [m], rotation angle error [rad]) between two 4x4 transforms."""
    T = np.asarray(T, dtype=np.float64)
    T_ref = np.asarray(T_ref, dtype=np.float64)
    dt = float(np.linalg.norm(T[:3, 3] - T_ref[:3, 3]))
    dR = T[:3, :3] @ T_ref[:3, :3].T
    c = max(-1.0, min(1.0, (np.trace(dR) - 1.0) / 2.0))
    # use the skew part for small angles (acos is ill-conditioned near 1)
    s = 0.5 * np.linalg.norm([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    return dt, float(math.atan2(s, c))
