"""Deterministic synthetic scenes / scans / maps for the BASELINE.json configs.

The reference ships no data set and no registration fixtures, so
the benchmark and the parity tests run on an analytic scene that is ray-cast
with the reference's own sensor models (``src/lidar/lidar_model.cpp:24-30``
Velodyne_16-like 16 x 900 grid for config 1, ``:39-45`` Velodyne_64: 64 rings,
-24.9 deg + 0.4 deg * ring, 1800 azimuth steps of 0.2 deg).

Scene ("campus"): ground plane z = -1.8 (sensor/body origin 1.8 m above it), a jittered grid of axis-aligned building
boxes and square 0.6 x 0.6 x 6 m pillars.  SURVEY.md 8d proposed a single
80 x 50 x 12 m room; that room has ~8e3 m^2 of surface, i.e. 1e6 map points would
give ~30 points per 0.5 m iVox voxel, whereas the survey's own traffic model
(and a real iVox map after the reference's down-sampling insert rule) has
~4-5.  The campus has ~4.5e4 m^2 within the 100 m sensor range so the 1e6-point
map lands at ~5 points / voxel.  Everything else follows 8d: range gate
4..100 m, N(0, 0.02 m) range noise, N(0, 0.01 m) map noise, T_gt ~ U(+-2 deg,
+-0.3 m), initial guess = identity, seed = 20241022 + 1000 * config + job.

Host-side numpy only -- this is input generation, not part of the hot path.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20241022
SENSOR_Z = 0.0    # the sensor / body origin ...
GROUND_Z = -1.8   # ... rides 1.8 m above the ground plane (the reference's plane model A x = -1 cannot
                  # represent a plane through the world origin, loam_point_to_plane_ivox.h:275-284)
MAX_RANGE = 100.0
MIN_RANGE = 4.0


def rng_for(config_id: int, job: int = 0, salt: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.MT19937(BASE_SEED + 1000 * config_id + job + 7919 * salt))


# ---------------------------------------------------------------------------------------------
# scene
# ---------------------------------------------------------------------------------------------
def make_scene(seed: int = BASE_SEED) -> dict:
    """Axis-aligned boxes (B,6) = [xmin,ymin,zmin,xmax,ymax,zmax]; first `n_buildings` are buildings."""
    rng = np.random.Generator(np.random.MT19937(seed))
    boxes = []
    centers = np.arange(-90.0, 91.0, 36.0)
    for cx in centers:
        for cy in centers:
            jx, jy = rng.uniform(-5, 5, 2)
            hx, hy = rng.uniform(6, 10), rng.uniform(5, 9)
            h = rng.uniform(8, 14)
            x, y = cx + jx, cy + jy
            # keep a clear disc around the sensor
            if (abs(x) - hx) < 9.0 and (abs(y) - hy) < 9.0:
                continue
            boxes.append([x - hx, y - hy, GROUND_Z, x + hx, y + hy, GROUND_Z + h])
    n_buildings = len(boxes)
    pc = np.arange(-72.0, 73.0, 36.0)
    for cx in pc:
        for cy in pc:
            jx, jy = rng.uniform(-3, 3, 2)
            x, y = cx + jx, cy + jy
            if np.hypot(x, y) < 6.0:
                x += 8.0
            boxes.append([x - 0.3, y - 0.3, GROUND_Z, x + 0.3, y + 0.3, GROUND_Z + 6.0])
    return {"boxes": np.asarray(boxes, dtype=np.float64), "n_buildings": n_buildings}


def _ray_cast(scene: dict, o: np.ndarray, d: np.ndarray) -> np.ndarray:
    """Range along unit rays d (N,3) from common origin o (3,) to the nearest surface (inf = miss)."""
    boxes = scene["boxes"]
    n = d.shape[0]
    t_best = np.full(n, np.inf)
    # ground
    dz = d[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dz < -1e-12, (GROUND_Z - o[2]) / dz, np.inf)
    t_best = np.minimum(t_best, np.where(tg > 0, tg, np.inf))
    # boxes: slab method, chunked over rays
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
    chunk = 16384
    lo = boxes[:, 0:3][None, :, :]
    hi = boxes[:, 3:6][None, :, :]
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        iv = inv[s:e, None, :]
        t1 = (lo - o[None, None, :]) * iv
        t2 = (hi - o[None, None, :]) * iv
        tmin = np.nanmax(np.minimum(t1, t2), axis=2)
        tmax = np.nanmin(np.maximum(t1, t2), axis=2)
        hit = (tmax >= tmin) & (tmax > 0)
        tb = np.where(hit, np.where(tmin > 0, tmin, np.inf), np.inf).min(axis=1)
        t_best[s:e] = np.minimum(t_best[s:e], tb)
    return t_best


def _ray_dirs(rings: np.ndarray, az: np.ndarray, elev0_deg: float, elev_step_deg: float) -> np.ndarray:
    el = np.deg2rad(elev0_deg + elev_step_deg * rings)
    ce = np.cos(el)
    return np.stack([ce * np.cos(az), ce * np.sin(az), np.sin(el)], axis=1)


def cast_scan(scene: dict, T_gt: np.ndarray, n_rings: int, n_az: int, elev0_deg: float, elev_step_deg: float,
              rng: np.random.Generator, range_noise: float = 0.02, max_range: float = MAX_RANGE) -> np.ndarray:
    """Body-frame scan with exactly n_rings * n_az points (misses are re-cast with jitter)."""
    R, t = T_gt[:3, :3], T_gt[:3, 3]
    o_body = np.array([0.0, 0.0, SENSOR_Z])
    o_world = R @ o_body + t
    ring = np.repeat(np.arange(n_rings), n_az).astype(np.float64)
    az = np.tile(np.arange(n_az) * (2.0 * np.pi / n_az), n_rings)
    n = ring.size
    pts = np.zeros((n, 3))
    todo = np.arange(n)
    for attempt in range(64):
        d_body = _ray_dirs(ring[todo], az[todo], elev0_deg, elev_step_deg)
        r = _ray_cast(scene, o_world, d_body @ R.T)
        ok = np.isfinite(r) & (r >= MIN_RANGE) & (r <= max_range)
        rr = r[ok] + rng.normal(0.0, range_noise, ok.sum())
        pts[todo[ok]] = o_body[None, :] + rr[:, None] * d_body[ok]
        todo = todo[~ok]
        if todo.size == 0:
            break
        # re-cast: random lower ring + jittered azimuth
        ring[todo] = rng.integers(0, max(1, n_rings // 2), todo.size).astype(np.float64)
        az[todo] = rng.uniform(0.0, 2.0 * np.pi, todo.size)
    if todo.size:
        raise RuntimeError("scan generation did not converge")
    return pts.astype(np.float32)


# raw driver point of the reference (include/lidar/lidar_point_type.h VelodynePointXYZIRT, 32 bytes, 16-byte aligned)
RAW_POINT_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"],
                            "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                            "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32})


def cast_raw_scan(scene: dict, T_gt: np.ndarray, n_rings: int, n_az: int, elev0_deg: float, elev_step_deg: float,
                  rng: np.random.Generator, range_noise: float = 0.02, max_range: float = MAX_RANGE, dup_frac: float = 0.02,
                  drop_frac: float = 0.01) -> np.ndarray:
    """Raw driver cloud (RAW_POINT_DTYPE) in firing order: azimuth-major, all lasers of one firing together; misses
    are simply absent; `dup_frac` extra returns land in already occupied range-image cells (the projector keeps the
    first), `drop_frac` of the returns are removed at random (ragged rows), a few are outside [min, max] range."""
    R, t = T_gt[:3, :3], T_gt[:3, 3]
    o_body = np.array([0.0, 0.0, SENSOR_Z])
    o_world = R @ o_body + t
    ring = np.tile(np.arange(n_rings), n_az).astype(np.float64)
    step = 2.0 * np.pi / n_az
    az = np.repeat(np.arange(n_az) * step, n_rings) - np.pi + 0.25 * step  # firing angle, (-pi, pi)
    d_body = _ray_dirs(ring, az, elev0_deg, elev_step_deg)
    r = _ray_cast(scene, o_world, d_body @ R.T)
    ok = np.isfinite(r) & (r >= 0.5) & (r <= max_range * 1.05)
    ok &= rng.uniform(size=ok.size) >= drop_frac
    idx = np.nonzero(ok)[0]
    rr = r[idx] + rng.normal(0.0, range_noise, idx.size)
    pts = o_body[None, :] + rr[:, None] * d_body[idx]
    rg = ring[idx]
    tm = (az[idx] + np.pi) / (2.0 * np.pi) * 0.1
    # duplicates: a second return of the same firing direction, slightly farther, appended later in the stream
    nd = int(dup_frac * idx.size)
    if nd:
        pick = rng.choice(idx.size, nd, replace=False)
        extra = o_body[None, :] + (rr[pick] + rng.uniform(0.05, 1.0, nd))[:, None] * d_body[idx[pick]]
        pos = np.sort(rng.integers(0, idx.size, nd))
        pts = np.insert(pts, pos, extra, axis=0)
        rg = np.insert(rg, pos, rg[pick])
        tm = np.insert(tm, pos, tm[pick])
    out = np.zeros(pts.shape[0], dtype=RAW_POINT_DTYPE)
    out["x"], out["y"], out["z"] = pts[:, 0].astype(np.float32), pts[:, 1].astype(np.float32), pts[:, 2].astype(np.float32)
    out["intensity"] = rng.uniform(0.0, 255.0, pts.shape[0]).astype(np.float32)
    out["ring"] = rg.astype(np.uint16)
    out["time"] = tm.astype(np.float32)
    return out


def _surfaces(scene: dict):
    """List of vertical wall rectangles (origin, u, v, normal) for area sampling."""
    rects = []
    for b in scene["boxes"]:
        x0, y0, z0, x1, y1, z1 = b
        h = z1 - z0
        rects.append((np.array([x0, y0, z0]), np.array([x1 - x0, 0, 0]), np.array([0, 0, h]), np.array([0, -1.0, 0])))
        rects.append((np.array([x0, y1, z0]), np.array([x1 - x0, 0, 0]), np.array([0, 0, h]), np.array([0, 1.0, 0])))
        rects.append((np.array([x0, y0, z0]), np.array([0, y1 - y0, 0]), np.array([0, 0, h]), np.array([-1.0, 0, 0])))
        rects.append((np.array([x1, y0, z0]), np.array([0, y1 - y0, 0]), np.array([0, 0, h]), np.array([1.0, 0, 0])))
    return rects


def sample_map(scene: dict, n: int, rng: np.random.Generator, noise: float = 0.01, radius: float = MAX_RANGE) -> np.ndarray:
    """n world-frame points, area-uniform over ground disc (outside footprints) + all vertical faces."""
    rects = _surfaces(scene)
    areas = np.array([np.linalg.norm(np.cross(u, v)) for (_, u, v, _) in rects])
    boxes = scene["boxes"]
    foot = ((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1])).sum()
    ground_area = np.pi * radius * radius - foot
    if radius < MAX_RANGE:
        # wall area inside the radius (coarse estimate from rectangle centres) and footprints inside it
        cen = np.stack([r[0] + 0.5 * r[1] for r in rects])
        inside = np.hypot(cen[:, 0], cen[:, 1]) <= radius
        wall_area = float(areas[inside].sum())
        bc = 0.5 * (boxes[:, 0:2] + boxes[:, 3:5])
        foot_in = ((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]))[np.hypot(bc[:, 0], bc[:, 1]) <= radius].sum()
        ground_area = np.pi * radius * radius - foot_in
    else:
        wall_area = float(areas.sum())
    total = wall_area + ground_area
    n_ground = int(round(n * ground_area / total))
    n_wall = n - n_ground
    out = np.zeros((n, 3))
    # ground (rejection on footprints)
    filled = 0
    while filled < n_ground:
        m = int((n_ground - filled) * 1.4) + 16
        rr = radius * np.sqrt(rng.uniform(0, 1, m))
        th = rng.uniform(0, 2 * np.pi, m)
        x, y = rr * np.cos(th), rr * np.sin(th)
        inside = np.zeros(m, bool)
        for b in boxes:
            inside |= (x > b[0]) & (x < b[3]) & (y > b[1]) & (y < b[4])
        x, y = x[~inside], y[~inside]
        k = min(x.size, n_ground - filled)
        out[filled:filled + k, 0] = x[:k]
        out[filled:filled + k, 1] = y[:k]
        out[filled:filled + k, 2] = GROUND_Z + rng.normal(0.0, noise, k)
        filled += k
    # walls (rejection on the horizontal radius so that reduced-size configs keep the full-size density)
    O = np.stack([r[0] for r in rects]); U = np.stack([r[1] for r in rects])
    V = np.stack([r[2] for r in rects]); Nn = np.stack([r[3] for r in rects])
    filled = 0
    while filled < n_wall:
        m = int((n_wall - filled) * 1.5) + 64
        which = rng.choice(len(rects), size=m, p=areas / areas.sum())
        a = rng.uniform(0, 1, m)
        b_ = rng.uniform(0, 1, m)
        off = rng.normal(0.0, noise, m)
        w = O[which] + a[:, None] * U[which] + b_[:, None] * V[which] + off[:, None] * Nn[which]
        w = w[np.hypot(w[:, 0], w[:, 1]) <= radius]
        k = min(w.shape[0], n_wall - filled)
        out[n_ground + filled:n_ground + filled + k] = w[:k]
        filled += k
    perm = rng.permutation(n)
    return out[perm].astype(np.float32)


def sample_edges(scene: dict, n: int, rng: np.random.Generator, noise: float = 0.01) -> np.ndarray:
    """n world-frame points on the vertical edges of all boxes (LOAM corner map)."""
    boxes = scene["boxes"]
    edges = []
    for b in boxes:
        for (x, y) in ((b[0], b[1]), (b[0], b[4]), (b[3], b[1]), (b[3], b[4])):
            edges.append((x, y, b[2], b[5]))
    edges = np.asarray(edges)
    lens = edges[:, 3] - edges[:, 2]
    which = rng.choice(len(edges), size=n, p=lens / lens.sum())
    z = edges[which, 2] + rng.uniform(0, 1, n) * lens[which]
    out = np.stack([edges[which, 0], edges[which, 1], z], axis=1) + rng.normal(0, noise, (n, 3))
    return out.astype(np.float32)


def cast_edge_scan(scene: dict, T_gt: np.ndarray, n: int, rng: np.random.Generator, noise: float = 0.02) -> np.ndarray:
    """n body-frame points on vertical edges visible-ish from the sensor (LOAM corner cloud)."""
    boxes = scene["boxes"]
    R, t = T_gt[:3, :3], T_gt[:3, 3]
    o_world = R @ np.array([0.0, 0.0, SENSOR_Z]) + t
    pts = []
    need = n
    while need > 0:
        cand = sample_edges(scene, need * 3 + 64, rng, noise=0.0).astype(np.float64)
        v = cand - o_world[None, :]
        r = np.linalg.norm(v, axis=1)
        d = v / r[:, None]
        el = np.degrees(np.arcsin((d @ R)[:, 2]))  # elevation in the body frame
        ok = (r >= MIN_RANGE) & (r <= 60.0) & (el > -24.9) & (el < 0.3 + 5.0)
        # occlusion: the edge point must be (about) the first hit along its ray
        tt = _ray_cast(scene, o_world, d)
        ok &= tt >= r - 0.35
        cand = cand[ok][:need]
        pts.append(cand)
        need -= cand.shape[0]
    w = np.concatenate(pts)[:n] + rng.normal(0, noise, (n, 3))
    body = (w - t[None, :]) @ R  # R^T (w - t)
    return body.astype(np.float32)


# ---------------------------------------------------------------------------------------------
# poses
# ---------------------------------------------------------------------------------------------
def so3_exp(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    if th < 1e-16:
        return np.eye(3)
    a = w / th
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(a, a) + np.sin(th) * K


def random_pose(rng: np.random.Generator, max_rot_deg: float = 2.0, max_trans: float = 0.3) -> np.ndarray:
    w = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg, 3))
    t = rng.uniform(-max_trans, max_trans, 3)
    T = np.eye(4)
    T[:3, :3] = so3_exp(w)
    T[:3, 3] = t
    return T


def pose_error(Ta: np.ndarray, Tb: np.ndarray):
    """(translation error [m], rotation error [rad]) between two 4x4 poses."""
    dt = float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    Rd = Ta[:3, :3].T @ Tb[:3, :3]
    c = max(-1.0, min(1.0, (np.trace(Rd) - 1.0) / 2.0))
    s = np.linalg.norm([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]]) / 2.0
    return dt, float(np.arctan2(s, c))


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs
# ---------------------------------------------------------------------------------------------
VELODYNE_64 = dict(n_rings=64, n_az=1800, elev0_deg=-24.9, elev_step_deg=0.4)   # lidar_model.cpp:39-45
VELODYNE_16 = dict(n_rings=16, n_az=900, elev0_deg=-15.0, elev_step_deg=2.0)    # lidar_model.cpp:24-30 (900 az: config 1)


def make_config(config_id: int, job: int = 0, scale: float = 1.0, with_map: bool = True) -> dict:
    """Inputs of BASELINE.json configs[config_id] (0-based).  `scale` < 1 shrinks N and M for CPU tests.

    returns dict(scan, map, T_gt, T_init [, corner_scan, corner_map]); with_map = False leaves the (expensive) map clouds out
    (ranks that receive the map image from rank 0).
    """
    scene = make_scene()
    rng = rng_for(config_id, job)
    T_gt = random_pose(rng)
    out = {"T_gt": T_gt, "T_init": np.eye(4), "scene": scene}
    if config_id == 0:
        lid = dict(VELODYNE_16)
        m = int(50000 * scale)
    else:
        lid = dict(VELODYNE_64)
        m = int(1000000 * scale)
    radius = MAX_RANGE
    if scale < 1.0:
        # keep the full-size map density: shrink the mapped disc (and the sensor range) instead of thinning
        lid["n_az"] = max(36, int(lid["n_az"] * scale))
        radius = max(18.0, MAX_RANGE * float(np.sqrt(scale)))
    out["radius"] = radius
    out["scan"] = cast_scan(scene, T_gt, rng=rng, max_range=radius, **lid)
    if with_map:
        out["map"] = sample_map(scene, m, rng_for(config_id, 0, salt=1), radius=radius)  # the map is shared by all jobs of a config
    if config_id == 3:
        n_corner = max(64, int(7680 * scale))
        out["corner_scan"] = cast_edge_scan(scene, T_gt, n_corner, rng)
        if with_map:
            out["corner_map"] = sample_edges(scene, max(2000, int(100000 * scale)), rng_for(config_id, 0, salt=2))
        out["scan"] = out["scan"][::2].copy()  # 57,600 surf points
    return out
