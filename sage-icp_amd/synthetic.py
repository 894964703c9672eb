"""Synthetic labelled LiDAR workloads for the registration hot path (bench.py, tests).

SemanticKITTI-like street scene sampled from analytic surfaces (SURVEY.md §8d / BASELINE.md §4):
ground split into road(40) / parking(44) / sidewalk(48) / terrain(72) strips around road centre
lines every 80 m, building(50) walls and fence(51) runs along them, vegetation(70) Gaussian blobs
with trunk(71) segments, pole(80)+sign(81), parked car(10) boxes, ~2 % unlabelled(0).  Class mix
follows ros/launch/semantic-kitti.yaml `content` roughly.  Surface noise N(0, 0.02 m).

  c2 (headline)  tile 200x200 m, map 1,000,000 pts (voxel 1.0, 20+20), scan 120,000 pts
  c4 (multi-GPU) tile 600x600 m, map 10,000,000 pts,               scan 500,000 pts
  c1 (plumbing)  map from ~35k pts (voxel 0.8), scan 10,000 pts

The map is whatever the semantic retention policy (VoxelHashMap.hpp:45-70) keeps of a point
stream, cut at exactly `map_points` retained points; the scan is an independent, range-weighted
sample with range in (5, 100) m, labels zeroed beyond 50 m (Preprocessing.cpp:177-178),
coordinates rounded to f32 then widened (ros/ros2/Utils.hpp:167-171), expressed in the sensor
frame of a ground-truth pose T_gt (so registering it from the identity guess must find T_gt).
"""
import numpy as np

ROAD_PITCH = 80.0
GROUND_Z = -1.73
BASIC_LABELS = (40, 44, 48, 49, 50, 70, 72)

# (label, mixture weight)
_CLASSES = [
    ("ground", 0.44), ("building", 0.13), ("fence", 0.07), ("vegetation", 0.27),
    ("trunk", 0.006), ("pole", 0.004), ("car", 0.045), ("other", 0.015),
]

SEED_C2 = 0x5A6E1C9


def _road_offset(y):
    """signed distance of y to the nearest road centre line (lines at y = 80 k)."""
    return (y + ROAD_PITCH / 2) % ROAD_PITCH - ROAD_PITCH / 2


def _line_index(y):
    return np.floor((y + ROAD_PITCH / 2) / ROAD_PITCH)


def sample_surfaces(rng, n, half):
    """n points (x, y, z, label) on the scene surfaces of a tile [-half, half]^2 (float64)."""
    w = np.array([c[1] for c in _CLASSES])
    w = w / w.sum()
    kind = rng.choice(len(_CLASSES), size=n, p=w)
    out = np.empty((n, 4))
    x = rng.uniform(-half, half, n)
    y = rng.uniform(-half, half, n)
    z = np.full(n, GROUND_Z)
    lab = np.zeros(n)

    # ---- ground: label by distance to the road centre line
    g = kind == 0
    d = np.abs(_road_offset(y))
    even = (_line_index(y) % 2) == 0
    gl = np.full(n, 72.0)
    gl[d < 9.0] = 48.0
    gl[(d < 6.5) & even] = 44.0
    gl[d < 4.0] = 40.0
    lab[g] = gl[g]

    nlines = int(np.floor(half / ROAD_PITCH)) * 2 + 1
    line_y = lambda m: ROAD_PITCH * (rng.integers(0, nlines, m) - nlines // 2)
    side = lambda m: rng.choice([-1.0, 1.0], m)

    # ---- buildings: walls along x at 15 m from each road line, 30 m on / 10 m off
    b = kind == 1
    m = int(b.sum())
    xb = rng.uniform(-half, half, m)
    xb = np.floor(xb / 40.0) * 40.0 + (xb % 40.0) * 0.75   # squeeze into the 30 m wall span
    x[b] = xb
    y[b] = line_y(m) + side(m) * 15.0
    z[b] = rng.uniform(GROUND_Z, 6.0, m)
    lab[b] = 50.0

    # ---- fence: 11 m from each road line, 1.5 m high
    f = kind == 2
    m = int(f.sum())
    y[f] = line_y(m) + side(m) * 11.0
    z[f] = rng.uniform(GROUND_Z, GROUND_Z + 1.5, m)
    lab[f] = 51.0

    # ---- vegetation blobs (and trunks) on a jittered 16 m grid, kept off the roads
    def blob_centres(m):
        gx = np.floor(rng.uniform(-half, half, m) / 16.0)
        gy = np.floor(rng.uniform(-half, half, m) / 16.0)
        # deterministic per-cell jitter so blobs are shared by every point of the cell
        h = (gx * 73856093.0 + gy * 19349663.0) % 1024.0
        cx = gx * 16.0 + 4.0 + (h % 32.0) / 4.0
        cy = gy * 16.0 + 4.0 + (np.floor(h / 32.0)) / 4.0
        off = _road_offset(cy)
        cy = np.where(np.abs(off) < 20.0, cy + np.sign(off + 1e-9) * (20.0 - np.abs(off)), cy)
        return cx, cy

    v = kind == 3
    m = int(v.sum())
    cx, cy = blob_centres(m)
    x[v] = cx + rng.normal(0, 1.2, m)
    y[v] = cy + rng.normal(0, 1.2, m)
    z[v] = 1.5 + rng.normal(0, 0.9, m)
    lab[v] = 70.0

    t = kind == 4
    m = int(t.sum())
    cx, cy = blob_centres(m)
    a = rng.uniform(0, 2 * np.pi, m)
    x[t] = cx + 0.15 * np.cos(a)
    y[t] = cy + 0.15 * np.sin(a)
    z[t] = rng.uniform(GROUND_Z, 1.0, m)
    lab[t] = 71.0

    # ---- poles every 25 m on the sidewalk, sign plate on top
    p = kind == 5
    m = int(p.sum())
    px = np.round(rng.uniform(-half, half, m) / 25.0) * 25.0
    a = rng.uniform(0, 2 * np.pi, m)
    x[p] = px + 0.08 * np.cos(a)
    y[p] = line_y(m) + side(m) * 7.5 + 0.08 * np.sin(a)
    zp = rng.uniform(GROUND_Z, 5.0, m)
    z[p] = zp
    is_sign = zp > 4.4
    lp = np.where(is_sign, 81.0, 80.0)
    x[p] = np.where(is_sign, px + rng.uniform(-0.3, 0.3, m), x[p])
    lab[p] = lp

    # ---- parked cars: 4.2 x 1.8 x 1.5 boxes every 12 m at 5.2 m from the line
    c = kind == 6
    m = int(c.sum())
    cx = np.round(rng.uniform(-half, half, m) / 12.0) * 12.0
    cyc = line_y(m) + side(m) * 5.2
    face = rng.integers(0, 3, m)
    u = rng.uniform(-0.5, 0.5, m)
    vv = rng.uniform(-0.5, 0.5, m)
    sgn = side(m)
    x[c] = cx + np.where(face == 0, sgn * 2.1, u * 4.2)
    y[c] = cyc + np.where(face == 1, sgn * 0.9, np.where(face == 0, u * 1.8, vv * 1.8))
    z[c] = GROUND_Z + np.where(face == 2, 1.5, (vv + 0.5) * 1.5)
    lab[c] = 10.0

    # ---- other: unlabelled clutter near the ground
    o = kind == 7
    m = int(o.sum())
    z[o] = GROUND_Z + np.abs(rng.normal(0, 0.5, m))
    lab[o] = 0.0

    # a little label noise: ~0.5 % of everything becomes unlabelled
    drop = rng.random(n) < 0.005
    lab[drop] = 0.0

    noise = rng.normal(0, 0.02, (n, 3))
    out[:, 0] = x + noise[:, 0]
    out[:, 1] = y + noise[:, 1]
    out[:, 2] = z + noise[:, 2]
    out[:, 3] = lab
    return out


def pose_from_rpy_t(rpy_deg, t):
    """pose[7] = (qx,qy,qz,qw,tx,ty,tz) from roll/pitch/yaw in degrees (ZYX) + translation."""
    r, p, y = np.deg2rad(rpy_deg)
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), \
        np.cos(y / 2), np.sin(y / 2)
    qw = cr * cp * cy + sr * sp * sy
    qx = sr * cp * cy - cr * sp * sy
    qy = cr * sp * cy + sr * cp * sy
    qz = cr * cp * sy - sr * sp * cy
    return np.array([qx, qy, qz, qw, t[0], t[1], t[2]], dtype=np.float64)


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def apply_pose(T, pts):
    out = np.array(pts, dtype=np.float64, copy=True)
    R = quat_to_mat(T[:4])
    out[:, :3] = pts[:, :3] @ R.T + T[4:]
    return out


def invert_pose(T):
    R = quat_to_mat(T[:4])
    q = np.array([-T[0], -T[1], -T[2], T[3]])
    return np.concatenate([q, -(R.T @ T[4:])])


T_GT_C2 = pose_from_rpy_t([0.1, 0.1, 1.0], [0.50, 0.10, 0.02])


def make_scan(rng, n, half, T_gt, max_range=100.0, min_range=5.0, label_max_range=50.0):
    """n scan points in the sensor frame of T_gt (sensor at T_gt's translation in the map)."""
    chunks = []
    have = 0
    origin = T_gt[4:]
    while have < n:
        s = sample_surfaces(rng, max(4 * (n - have), 4096), half)
        r = np.linalg.norm(s[:, :3] - origin, axis=1)
        keep = (r > min_range) & (r < max_range)
        # LiDAR-like density: thin out with range
        keep &= rng.random(len(s)) < np.clip((12.0 / np.maximum(r, 1e-3)) ** 1.5, 0.03, 1.0)
        s = s[keep]
        r = r[keep]
        s[r > label_max_range, 3] = 0.0
        chunks.append(s)
        have += len(s)
    world = np.concatenate(chunks)[:n]
    local = apply_pose(invert_pose(T_gt), world)
    local[:, :3] = local[:, :3].astype(np.float32).astype(np.float64)
    return np.ascontiguousarray(local)


def build_map_points(new_map, rng, half, map_points, batch=500_000):
    """Feed a point stream through `new_map().AddPoints` until the map holds exactly
    `map_points` points.  Returns (map, inserted_stream) — the stream prefix reproduces the same
    map through any implementation of the AddPoint policy (oracle or product)."""
    m = new_map()
    stream = []
    while m.size() < map_points:
        remaining = map_points - m.size()
        # each inserted point adds at most one retained point, so `remaining` cannot overshoot
        take = min(batch, remaining)
        s = sample_surfaces(rng, take, half)
        m.AddPoints(s)
        stream.append(s)
    assert m.size() == map_points
    return m, np.concatenate(stream)


WORKLOADS = {
    # name: tile half-size, map points, scan points, map voxel size
    "c1": dict(half=60.0, map_stream=35_000, scan=10_000, voxel=0.8, seed=0xC1),
    "c2": dict(half=100.0, map_points=1_000_000, scan=120_000, voxel=1.0, seed=SEED_C2),
    "c4": dict(half=300.0, map_points=10_000_000, scan=500_000, voxel=1.0, seed=0xC4),
    # c5: dense scan against a 0.1 m voxel map; the 27-voxel neighbourhood only
    # reaches 0.1-0.2 m, so the planted offset is centimetres and sigma is small
    "c5": dict(half=40.0, map_stream=3_000_000, scan=200_000, voxel=0.1, seed=0xC5,
               T_gt=([0.0, 0.0, 0.03], [0.03, 0.01, 0.0])),
}

PARAMS = {
    # sigma -> (max_correspondence_distance = 3 sigma, kernel = sigma / 3), sageICP.cpp:83-84
    "cold": dict(max_dist=6.0, kernel=2.0 / 3.0, sem_th=0.4),     # sigma = 2.0 (start-up)
    "steady": dict(max_dist=0.9, kernel=0.1, sem_th=0.4),         # sigma = 0.3
    # c5: sigma = 0.1; sem_th 0.8 as in ros/launch/odometry_360.launch.py:63, 1.0 = semantics off
    "dense": dict(max_dist=0.3, kernel=0.1 / 3.0, sem_th=0.8),
    "dense_nosem": dict(max_dist=0.3, kernel=0.1 / 3.0, sem_th=1.0),
}


def make_workload(name, new_map, scale=1.0):
    """Returns dict(map=<map built through new_map()>, scan=(n,4), T_gt=pose7, stream=points).
    `scale` < 1 shrinks map/scan counts (tests)."""
    w = WORKLOADS[name]
    rng = np.random.default_rng(w["seed"])
    half = w["half"]
    if "map_stream" in w:
        stream = sample_surfaces(rng, int(w["map_stream"] * scale), half)
        m = new_map()
        m.AddPoints(stream)
    else:
        m, stream = build_map_points(new_map, rng, half, int(w["map_points"] * scale))
    T_gt = pose_from_rpy_t(*w["T_gt"]) if "T_gt" in w else T_GT_C2
    scan = make_scan(rng, int(w["scan"] * scale), half, T_gt)
    return dict(map=m, scan=scan, T_gt=T_gt.copy(), stream=stream, voxel=w["voxel"])


def make_stream(seed, n_frames, points_per_frame=30000, half=120.0, step=(1.0, 0.0, 0.0),
                yaw_step_deg=0.5, max_range=100.0):
    """c3-style synthetic stream: a sensor advancing `step` metres and turning
    `yaw_step_deg` per frame through the street scene; each frame is a labelled scan in the sensor
    frame with KITTI-like content (x, y, z as f32 values, integer labels).  Returns
    (frames list, true poses list)."""
    rng = np.random.default_rng(seed)
    frames, poses = [], []
    T = pose_from_rpy_t([0, 0, 0], [-0.5 * n_frames * step[0], 0.0, 0.0])
    dT = pose_from_rpy_t([0, 0, yaw_step_deg], list(step))
    for _ in range(n_frames):
        frames.append(make_scan(rng, points_per_frame, half, T, max_range=max_range))
        poses.append(T.copy())
        # T <- T * dT
        R = quat_to_mat(T[:4])
        t = R @ dT[4:] + T[4:]
        a, b = T[:4], dT[:4]
        q = np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                      a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                      a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3],
                      a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])
        T = np.concatenate([q / np.linalg.norm(q), t])
    return frames, poses


# ---------------------------------------------------------------------------------------------------------------------
# A second scene family (round 6): the same street, but SEEN as a spinning 64-beam sensor sees it — rays cast from the
# sensor, so the density of a scan falls with range ring by ring, surfaces are occluded, and a map built from such scans
# is dense near the driven path and thin far from it.  (sample_surfaces / make_scan draw points on the surfaces directly,
# thinned by a range rule: every surface is seen from everywhere.)  Used to check that the library's break-points (lanes per
# query, compact-scan filter, flat order: capi_internal.h::icp_lw, capi_run.hip::wants_filter / wants_flat) — measured on
# the first family — hold on scans of LiDAR geometry: profiles/README.md, tests/test_gpu_parity.py.
def _blob_cells(lo, hi):
    """the vegetation blobs of sample_surfaces whose 16-m grid cells intersect [lo, hi]^2: (cx, cy) arrays"""
    g = np.arange(np.floor(lo / 16.0), np.floor(hi / 16.0) + 1.0)
    gx, gy = np.meshgrid(g, g, indexing="ij")
    gx, gy = gx.ravel(), gy.ravel()
    h = (gx * 73856093.0 + gy * 19349663.0) % 1024.0
    cx = gx * 16.0 + 4.0 + (h % 32.0) / 4.0
    cy = gy * 16.0 + 4.0 + (np.floor(h / 32.0)) / 4.0
    off = _road_offset(cy)
    cy = np.where(np.abs(off) < 20.0, cy + np.sign(off + 1e-9) * (20.0 - np.abs(off)), cy)
    return cx, cy


def make_ring_scan(rng, T, beams=64, az_steps=2048, max_range=100.0, min_range=5.0, label_max_range=50.0,
                   elev_deg=(2.0, -24.8), dropout=0.02):
    """One revolution of a 64-beam sensor (HDL-64E geometry: beams from +2 to -24.8 degrees, `az_steps` firings per turn)
    at pose T in the street scene, by ray casting: ground strips, building walls, fences, poles, trunks, vegetation blobs
    (spheres), parked cars (boxes) — nearest hit per ray.  Returns the hits in the SENSOR frame, (n, 4), fp32-rounded,
    labels zeroed beyond `label_max_range` (Preprocessing.cpp:177-178); rays without a hit inside (min, max) range drop out."""
    R = quat_to_mat(T[:4])
    o = np.asarray(T[4:], dtype=np.float64)
    az = (np.arange(az_steps) + rng.uniform(0, 1)) * (2 * np.pi / az_steps)
    el = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], beams))
    ca, sa = np.cos(az)[:, None], np.sin(az)[:, None]
    ce, se = np.cos(el)[None, :], np.sin(el)[None, :]
    dl = np.stack([ca * ce, sa * ce, np.broadcast_to(se, (az_steps, beams))], axis=-1)      # sensor frame
    d = dl @ R.T                                                                           # map frame
    tmin = np.full((az_steps, beams), np.inf)
    lab = np.zeros((az_steps, beams))
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]

    def take(t, label, ok):
        better = ok & (t > 0.0) & (t < tmin)
        tmin[better] = t[better]
        lab[better] = label[better] if isinstance(label, np.ndarray) else label

    with np.errstate(divide="ignore", invalid="ignore"):
        # ground, labelled by the strip it lies in
        t = (GROUND_Z - o[2]) / dz
        y = o[1] + t * dy
        dd = np.abs(_road_offset(y))
        even = (_line_index(y) % 2) == 0
        gl = np.full(y.shape, 72.0)
        gl[dd < 9.0] = 48.0
        gl[(dd < 6.5) & even] = 44.0
        gl[dd < 4.0] = 40.0
        take(t, gl, dz < 0.0)
        # walls (30 m on, 10 m off) and fences along every road line within reach
        k0 = np.round(o[1] / ROAD_PITCH)
        for k in (k0 - 1, k0, k0 + 1):
            for side in (-1.0, 1.0):
                yw = ROAD_PITCH * k + side * 15.0
                t = (yw - o[1]) / dy
                x = o[0] + t * dx
                z = o[2] + t * dz
                take(t, 50.0, ((x % 40.0) < 30.0) & (z > GROUND_Z) & (z < 6.0))
                yf = ROAD_PITCH * k + side * 11.0
                t = (yf - o[1]) / dy
                z = o[2] + t * dz
                take(t, 51.0, (z > GROUND_Z) & (z < GROUND_Z + 1.5))

    # small objects: only the rays whose azimuth can reach them are tested
    dxy = np.hypot(dx, dy)
    yaw = np.arctan2(R[1, 0], R[0, 0])

    def window(cx, cy, r):
        rel = np.array([cx - o[0], cy - o[1]])
        dist = np.hypot(*rel)
        if dist > max_range + r or dist < r + 0.5:
            return None
        a0 = np.arctan2(rel[1], rel[0]) - yaw
        half_w = np.arcsin(min(1.0, r / dist)) + 2 * np.pi / az_steps
        i0 = int(np.floor((a0 - half_w) / (2 * np.pi) * az_steps))
        i1 = int(np.ceil((a0 + half_w) / (2 * np.pi) * az_steps)) + 1
        return np.arange(i0, i1) % az_steps

    def cylinder(cx, cy, r, z0, z1, label_fn):
        idx = window(cx, cy, r)
        if idx is None:
            return
        ex, ey = dx[idx], dy[idx]
        fx, fy = o[0] - cx, o[1] - cy
        a = ex * ex + ey * ey
        b = 2 * (fx * ex + fy * ey)
        c = fx * fx + fy * fy - r * r
        disc = b * b - 4 * a * c
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (-b - np.sqrt(disc)) / (2 * a)
        z = o[2] + t * dz[idx]
        ok = (disc > 0) & (t > 0) & (z > z0) & (z < z1)
        sub_t, sub_l = tmin[idx], lab[idx]
        better = ok & (t < sub_t)
        sub_t[better] = t[better]
        sub_l[better] = label_fn(z[better])
        tmin[idx], lab[idx] = sub_t, sub_l

    def sphere(cx, cy, cz, r, label):
        idx = window(cx, cy, r)
        if idx is None:
            return
        f = o - np.array([cx, cy, cz])
        dd_ = d[idx]
        b = 2 * (dd_ @ f)
        c = f @ f - r * r
        disc = b * b - 4 * c
        with np.errstate(invalid="ignore"):
            t = (-b - np.sqrt(disc)) / 2
        ok = (disc > 0) & (t > 0)
        sub_t, sub_l = tmin[idx], lab[idx]
        better = ok & (t < sub_t)
        sub_t[better] = t[better]
        sub_l[better] = label
        tmin[idx], lab[idx] = sub_t, sub_l

    def box(cx, cy, hx, hy, z0, z1, label):
        idx = window(cx, cy, np.hypot(hx, hy))
        if idx is None:
            return
        lo = np.array([cx - hx, cy - hy, z0])
        hi = np.array([cx + hx, cy + hy, z1])
        dd_ = d[idx]
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo - o) / dd_
            t2 = (hi - o) / dd_
        tn = np.minimum(t1, t2).max(axis=-1)
        tf = np.maximum(t1, t2).min(axis=-1)
        ok = (tn < tf) & (tn > 0)
        sub_t, sub_l = tmin[idx], lab[idx]
        better = ok & (tn < sub_t)
        sub_t[better] = tn[better]
        sub_l[better] = label
        tmin[idx], lab[idx] = sub_t, sub_l

    reach = max_range
    for k in (k0 - 1, k0, k0 + 1):
        for side in (-1.0, 1.0):
            for px in np.arange(np.ceil((o[0] - reach) / 25.0), np.floor((o[0] + reach) / 25.0) + 1) * 25.0:
                cylinder(px, ROAD_PITCH * k + side * 7.5, 0.08, GROUND_Z, 5.0, lambda z: np.where(z > 4.4, 81.0, 80.0))
            for cxx in np.arange(np.ceil((o[0] - reach) / 12.0), np.floor((o[0] + reach) / 12.0) + 1) * 12.0:
                box(cxx, ROAD_PITCH * k + side * 5.2, 2.1, 0.9, GROUND_Z, GROUND_Z + 1.5, 10.0)
    bx, by = _blob_cells(min(o[0], o[1]) - reach, max(o[0], o[1]) + reach)
    for cx, cy in zip(bx, by):
        sphere(cx, cy, 1.5, 2.0, 70.0)
        cylinder(cx, cy, 0.15, GROUND_Z, 1.0, lambda z: np.full(z.shape, 71.0))

    hit = np.isfinite(tmin)
    t = tmin + rng.normal(0, 0.02, tmin.shape)
    hit &= (t > min_range) & (t < max_range) & (rng.random(tmin.shape) > dropout)
    pts = dl[hit] * t[hit][:, None]                      # sensor frame: range along the ray
    l = lab[hit]
    l[t[hit] > label_max_range] = 0.0
    l[rng.random(len(l)) < 0.005] = 0.0
    out = np.empty((len(pts), 4))
    out[:, :3] = pts.astype(np.float32).astype(np.float64)
    out[:, 3] = l
    return np.ascontiguousarray(out)


def make_ring_workload(new_map, n_map_scans=30, step=2.0, az_steps=2048, voxel=1.0, seed=0xA64, scan_az_steps=2048):
    """Ring family: a map built from `n_map_scans` revolutions taken every `step` metres along the road (inserted through
    new_map().AddPoints in the map frame, as the pipeline's Update does), and one more revolution from a pose between the
    last two, registered from a guess half a metre off.  Returns dict(map, scan, T_gt, stream, voxel) like make_workload."""
    rng = np.random.default_rng(seed)
    m = new_map()
    stream = []
    for k in range(n_map_scans):
        T = pose_from_rpy_t([0.0, 0.0, 0.3 * k], [step * k, 0.4 * np.sin(0.2 * k), 0.0])
        s = apply_pose(T, make_ring_scan(rng, T, az_steps=az_steps))
        m.AddPoints(s)
        stream.append(s)
    T_gt = pose_from_rpy_t([0.1, 0.1, 0.3 * (n_map_scans - 1.5) + 1.0], [step * (n_map_scans - 1.5) + 0.5, 0.1, 0.02])
    scan = make_ring_scan(rng, T_gt, az_steps=scan_az_steps)
    return dict(map=m, scan=scan, T_gt=T_gt.copy(), stream=np.concatenate(stream), voxel=voxel)
