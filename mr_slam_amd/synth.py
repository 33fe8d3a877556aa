"""Deterministic synthetic scans for tests and bench.py.

No dataset ships with the reference (one NCLT scan only), so measurements use seeded
synthetic clouds of the BASELINE size (120 000 points per scan):
  * lidar_scan : a 64-beam spinning lidar ray-cast against a piecewise-planar scene (ground
    plane + random boxes within +-70 m), sigma = 2 cm range noise, then the reference
    pre-processing crop/scale of RING_ros/util.py:91-112 (|x|,|y| < 70, 0 < z < 30,
    divided by 70/70/30).  Points stay in acquisition order (beam-major), like a real scan.
  * uniform_scan: uniform in [-1,1]^3 (stress case; what BASELINE.md section 3 timed).
"""
import numpy as np

N_POINTS = 120_000


def uniform_scan(seed, n=N_POINTS, zmax=0.999):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    p[:, 2] *= zmax
    return p


def _raycast(seed, n_az, n_beams=64, sensor_h=1.73, n_boxes=40, extent=70.0):
    rng = np.random.default_rng(seed)
    az = (np.arange(n_az) + rng.uniform(0, 1)) * (2 * np.pi / n_az)
    el = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    elg, azg = np.meshgrid(el, az, indexing="ij")                 # beam-major
    d = np.stack([np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], -1).reshape(-1, 3)
    t = np.full(d.shape[0], np.inf)
    down = d[:, 2] < -1e-6
    t[down] = -sensor_h / d[down, 2]                              # ground plane z = -sensor_h
    c = rng.uniform(-extent, extent, size=(n_boxes, 2))
    c = c[np.hypot(c[:, 0], c[:, 1]) > 4.0]
    half = rng.uniform(1.0, 8.0, size=(c.shape[0], 2))
    top = rng.uniform(1.5, 12.0, size=c.shape[0]) - sensor_h
    lo = np.concatenate([c - half, np.full((c.shape[0], 1), -sensor_h)], 1)
    hi = np.concatenate([c + half, top[:, None]], 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        for b in range(lo.shape[0]):                              # slab test, origin = 0
            t0 = lo[b] * inv
            t1 = hi[b] * inv
            tn = np.nanmax(np.minimum(t0, t1), axis=1)
            tf = np.nanmin(np.maximum(t0, t1), axis=1)
            hit = (tf >= np.maximum(tn, 0.0)) & (tn > 0.5)
            t = np.where(hit & (tn < t), tn, t)
    ok = np.isfinite(t) & (t < 100.0)
    r = t[ok] + rng.normal(0.0, 0.02, size=int(ok.sum()))
    p = d[ok] * r[:, None]
    p[:, 2] += sensor_h                                            # ground at z ~ 0
    return p


def preprocess(pc):
    """Crop/scale exactly like load_pc_infer (RING_ros/util.py:91-112)."""
    pc = np.asarray(pc, dtype=np.float32)
    keep = (np.abs(pc[:, 0]) < 70.0) & (np.abs(pc[:, 1]) < 70.0) & (pc[:, 2] < 30.0) & (pc[:, 2] > 0.0)
    hits = pc[keep]
    hits[:, 0] = hits[:, 0] / 70.0
    hits[:, 1] = hits[:, 1] / 70.0
    hits[:, 2] = hits[:, 2] / 30.0
    return hits


def lidar_scan(seed, n=N_POINTS, metric=False):
    """n pre-processed points (float32 [n,3], normalised like the reference) or, with
    metric=True, the same points in metres (for GICP)."""
    n_az = 2600
    while True:
        p = _raycast(seed, n_az)
        keep = (np.abs(p[:, 0]) < 70.0) & (np.abs(p[:, 1]) < 70.0) & (p[:, 2] < 30.0) & (p[:, 2] > 0.0)
        p = p[keep]
        if p.shape[0] >= n:
            break
        n_az = int(n_az * 1.3) + 1
    idx = np.floor(np.arange(n) * (p.shape[0] / n)).astype(np.int64)   # order-preserving thinning
    p = p[idx].astype(np.float32)
    return p if metric else preprocess(p)


def to_soa(pc):
    """[n,3] -> the reference SoA [x.., y.., z..] float32 (util.py:177)."""
    return np.ascontiguousarray(np.asarray(pc, dtype=np.float32)[:, :3].T).reshape(-1)
