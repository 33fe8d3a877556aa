"""One multi-frame elevation-mapping session (row N3), replayed by the tests on the HIP library, on the sequential
restatement (oracle/elev_oracle.cpp) and on the reference's own gpu_process.cu built for the host (oracle/_ref/libref_elev.so)."""
import numpy as np


def frame_points(rng, n, pose_xy):
    """A terrain-like cloud in the sensor frame: points behind the robot (y < -1) survive the reference's filter."""
    x = rng.uniform(-6, 6, n).astype(np.float32)
    y = rng.uniform(-7, 2, n).astype(np.float32)
    z = (0.15 * np.sin(0.8 * (x + pose_xy[0])) + 0.1 * np.cos(1.1 * (y + pose_xy[1])) - 0.6 + rng.normal(0, 0.02, n)).astype(np.float32)
    bump = (np.abs(x - 2) < 0.4) & (np.abs(y + 4) < 0.4)
    z[bump] += 0.8
    return x, y, z


def session(m, rng, frames, L):
    out = []
    pose = np.array([0.0, 0.0, 0.9], np.float32)
    for k in range(frames):
        pose[:2] += rng.uniform(-0.5, 0.7, 2).astype(np.float32)
        out.append(("move", m.move(pose)))
        x, y, z = frame_points(rng, 6000, pose)
        yaw = 0.1 * k
        T = np.eye(4, dtype=np.float32)
        T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        T[:3, 3] = [pose[0], pose[1], 0.9]
        rv = np.diag([1e-4, 1e-4, 4e-4]).astype(np.float32)
        res = m.process_points(x, y, z, T, -2.0, 3.0, 0.02, 0.003, 0.01, [0.0, 0.0, 1.0], rv, np.eye(3), [0.0, 0.0, 1.0],
                               [[0, -0.2, 0.1], [0.2, 0, -0.05], [-0.1, 0.05, 0]])
        out.append(("points", res))
        n = x.size
        cr = rng.integers(0, 256, n); cg = rng.integers(0, 256, n); cb = rng.integers(0, 256, n)
        inten = rng.uniform(0, 1, n).astype(np.float32)
        m.fuse(res["map_index"], cr, cg, cb, inten, res["z_ts"], res["var"])
        m.mapvar_update(1e-4)
        out.append(("feature", m.map_feature()))
        m.raytracing()
        out.append(("layers", [m.layer(w) for w in range(5)]))
        if k == 2:
            out.append(("optmove", m.map_optmove(pose[:2] + 0.33, 0.05)))
        if k == 3:
            m.map_closeloop(pose[:2] - 0.41, -0.02)
            out.append(("frame", m.frame()))
    return out
