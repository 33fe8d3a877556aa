"""Throughput of the rows outside the headline step (development aid; numbers quoted in DESIGN.md section 4):
D1/D2 DiSCO descriptors + phase correlation, C3/C4 translation, N1 point features, N2 pre-processing,
N3 elevation-map frame update, N4 signature search / calcRelOri."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mr_slam_amd import bev, ring, disco, pointfeat, preprocess, elevation, synth

dev = "cuda:0"


def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


B = 256
base = [synth.lidar_scan(s) for s in range(4)]
xyz, offs = bev.pack_scans([base[i % 4] for i in range(B)], dev)
t = timeit(lambda: disco.disco_descriptors(xyz, offs, 40, 120, 20))
print(f"D1 DiSCO descriptors (polar BEV 40x120x20 + fft2 + signature): {B/t:,.0f} scans/s")
sig, spec = disco.disco_descriptors(xyz, offs, 40, 120, 20)
t = timeit(lambda: disco.phase_corr(spec, spec.roll(1, 0)))
print(f"D2 phase correlation: {B/t:,.0f} pairs/s")
t = timeit(lambda: disco.calc_rel_ori(spec, spec.roll(1, 0)))
print(f"N4 calcRelOri (double, literal): {B/t:,.0f} pairs/s")
db = torch.randn((100000, sig.shape[1]), device=dev)
t = timeit(lambda: disco.signature_search(sig, db))
print(f"N4 signature search: {B} queries x 100k signatures in {1e3*t:.2f} ms = {B*100000/t/1e9:.1f} G pairs/s")

_, sino, norm = ring.ring_descriptors(xyz, offs)
img = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1).reshape(B, 1, 120, 120)
rot = torch.zeros(1, device=dev)
t = timeit(lambda: [ring.solve_translation(sino[i:i + 1], sino[(i + 1) % B:(i + 1) % B + 1], 0.0, device=dev) for i in range(16)])
print(f"C3 solve_translation (drop-in, one pair per call): {16/t:,.0f} pairs/s")
t = timeit(lambda: [ring.solve_translation_bev(img[i], img[(i + 1) % B]) for i in range(16)])
print(f"C4 solve_translation_bev (drop-in, one pair per call): {16/t:,.0f} pairs/s")

n_sc = 8
pts = torch.from_numpy(np.concatenate([synth.lidar_scan(s, metric=True) for s in range(n_sc)])).to(dev)
o = np.arange(n_sc + 1, dtype=np.int64) * 120000
t = timeit(lambda: pointfeat.point_features(pts, o, want=("features", "planes")), n=3, warm=1)
print(f"N1 point features (kNN k=30 + eigen + 13 features): {n_sc/t:,.1f} clouds/s of 120k points")
raw = torch.from_numpy(np.concatenate([synth.lidar_scan(s, metric=True) * 1.0 for s in range(n_sc)])).to(dev)
t = timeit(lambda: preprocess.load_pc_infer_batch(raw, o))
print(f"N2 crop/scale: {n_sc*120000/t/1e9:.2f} G points/s")
t = timeit(lambda: preprocess.voxel_down_sample(raw[:120000], 0.5))
print(f"N2 voxel_down_sample(0.5) of 120k points: {1e3*t:.3f} ms")

m = elevation.ElevationMap(200, 0.1)
rng = np.random.default_rng(0)
n = 120000
x = rng.uniform(-9, 9, n).astype(np.float32); y = rng.uniform(-9, -1.2, n).astype(np.float32)
z = (0.1 * np.sin(x) - 0.6 + rng.normal(0, 0.02, n)).astype(np.float32)
T = np.eye(4, dtype=np.float32); T[2, 3] = 0.9
rv = np.diag([1e-4, 1e-4, 4e-4]).astype(np.float32)
cr = rng.integers(0, 256, n); inten = rng.uniform(0, 1, n).astype(np.float32)


def frame():
    m.move(np.array([0.0, 0.0, 0.9], np.float32))
    r = m.process_points(x, y, z, T, -2.0, 3.0, 0.02, 0.003, 0.01, [0, 0, 1.0], rv, np.eye(3), [0, 0, 1.0], np.zeros((3, 3)))
    m.fuse(r["map_index"], cr, cr, cr, inten, r["z_ts"], r["var"])
    m.mapvar_update(1e-4); m.map_feature(); m.raytracing()


t = timeit(frame, n=5, warm=2)
print(f"N3 elevation frame (move + 120k points + fuse + features + ray tracing, host arrays in/out, 200x200 map): {1e3*t:.2f} ms/frame")
