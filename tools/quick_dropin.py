"""Latency of the drop-in single-scan entry points, as the reference's Python nodes call them (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mr_slam_amd import synth
from mr_slam_amd.compat import gputransform, voxelocc, voxelfeat, torch_radon

s = synth.lidar_scan(5)
soa = synth.to_soa(s)
n = s.shape[0]


def lat(fn, reps=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e3 * (time.perf_counter() - t0) / reps


def cart():
    t = voxelocc.GPUTransformer(soa, n, 1, 1, 120, 120, 1, 1); t.transform(); return t.retreive()


def polar(H):
    def f():
        t = gputransform.GPUTransformer(soa, n, 1, 1, 40, 120, H, 1); t.transform(); return t.retreive()
    return f


print(f"voxelocc 120x120x1      : {lat(cart):.3f} ms / scan (120k pts, host in -> host out)")
print(f"gputransform 40x120x1   : {lat(polar(1)):.3f} ms")
print(f"gputransform 40x120x20  : {lat(polar(20)):.3f} ms")
F = 9
pts = np.concatenate([soa, np.random.default_rng(6).uniform(0, 1, size=(F - 3) * n).astype(np.float32)])


def feat():
    t = voxelfeat.GPUTransformer(pts, n, 1, 1, 120, 120, 1, F); t.transform(); return t.retreive()


print(f"voxelfeat 120x120x1 F=9 : {lat(feat):.3f} ms")
img = torch.from_numpy(cart().reshape(-1, 3)[:, 2].reshape(1, 120, 120).copy()).cuda()
radon = torch_radon.ParallelBeam(120, np.linspace(0, 2 * np.pi, 120).astype(np.float32))
def rad():
    y = radon.forward(img); torch.cuda.synchronize(); return y
print(f"torch_radon forward 1 img: {lat(rad):.3f} ms")
