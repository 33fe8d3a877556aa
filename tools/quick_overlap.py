"""Experiment (development aid): the bench step's correlation + sweep of launch c on a side stream while the rasteriser of
launch c+1 runs on the main one.  Prints serial vs overlapped time per launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mr_slam_amd import bev, ring

dev = "cuda:0"
B, CH = 1024, 8
chunks = bench.make_shard(B, CH, 0, dev)
plan = ring.ring_plan(0)
img = [torch.empty((B, 1, 120, 120), dtype=torch.float32, device=dev) for _ in range(2)]
spec32 = torch.empty((CH + 2, B, 61, 120), dtype=torch.complex64, device=dev)
for c, (xyz, offs) in enumerate(chunks):
    _, _, nrm = ring.ring_descriptors(xyz, offs)
    spec32[c] = ring.half_spectrum(nrm)
spec32[CH:] = spec32[CH - 2:CH]
g = torch.Generator(device=dev).manual_seed(7)
cand = torch.randint(0, B, (CH, B), generator=g, device=dev, dtype=torch.int32)
od = torch.empty((CH, B), device=dev); oa = torch.empty((CH, B), dtype=torch.int32, device=dev)
sv = torch.empty(CH, device=dev); sr = torch.empty(CH, dtype=torch.int64, device=dev)
slot = lambda c: c - 2 if c >= 2 else CH + c
side = torch.cuda.Stream()


def tail(c, norm):
    db = spec32[slot(c)]
    spec, _, _, _ = ring.spectrum_corr_pairs_db(norm, db, cand[c], out=(od[c], oa[c]), spec_out=spec32[c])
    d, a = ring.corr_sweep_fft(spec[:1], db)
    torch.min(d, 1, out=(sv[c:c + 1], sr[c:c + 1]))


def serial():
    for c, (xyz, offs) in enumerate(chunks):
        bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1, out=img[0].view(B, -1))
        _, norm = plan.forward(img[0].view(B, 120, 120), raw=False, normalized=True)
        tail(c, norm)


def overlapped():
    pending = None
    main = torch.cuda.current_stream()
    for c, (xyz, offs) in enumerate(chunks):
        bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1, out=img[c & 1].view(B, -1))       # runs beside the previous launch's tail
        _, norm = plan.forward(img[c & 1].view(B, 120, 120), raw=False, normalized=True)
        ev = torch.cuda.Event(); ev.record(main)
        if pending is not None:
            main.wait_event(pending)            # keep the database slots in order
        with torch.cuda.stream(side):
            side.wait_event(ev)
            norm.record_stream(side)
            tail(c, norm)
            pending = torch.cuda.Event(); pending.record(side)
    main.wait_event(pending)


for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e6 * (time.perf_counter() - t0) / 5 / CH:.1f} us per launch of {B} pairs")
