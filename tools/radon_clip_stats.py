"""Development aid (CPU only): what clipping every ray of the Radon march to the non-zero part of the BEV could save (the review's "exact-skip"
lever: adding w * 0 is exact).  For pairs of synthetic scans shaped like bench.py's (rotated / shifted copies of synth.lidar_scan, the
reference crop): occupancy, bounding rows / columns of the non-zero texels of the two images of a pair, share of the ray samples inside
that box, share between each ray's first and last non-zero sample (the bound of ANY clipping scheme), and the same in wave-steps (a wave
executes the longest of its 64 rays).  Output: profiles/r03_clip_stats.md."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from mr_slam_amd import synth
from radon_lds_sim import ray_table, PAD
t = ray_table()
n, ydom, major, q, vm = t['n'], t['ydom'], t['major'], t['q'], t['vm']
R = n.size
rng = np.random.default_rng(0)
def bev_of(p):   # normalised points -> occupancy (any z > 0 raises the cell)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    ok = (z > 0) & (z < 1) & (np.abs(x) <= 1) & (np.abs(y) <= 1)
    ix = np.floor((x[ok].astype(np.float64) + 1) * 60).astype(int).clip(0, 119)
    iy = np.floor((y[ok].astype(np.float64) + 1) * 60).astype(int).clip(0, 119)
    img = np.zeros((120, 120), bool); img[ix, iy] = True
    return img
def stats(imgs):
    occ = np.zeros((120,120), bool)
    for im in imgs: occ |= im
    rows = np.where(occ.any(1))[0]; cols = np.where(occ.any(0))[0]
    r0, r1, c0, c1 = rows.min(), rows.max(), cols.min(), cols.max()
    # texture coordinates: which image axis is the "line" for ydom rays?  unknown here -> evaluate both conventions and report the larger saving
    out = {}
    for conv in (0, 1):
        tot = 0; clip = 0; exact = 0
        nclip = np.zeros(R, int)
        for r in range(R):
            k = np.arange(n[r])
            if n[r] == 0: continue
            line = major[r] + k
            idx = np.floor(q[r] + k * vm[r]).astype(int) - PAD + 0   # q is shifted by +1.5 (border + centre): tap cells idx-?..
            # taps: minor cells (idx - 2) and (idx - 1) in image coordinates (q = coord + 1.5, cell = floor(coord - 0.5) -> floor(q) - 2), conservative: use both +-1
            m0 = np.floor(q[r] + k * vm[r]).astype(int) - 2
            if (ydom[r] and conv == 0) or ((not ydom[r]) and conv == 1):
                rr, cc0 = line, m0      # line = row index (first image axis)
                inb = (rr >= r0) & (rr <= r1) & (cc0 + 1 >= c0) & (cc0 <= c1)
                val = np.zeros(n[r], bool)
                okk = (rr >= 0) & (rr < 120)
                for dc in (0, 1):
                    c = cc0 + dc; o2 = okk & (c >= 0) & (c < 120)
                    val[o2] |= occ[rr[o2], c[o2]]
            else:
                cc, rr0 = line, m0
                inb = (cc >= c0) & (cc <= c1) & (rr0 + 1 >= r0) & (rr0 <= r1)
                val = np.zeros(n[r], bool)
                okk = (cc >= 0) & (cc < 120)
                for dr in (0, 1):
                    rr = rr0 + dr; o2 = okk & (rr >= 0) & (rr < 120)
                    val[o2] |= occ[rr[o2], cc[o2]]
            tot += n[r]
            w = np.where(inb)[0]; nclip[r] = (w.max() - w.min() + 1) if w.size else 0
            clip += nclip[r]
            w = np.where(val)[0]; exact += (w.max() - w.min() + 1) if w.size else 0
        out[conv] = (tot, clip, exact, nclip)
    return occ.mean(), (r0, r1, c0, c1), out
base = [synth.lidar_scan(s, metric=True) for s in range(4)]
def scan(i):
    p = base[i % 4]
    th = rng.uniform(0, 2*np.pi); tx, ty = rng.uniform(-3, 3, 2)
    x = np.cos(th)*p[:,0] - np.sin(th)*p[:,1] + tx; y = np.sin(th)*p[:,0] + np.cos(th)*p[:,1] + ty; z = p[:,2]
    ok = (np.abs(x) < 70) & (np.abs(y) < 70) & (z < 30) & (z > 0)
    return np.stack([x[ok]/70, y[ok]/70, z[ok]/30], 1)
for trial in range(4):
    a, b = bev_of(scan(2*trial)), bev_of(scan(2*trial+1))
    occ, bb, out = stats([a, b])
    print(f"pair {trial}: single-image occupancy {a.mean():.3f}/{b.mean():.3f}, union {occ:.3f}, bbox rows {bb[0]}-{bb[1]} cols {bb[2]}-{bb[3]}")
    for conv, (tot, clip, exact, nclip) in out.items():
        # wave-level: slots sorted by (ydom, n) as the library does; wave executes max
        order = np.lexsort((n, ydom))
        cur = sum(n[order[i:i+64]].max() for i in range(0, R, 64))
        order2 = np.lexsort((nclip, ydom))
        new_same = sum(nclip[order[i:i+64]].max() for i in range(0, R, 64))
        new_resort = sum(nclip[order2[i:i+64]].max() for i in range(0, R, 64))
        print(f"   conv {conv}: samples {tot}, inside bbox {clip/tot:.3f}, first..last non-zero (bound of any clipping) {exact/tot:.3f}; wave-steps now {cur}, clipped with the static order {new_same/cur:.3f}, re-sorted per pair {new_resort/cur:.3f}")


def per_ray_exact(occ):
    """length of [first non-zero sample, last non-zero sample] per ray (0: the ray meets no non-zero texel)"""
    ex = np.zeros(R, int)
    for r in range(R):
        if n[r] == 0:
            continue
        k = np.arange(n[r])
        line = major[r] + k
        m0 = np.floor(q[r] + k * vm[r]).astype(int) - 2
        val = np.zeros(n[r], bool)
        for d in (0, 1):
            rr, cc = (line, m0 + d) if ydom[r] else (m0 + d, line)
            o2 = (rr >= 0) & (rr < 120) & (cc >= 0) & (cc < 120)
            val[o2] |= occ[rr[o2], cc[o2]]
        w = np.where(val)[0]
        ex[r] = (w.max() - w.min() + 1) if w.size else 0
    return ex


print("per-ray exact extents in wave-steps (a wave = 64 lane slots, executes its longest ray):")
for trial in range(4):
    occ = bev_of(scan(2 * trial)) | bev_of(scan(2 * trial + 1))
    ex = per_ray_exact(occ)
    order = np.lexsort((n, ydom))
    cur = sum(n[order[i:i + 64]].max() for i in range(0, R, 64))
    static = sum(ex[order[i:i + 64]].max() for i in range(0, R, 64))
    order2 = np.lexsort((ex, ydom))
    resort = sum(ex[order2[i:i + 64]].max() for i in range(0, R, 64))
    print(f"   pair {trial}: samples {ex.sum() / n.sum():.3f}; wave-steps with the static (orientation, length) order {static / cur:.3f}, rays re-dealt per pair by cut length {resort / cur:.3f}")
