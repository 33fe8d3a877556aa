"""Development aid: cost model of the two-image Radon march (radon_device.hpp march2) for a given ray -> lane assignment.
Per wave and sample step: 2 ds_read_b64 (cells of 8 bytes; 64 banks of 4 B; lane groups {0-31}, {32-63}; distinct cells with equal
(cell index mod 32) inside a group serialise) and ~24 VALU cycles (tools/ubench/valu.hip: v_fract, v_cvt, v_lshl_add, 2 v_pk_fma at 4,
v_add, v_sub at 2).  A wave executes max(n_steps) of its lanes per orientation.  Prints per workgroup round (one pair of images):
VALU SIMD-cycles / 4 SIMDs, LDS-array cycles, lane efficiency, conflict share."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__file__))
from radon_lds_sim import ray_table, PAD, STRIDE

VALU_PER_STEP = 24.0
U = 6


def wave_cost(t, rays):
    rays = np.asarray(rays)
    act = rays >= 0
    rid = np.where(act, rays, 0)
    n = np.where(act, t["n"][rid], 0)
    ydom = t["ydom"][rid]
    lds = 0
    steps = 0
    ideal = 0
    for yd in (True, False):
        sel = act & (ydom == yd) & (n > 0)
        if not sel.any():
            continue
        ns = int(n[sel].max())
        steps += ns
        k = np.arange(ns)[:, None]
        live = sel[None, :] & (k < n[None, :])
        idx = np.floor(t["q"][rid][None, :] + k * t["vm"][rid][None, :]).astype(np.int64)
        line = (t["major"][rid][None, :] + PAD + k)
        c0 = np.where(yd, line * STRIDE + idx, idx * STRIDE + line)     # cell index of tap 0
        tap = 1 if yd else STRIDE
        for g in (slice(0, 32), slice(32, 64)):
            for c in (c0[:, g], c0[:, g] + tap):
                lv = live[:, g]
                for row in range(ns):
                    cells = np.unique(c[row][lv[row]])
                    if cells.size:
                        lds += np.bincount(cells % 32, minlength=32).max()
                        ideal += 1
    return steps, lds, ideal, int(n[act].sum())


def evaluate(t, waves, name):
    """waves: list (per hardware wave 0..15) of lists of 64-ray arrays"""
    per_wave_steps = []
    lds = ideal = samples = 0
    for wl in waves:
        s = 0
        for rays in wl:
            st, l, i, sm = wave_cost(t, rays)
            s += st; lds += l; ideal += i; samples += sm
        per_wave_steps.append(s)
    per_wave_steps = np.array(per_wave_steps)
    simd = per_wave_steps.reshape(4, -1).sum(1) if len(per_wave_steps) % 4 == 0 else per_wave_steps
    # waves w, w+4, w+8, w+12 share a SIMD (any assignment gives 4 per SIMD)
    simd = np.array([per_wave_steps[i::4].sum() for i in range(4)])
    valu = simd.max() * VALU_PER_STEP
    print(f"{name:34s} wave-steps {per_wave_steps.sum():6d} (max/mean per wave {per_wave_steps.max() / per_wave_steps.mean():.3f})  "
          f"lane eff {samples / (64 * per_wave_steps.sum()):.3f}  VALU cyc {valu:8.0f}  LDS cyc {lds:7d} (conflict-free {ideal})  x{lds / ideal:.2f}")
    return valu, lds


def chunks_to_waves(order, nw=16, balance=False, t=None):
    R = order.size
    chunks = []
    for s0 in range(0, R, 64):
        rays = np.full(64, -1, dtype=np.int64)
        m = min(64, R - s0)
        rays[:m] = order[s0:s0 + m]
        chunks.append(rays)
    waves = [[] for _ in range(nw)]
    if not balance:
        for i, c in enumerate(chunks):
            waves[i % nw].append(c)
    else:       # longest-processing-time first
        cost = [max(t["n"][c[c >= 0]].max(), 0) for c in chunks]
        load = np.zeros(nw)
        for i in np.argsort(cost)[::-1]:
            w = int(np.argmin(load))
            waves[w].append(chunks[i]); load[w] += cost[i]
    return waves


if __name__ == "__main__":
    t = ray_table()
    R = t["n"].size
    n, yd = t["n"], t["ydom"]
    evaluate(t, chunks_to_waves(np.arange(R)), "natural (angle, det)")
    o = np.lexsort((n, yd))
    evaluate(t, chunks_to_waves(o), "sorted (ydom, n)")
    evaluate(t, chunks_to_waves(o, balance=True, t=t), "sorted (ydom, n) + LPT waves")
    b = (n + U - 1) // U
    o = np.lexsort((np.arange(R), b, yd))
    evaluate(t, chunks_to_waves(o, balance=True, t=t), "bucket6 (ydom, ceil(n/6), ray) + LPT")
    b = (n + 11) // 12
    o = np.lexsort((np.arange(R), b, yd))
    evaluate(t, chunks_to_waves(o, balance=True, t=t), "bucket12 + LPT")
    b = (n + 23) // 24
    o = np.lexsort((np.arange(R), b, yd))
    evaluate(t, chunks_to_waves(o, balance=True, t=t), "bucket24 + LPT")
