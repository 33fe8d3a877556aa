"""Development aid: LDS bank-conflict model of the Radon sample loop (mr_slam_amd/csrc/radon.hip).

Rebuilds the ray table in numpy (same formulas as ray_setup), walks every wave through its sample
loop and counts LDS-array cycles per the gfx950 rules (ds_read2_b32 = two ds_read_b32; lane groups
{0-31},{32-63}; 32 banks of 4 B; identical addresses broadcast; N distinct addresses on one bank
in a group = N cycles).  Prints cycles per wave-sample for a given ray -> lane assignment.
"""
import sys
import numpy as np

f32 = np.float32
PAD, STRIDE = 2, 125


def ray_table(A=120, D=120, H=120, W=120, spacing=1.0):
    ang = np.linspace(0, 2 * np.pi, A).astype(f32)
    cs = np.cos(ang.astype(np.float64)).astype(f32)[:, None]
    sn = np.sin(ang.astype(np.float64)).astype(f32)[:, None]
    L = f32(np.sqrt(f32(W * 0.5) ** 2 + f32(H * 0.5) ** 2))
    r = np.arange(D, dtype=f32)[None, :]
    sx = (r - f32(D) * f32(0.5) + f32(0.5)) * f32(spacing)
    sy, ex, ey = L, sx, -L
    rsx = sx * cs + sy * sn
    rsy = -sx * sn + sy * cs
    rdx = ex * cs + ey * sn - rsx
    rdy = -ex * sn + ey * cs - rsy
    rsx = rsx + f32(0.5 * W)
    rsy = rsy + f32(0.5 * H)
    dx = np.where(rdx >= 0, np.maximum(rdx, f32(1e-6)), np.minimum(rdx, f32(-1e-6)))
    dy = np.where(rdy >= 0, np.maximum(rdy, f32(1e-6)), np.minimum(rdy, f32(-1e-6)))
    axm, axp = -rsx / dx, (f32(W) - rsx) / dx
    aym, ayp = -rsy / dy, (f32(H) - rsy) / dy
    a_s = np.maximum(np.minimum(axp, axm), np.minimum(ayp, aym))
    a_e = np.minimum(np.maximum(axp, axm), np.maximum(ayp, aym))
    miss = a_s.astype(np.float64) > a_e.astype(np.float64) - 1e-6
    a_e = np.where(miss, a_s + 1, a_e)
    rsx = rsx + rdx * a_s
    rsy = rsy + rdy * a_s
    rdx = rdx * (a_e - a_s)
    rdy = rdy * (a_e - a_s)
    m = np.maximum(np.abs(rdx), np.abs(rdy))
    n = np.rint(m).astype(np.int64)
    vx, vy = rdx / m, rdy / m
    ydom = np.abs(rdy) >= np.abs(rdx)
    inc = np.where(ydom, f32(0.5) - rsy + np.rint(rsy), f32(0.5) - rsx + np.rint(rsx))
    vd = np.where(ydom, vy, vx)
    step = inc / vd + np.where(vd < 0, f32(1), f32(0))
    rsx = rsx + step * vx
    rsy = rsy + step * vy
    major = np.floor(np.where(ydom, rsy, rsx)).astype(np.int64)
    q = np.where(ydom, rsx, rsy) + f32(1.5)
    vm = np.where(ydom, vx, vy)
    neg = vd < 0
    major = np.where(neg, major - (n - 1), major)
    q = np.where(neg, q + (n - 1).astype(f32) * vm, q)
    vm = np.where(neg, -vm, vm)
    n = np.where(miss, 0, n)
    return dict(n=n.ravel(), ydom=ydom.ravel(), major=major.ravel(), q=q.ravel().astype(np.float64),
                vm=vm.ravel().astype(np.float64))


def wave_cycles(t, rays):
    """LDS-array cycles and wave-samples for one wave given the ray ids of its 64 lanes (-1 = idle)."""
    rays = np.asarray(rays)
    act = rays >= 0
    rid = np.where(act, rays, 0)
    n = np.where(act, t["n"][rid], 0)
    ydom = t["ydom"][rid]
    steps = int(n.max()) if n.size else 0
    cyc = 0
    for yd in (True, False):     # the two orientations run one after the other (divergent branch)
        sel = act & (ydom == yd) & (n > 0)
        if not sel.any():
            continue
        ns = int(n[sel].max())
        k = np.arange(ns)[:, None]
        live = sel[None, :] & (k < n[None, :])
        idx = np.floor(t["q"][rid][None, :] + k * t["vm"][rid][None, :]).astype(np.int64)
        line = (t["major"][rid][None, :] + PAD + k)
        a0 = np.where(yd, line * STRIDE + idx, idx * STRIDE + line)     # dword address of tap 0
        tap = 1 if yd else STRIDE
        for g in (slice(0, 32), slice(32, 64)):
            for a in (a0[:, g], a0[:, g] + tap):
                lv = live[:, g]
                for row in range(ns):
                    ad = np.unique(a[row][lv[row]])
                    if ad.size:
                        cyc += np.bincount(ad % 32, minlength=32).max()
    return cyc, int(n[act].sum()), steps


def evaluate(t, order, wg=960):
    """order: ray id per slot; slot s -> lane s % wg of ray-round s // wg (as in k_radon)."""
    R = order.size
    per_lane = (R + wg - 1) // wg
    tot_c = tot_s = tot_ws = 0
    for rnd in range(per_lane):
        for w in range(wg // 64):
            s0 = rnd * wg + w * 64
            rays = np.full(64, -1, dtype=np.int64)
            m = max(0, min(64, R - s0))
            rays[:m] = order[s0:s0 + m]
            c, s, steps = wave_cycles(t, rays)
            tot_c += c
            tot_s += s
            tot_ws += steps
    return tot_c, tot_s, tot_ws


if __name__ == "__main__":
    t = ray_table()
    R = t["n"].size
    ident = np.arange(R)
    c, s, ws = evaluate(t, ident)
    print(f"natural order : LDS cycles {c}  lane-samples {s}  wave-steps(max n per wave) {ws}  "
          f"cycles/wave-step {c / ws:.2f}  ideal 4.00   lane efficiency {s / (64 * ws):.2f}")
