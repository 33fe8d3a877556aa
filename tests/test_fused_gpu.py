"""GPU parity of the single-launch descriptor kernel (mrs_ring_descriptors_batch: Cartesian BEV rasterised into the Radon
kernel's LDS tile, sinogram, normalisation) against the two-call path and the oracle: the same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from mr_slam_amd import _lib
    _lib.load()
    return "cuda:0"


def _adversarial_scan(rng, n):
    """points on / next to bin edges, zeros, values past +-1, NaN / inf, negative z: everything the slow path exists for"""
    p = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    p[:, 2] = rng.uniform(-0.2, 1.0, size=n).astype(np.float32)
    k = n // 8
    edges = (rng.integers(0, 121, size=k) / 60.0 - 1.0).astype(np.float32)           # exact bin edges in x
    p[:k, 0] = edges
    p[k:2 * k, 1] = np.nextafter(edges, np.float32(2.0))                               # one ulp past an edge in y
    p[2 * k:2 * k + 8, 0] = [0.0, 1.0, -1.0, 1.5, -3.0, np.nan, np.inf, -np.inf]
    p[2 * k + 8:2 * k + 16, 1] = [0.0, 1.0, -1.0, 1.5, -3.0, np.nan, np.inf, -np.inf]
    p[2 * k + 16:2 * k + 22, 2] = [0.0, 1.0, 1.5, np.nan, np.inf, 0.9999]
    return p


def _both(xyz, offs, want_bev=True):
    from mr_slam_amd import ring
    a = ring.ring_descriptors(xyz, offs, want_bev=want_bev, fused=False)
    b = ring.ring_descriptors(xyz, offs, want_bev=want_bev, fused=True)
    return a, b


def _same(a, b):
    import torch
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            assert x.shape == y.shape
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), "fused kernel differs from the two-call path"


def test_fused_equals_two_call_path_and_oracle(dev, oracle):
    from mr_slam_amd import bev, synth
    rng = np.random.default_rng(11)
    scans = [synth.lidar_scan(3, 20000), _adversarial_scan(rng, 4096), synth.uniform_scan(5, 12345),    # 12345: planes not 16-byte aligned
             synth.lidar_scan(4, 8000), _adversarial_scan(rng, 1001)]                                    # odd batch: last pair half empty
    xyz, offs = bev.pack_scans(scans, dev)
    a, b = _both(xyz, offs)
    _same(a, b)
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    for i, s in enumerate(scans):
        want = oracle.bev_cart(synth.to_soa(s), 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(120, 120)
        assert np.array_equal(b[0][i].cpu().numpy(), want), "fused BEV image differs from the oracle"
        # the adversarial scans carry z = +inf (stored as the cell's maximum, like the reference does): inf * 0 weights -> NaN samples
        assert np.array_equal(b[1][i].cpu().numpy(), oracle.radon_parallel(want[None], ang, 120, 1.0)[0], equal_nan=True), \
            "fused sinogram differs from the oracle"


def test_fused_single_scan_and_outputs_optional(dev):
    import torch
    from mr_slam_amd import bev, ring, synth
    xyz, offs = bev.pack_scans([synth.lidar_scan(7, 30000)], dev)
    a, b = _both(xyz, offs)
    _same(a, b)
    _, _, n_only = ring.ring_descriptors_fused(xyz, offs, want_bev=False, raw=False, normalized=True)
    assert torch.equal(n_only, a[2])
    img_only, s_none, n_none = ring.ring_descriptors_fused(xyz, offs, want_bev=True, raw=False, normalized=False)
    assert s_none is None and n_none is None and torch.equal(img_only, a[0])


def test_fused_blank_scan_counts_as_degenerate(dev):
    import torch
    from mr_slam_amd import bev, ring, synth
    blank = np.zeros((500, 3), np.float32)
    blank[:, 2] = -0.5                                          # nothing above the ground: empty image, constant sinogram
    xyz, offs = bev.pack_scans([synth.lidar_scan(1, 5000), blank, synth.lidar_scan(2, 5000)], dev)
    plan = ring.ring_plan(0)
    plan.degenerate_count(reset=True)
    _, sino, norm = ring.ring_descriptors(xyz, offs, fused=True)
    assert plan.degenerate_count(reset=True) == 1
    assert float(sino[1].abs().max()) == 0.0 and float(norm[1].abs().max()) == 0.0
    assert torch.isfinite(norm).all()


@pytest.mark.parametrize("opts", [dict(), dict(stagger=70), dict(prefetch=4), dict(prefetch=6), dict(grid=7), dict(stagger=25, prefetch=4, grid=64),
                                  dict(variant=0), dict(variant=0, prefetch=4, grid=7), dict(variant=1, prefetch=6, grid=3)])
def test_fused_persistent_rounds_and_tuning_knobs(dev, opts):
    """more pairs than workgroups (rounds handed out by the global counter), with every tuning knob -- incl. the lane <-> ray dealing
    (variant 1: slot tables, rays sorted by length; variant 0: (angle, detector) order): same bits"""
    import torch
    from mr_slam_amd import bev, ring, synth
    base = [synth.lidar_scan(20 + s, 6000) for s in range(3)]
    rng = np.random.default_rng(5)
    scans = []
    for i in range(601):                                        # 301 pairs > 256 compute units, ragged sizes
        p = base[i % 3][: 6000 - 7 * (i % 11)].copy()
        th = rng.uniform(0, 2 * np.pi)
        c, s = np.float32(np.cos(th)), np.float32(np.sin(th))
        q = p.copy()
        q[:, 0] = c * p[:, 0] - s * p[:, 1]
        q[:, 1] = s * p[:, 0] + c * p[:, 1]
        scans.append(q)
    xyz, offs = bev.pack_scans(scans, dev)
    plan = ring.ring_plan(0)
    try:
        plan.set_option(plan.OPT_FUSED_STAGGER_US, opts.get("stagger", 0))
        plan.set_option(plan.OPT_FUSED_PREFETCH, opts.get("prefetch", 2))
        plan.set_option(plan.OPT_FUSED_GRID, opts.get("grid", 0))
        plan.set_option(plan.OPT_FUSED_VARIANT, opts.get("variant", 1))
        a, b = _both(xyz, offs)
        _same(a, b)
        b2 = ring.ring_descriptors(xyz, offs, want_bev=True, fused=True)        # run to run: the same bits (order-free max, fixed sums)
        _same(b, b2)
    finally:
        plan.set_option(plan.OPT_FUSED_STAGGER_US, 70); plan.set_option(plan.OPT_FUSED_PREFETCH, 2); plan.set_option(plan.OPT_FUSED_GRID, 0)
        plan.set_option(plan.OPT_FUSED_VARIANT, 1)


def test_fused_rejects_what_it_cannot_do(dev):
    import ctypes as C
    import torch
    from mr_slam_amd import _lib, bev, ring, synth
    xyz, offs = bev.pack_scans([synth.lidar_scan(1, 2000)], dev)
    plan = ring.ring_plan(0)
    out = torch.empty((1, 120, 120), dtype=torch.float32, device=dev)
    lib = _lib.load()
    for cfg in (_lib.BevCfg(1, 1, 120, 120, 2, 1), _lib.BevCfg(1, 1, 100, 120, 1, 1)):      # two height layers; grid != the plan's image
        st = lib.mrs_ring_descriptors_batch(plan._h, _lib.ptr(xyz), _lib.ptr(offs), 1, C.byref(cfg), None, None, _lib.ptr(out), None)
        assert st != 0 and b"unsupported" in lib.mrs_status_str(st)
    with pytest.raises(_lib.MrsError):
        plan.set_option(plan.OPT_FUSED_PREFETCH, 3)
    with pytest.raises(_lib.MrsError):
        plan.set_option(99, 1)
