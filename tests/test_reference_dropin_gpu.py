"""The reference's OWN Python (RING_ros/util.py, disco_ros/models/DiSCO.py, phase_corr of disco_ros/main.py), unmodified, executed on the
GPU through the drop-in modules (`mr_slam_amd.compat.install()`): the glue the golden vectors cannot show -- tensor devices, dtypes,
`.cpu()` round trips, constructor / retreive() call sequences.  The files are read from /root/reference where that tree exists and from
the git-ignored scratch copies tools/stage_reference_py.py makes for the GPU box (tests/_refpy/); skipped when neither is there."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_import  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_import.available(), reason="reference Python files not present")]


def test_reference_generate_ring_and_fast_corr_run_on_the_dropin():
    import torch
    from mr_slam_amd import ring, synth
    from oracle import corr_oracle as K
    from oracle import pyoracle as O
    pcs = [synth.lidar_scan(41, 30000), synth.lidar_scan(42, 30000)]
    with ref_import.reference_modules("dropin") as ref:
        u = ref.util
        assert str(u.device).startswith("cuda")                       # util.py:23: the reference picks the GPU
        out = [u.generate_RING(pc) for pc in pcs]                     # util.py:174-200, unmodified, on voxelocc + torch_radon drop-ins
        d_ref, a_ref = u.fast_corr(out[0][2].to(u.device), out[1][2].to(u.device))   # util.py:362-374 on the reference's own tensors
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    for pc, (bev, RING, TIRING) in zip(pcs, out):
        assert isinstance(bev, np.ndarray) and bev.shape == (1, 120, 120) and RING.device.type == "cpu" and TIRING.dtype == torch.complex64
        want_bev = O.bev_cart(synth.to_soa(pc), 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(1, 120, 120)
        assert np.array_equal(bev, want_bev)                                             # bit-exact BEV through the reference's call sequence
        want_sino = O.radon_parallel(want_bev, ang, 120, 1.0)
        assert np.array_equal(RING.numpy(), want_sino)                                   # bit-exact sinogram through torch_radon.ParallelBeam
        np.testing.assert_allclose(torch.view_as_real(TIRING).numpy(), torch.view_as_real(K.tiring_from_sinogram(want_sino)).numpy(), atol=2e-5)
        mine = ring.generate_RING(pc)                                                    # the host mirror gives the same three objects
        assert np.array_equal(mine[0], bev) and torch.equal(mine[1], RING)
    wd, wa, _ = K.fast_corr(K.tiring_from_sinogram(out[0][1].numpy()), K.tiring_from_sinogram(out[1][1].numpy()))
    assert int(a_ref) == int(wa) and abs(float(d_ref) - float(wd)) < 1e-5
    d_mine, a_mine = ring.fast_corr(out[0][2], out[1][2])
    assert int(a_mine) == int(a_ref) and abs(float(d_mine) - float(d_ref)) < 1e-5


def test_reference_generate_ringplusplus_runs_on_the_dropin():
    import torch
    from mr_slam_amd import synth
    pc = synth.lidar_scan(43, 6000)
    with ref_import.reference_modules("dropin") as ref:
        u = ref.util
        bev_t, RING, TIRING = u.generate_RINGplusplus(pc)             # util.py:204-250: sklearn kNN on the host, voxelfeat + torch_radon drop-ins
        d, a = u.fast_corr_RINGplusplus(TIRING.to(u.device), TIRING.to(u.device))
    assert tuple(bev_t.shape) == (6, 120, 120) and bev_t.device.type == "cuda" and tuple(RING.shape) == (6, 120, 120) and RING.device.type == "cpu"
    assert torch.isfinite(RING).all() and torch.isfinite(TIRING).all() and float(RING.abs().max()) > 0
    assert int(a) == 0 and float(d) < 0.5                              # a descriptor against itself: zero shift


def test_reference_disco_forward_and_phase_corr_run_on_the_dropin():
    import torch
    from mr_slam_amd import bev, disco, synth
    pc = synth.lidar_scan(44, 30000)
    with ref_import.reference_modules("dropin") as ref:
        D = ref.disco
        soa = np.ascontiguousarray(pc.T.reshape(-1).astype(np.float32))
        t = D.gputransform.GPUTransformer(soa, pc.shape[0], 1, 1, 40, 120, 20, 1)      # disco_ros/main.py:118-121
        t.transform()
        occ = t.retreive().reshape(-1, 3)[:, 2].reshape(20, 40, 120)
    xyz, offs = bev.pack_scans([pc], "cuda:0")
    mine = bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20).cpu().numpy().reshape(20, 40, 120)
    assert np.array_equal(occ, mine)
    sig, spec = disco.disco_from_bev(torch.from_numpy(occ[None]).cuda())
    assert tuple(sig.shape) == (1, 1024) and torch.isfinite(sig).all() and tuple(spec.shape) == (1, 1, 40, 120)
