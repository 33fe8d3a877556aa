"""The reference's nodes call the native modules from three rospy callback threads at once
(RING_ros/main_RING.py:433-435).  The C ABI is re-entrant: concurrent callers (host-buffer drop-ins and
device-pointer entry points on their own streams) must get the same results as serial ones."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_callers_match_serial_results():
    import torch
    assert torch.cuda.is_available()
    from mr_slam_amd import bev, gicp, ring, synth
    from mr_slam_amd.compat import gputransform, voxelocc

    scans = [synth.lidar_scan(40 + i, 30000) for i in range(3)]

    def work(i, out):
        """What one robot's callback does: BEV drop-ins, RING descriptor, correlation, a small GICP."""
        s = scans[i]
        soa = synth.to_soa(s)
        with torch.cuda.stream(torch.cuda.Stream()):
            t = voxelocc.GPUTransformer(soa, s.shape[0], 1, 1, 120, 120, 1, 1); t.transform()
            cart = t.retreive()
            g = gputransform.GPUTransformer(soa, s.shape[0], 1, 1, 40, 120, 20, 1); g.transform()
            polar = g.retreive()
            xyz, offs = bev.pack_scans([s], "cuda:0")
            _, sino, norm = ring.ring_descriptors(xyz, offs)
            spec = ring.half_spectrum(norm)
            d, a = ring.corr_pairs_fft(spec[:, 0], spec[:, 0])
            src = (s * [70, 70, 30]).astype(np.float32)
            tgt = (src + [0.3, -0.2, 0.0]).astype(np.float32)
            b = gicp.GicpBatch(1)
            b.set_params(max_correspondence_distance=5.0)
            b.set_sources([src]); b.set_targets([tgt])
            T, conv, _ = b.align()
            torch.cuda.current_stream().synchronize()
            out[i] = (cart, polar, sino.cpu().numpy(), float(d[0]), int(a[0]), T[0], bool(conv[0]))

    serial = {}
    for i in range(3):
        work(i, serial)
    for rep in range(3):
        conc = {}
        th = [threading.Thread(target=work, args=(i, conc)) for i in range(3)]
        for t in th: t.start()
        for t in th: t.join()
        assert sorted(conc) == [0, 1, 2]
        for i in range(3):
            for x, y in zip(serial[i], conc[i]):
                if isinstance(x, np.ndarray):
                    np.testing.assert_array_equal(x, y)
                else:
                    assert x == y
        assert all(serial[i][6] for i in range(3))
