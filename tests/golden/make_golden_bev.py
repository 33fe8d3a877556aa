"""Generates tests/golden/bev_polar_*.npz from the REFERENCE's own CPU polar rasteriser.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_bev.py
  inputs : the reference fixtures 1.bin / 2.bin (4096 x 3 float64, already normalised)
           disco_ros/tools/multi-layer-polar-cpu/cython/{1,2}.bin, read as test.py:24-31 does
  outputs: ring/sector/height per point and the occupied-cell list produced by
           oracle/_ref/libref_polar.so (reference kernel.cpp + manager.cpp, unmodified),
           at the DiSCO layout 40 x 120 x 20 and the RING layout 120 x 120 x 1.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402

SRC = "/root/reference/LoopDetection/src/disco_ros/tools/multi-layer-polar-cpu/cython/"

for name in ("1", "2"):
    pts = np.fromfile(SRC + name + ".bin", dtype=np.float64).reshape(-1, 3)
    soa = np.ascontiguousarray(pts.T.astype(np.float32)).reshape(-1)
    rec = {"xyz_soa": soa}
    for (R, S, H) in ((40, 120, 20), (120, 120, 1)):
        ring, sector, height = O.ref_bev_polar_indices(soa, 1, 1, R, S, H)
        out = O.ref_bev_polar(soa, 1, 1, R, S, H, 1)
        fp, cnt = O.occupied_fingerprint(out)
        tag = f"{R}x{S}x{H}"
        rec[f"ring_{tag}"] = ring
        rec[f"sector_{tag}"] = sector
        rec[f"height_{tag}"] = height
        rec[f"occupied_{tag}"] = np.flatnonzero(out.reshape(-1, 3)[:, 2]).astype(np.int32)
        rec[f"fingerprint_{tag}"] = np.array([fp], dtype=np.uint64)
        print(name, tag, cnt, "%016x" % fp)
    np.savez_compressed(os.path.join(HERE, f"bev_polar_{name}.npz"), **rec)
