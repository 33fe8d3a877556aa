"""Generates tests/golden/bev_cart_ref.npz from the REFERENCE's own Cartesian / feature rasterisers
(generate_bev_cython_binary, generate_bev_pointfeat_cython) compiled for the host by oracle/Makefile
(oracle/_ref/libref_cart.so, libref_feat.so: the reference's kernel.cu + manager.cu with the single
<<<>>> launch turned into a host loop, see oracle/ref_cuda_host/cuda_runtime.h).

Run in the build container (needs /root/reference):  python tests/golden/make_golden_bev_cart.py
  inputs : generate_bev_cython_binary/test.bin (4096 x 3 float64, == 1.bin), read like test.py does;
           the NCLT scan of tests/golden/nclt_scan.npz (`hits`)
  outputs: per-point indices, the occupied cells of channel 2 with their values, a fingerprint of the whole
           3-channel output at the RING layout 120 x 120 x 1 (RING_ros/config.py:7-11) and at 40 x 120 x 20;
           the feature BEV (F = 9) of the same clouds with deterministic extra channels.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402


def extra_channels(xyz, F):
    """Deterministic stand-ins for the 6 point features (positive and negative values, some zeros)."""
    n = xyz.shape[0]
    j = np.arange(n, dtype=np.float64)
    cols = [np.sin(0.37 * j * (c + 1)) * (1.0 + 0.1 * c) for c in range(F - 3)]
    out = np.stack(cols, 1).astype(np.float32)
    out[::17] = 0.0
    return out


def main():
    pts = np.fromfile("/root/reference/LoopDetection/generate_bev_cython_binary/test.bin", dtype=np.float64).reshape(-1, 3)
    nclt = np.load(os.path.join(HERE, "nclt_scan.npz"))["hits"]
    rec = {}
    for name, cloud in (("testbin", pts), ("nclt", nclt)):
        xyz = cloud.astype(np.float32)
        soa = np.ascontiguousarray(xyz.T).reshape(-1)
        for (NX, NY, H) in ((120, 120, 1), (40, 120, 20)):
            tag = f"{name}_{NX}x{NY}x{H}"
            ix, iy, ih = O.ref_bev_cart_indices(soa, 1, 1, NX, NY, H)
            out = O.ref_bev_cart(soa, 1, 1, NX, NY, H, 1)
            occ = np.flatnonzero(out.reshape(-1, 3)[:, 2]).astype(np.int32)
            rec[f"ix_{tag}"] = ix.astype(np.int16); rec[f"iy_{tag}"] = iy.astype(np.int16); rec[f"ih_{tag}"] = ih.astype(np.int16)
            rec[f"occ_{tag}"] = occ
            rec[f"z_{tag}"] = out.reshape(-1, 3)[occ, 2]
            rec[f"crc_{tag}"] = np.array([zlib.crc32(out.tobytes())], np.uint32)
            print(tag, "occupied", occ.size, "crc %08x" % rec[f"crc_{tag}"][0])
        F = 9
        cm = np.ascontiguousarray(np.concatenate([xyz, extra_channels(xyz, F)], 1).T).reshape(-1)
        fout = O.ref_bev_feat(cm, F, 1, 1, 120, 120, 1)
        nz = np.flatnonzero(fout).astype(np.int32)
        rec[f"feat_nz_{name}"] = nz
        rec[f"feat_val_{name}"] = fout[nz]
        print(name, "feature BEV non-zeros", nz.size)
    np.savez_compressed(os.path.join(HERE, "bev_cart_ref.npz"), **rec)
    print(os.path.getsize(os.path.join(HERE, "bev_cart_ref.npz")), "bytes")


if __name__ == "__main__":
    main()
