"""Generates tests/golden/nclt_scan.npz from the one real scan the reference ships
(LoopDetection/src/disco_ros/test.bin, NCLT velodyne_sync record format) = BASELINE configs[0] input.

  raw_u16      : the records' x, y, z as stored (uint16 [n,3]) + intensity/laser bytes
  hits         : the cloud after the reference's own loader, restated literally record by record from
                 disco_ros/loading_pointclouds.py:38-68 (struct.unpack loop, crop, /70 /70 /20, z flip)
  ring/sector/height/occupied at 40x120x1 (disco_ros/config.py:49-53), 40x120x20 and 120x120x1:
                 produced by the REFERENCE CPU rasteriser oracle/_ref/libref_polar.so on `hits`
                 exactly as load_pc_file_infer feeds it (transpose().flatten().astype(float32))
Run in the build container:  python tests/golden/make_golden_nclt.py
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402

SRC = "/root/reference/LoopDetection/src/disco_ros/test.bin"

raw, hits = [], []
with open(SRC, "rb") as f:
    while True:
        xs = f.read(2)
        if xs == b"":
            break
        x_s = struct.unpack("<H", xs)[0]
        y_s = struct.unpack("<H", f.read(2))[0]
        z_s = struct.unpack("<H", f.read(2))[0]
        i = struct.unpack("B", f.read(1))[0]
        l = struct.unpack("B", f.read(1))[0]
        raw.append((x_s, y_s, z_s, i, l))
        x, y, z = x_s * 0.005 + -100.0, y_s * 0.005 + -100.0, z_s * 0.005 + -100.0
        if np.abs(x) < 70. and z > -20. and z < -2. and np.abs(y) < 70. and not (np.abs(x) < 5. and np.abs(y) < 5.):
            hits += [[x / 70., y / 70., z / 20.]]
raw = np.asarray(raw)
hits = np.asarray(hits)
hits[:, 2] = -hits[:, 2]
soa = hits.transpose().flatten().astype(np.float32)
rec = {"raw_u16": raw[:, :3].astype(np.uint16), "intensity": raw[:, 3].astype(np.uint8), "laser": raw[:, 4].astype(np.uint8),
       "hits": hits}
for (R, S, H) in ((40, 120, 1), (40, 120, 20), (120, 120, 1)):
    ring, sector, height = O.ref_bev_polar_indices(soa, 1, 1, R, S, H)
    out = O.ref_bev_polar(soa, 1, 1, R, S, H, 1)
    fp, cnt = O.occupied_fingerprint(out)
    tag = f"{R}x{S}x{H}"
    rec[f"ring_{tag}"] = ring.astype(np.int16)
    rec[f"sector_{tag}"] = sector.astype(np.int16)
    rec[f"height_{tag}"] = height.astype(np.int16)
    rec[f"occupied_{tag}"] = np.flatnonzero(out.reshape(-1, 3)[:, 2]).astype(np.int32)
    rec[f"fingerprint_{tag}"] = np.array([fp], dtype=np.uint64)
    print(tag, "points", hits.shape[0], "occupied", cnt, "%016x" % fp)
np.savez_compressed(os.path.join(HERE, "nclt_scan.npz"), **rec)
print(os.path.getsize(os.path.join(HERE, "nclt_scan.npz")), "bytes")
