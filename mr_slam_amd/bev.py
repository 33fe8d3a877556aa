"""Batched BEV rasterisers on device tensors (rows A1-A5 of SURVEY.md section 8).

Thin host logic over the C ABI (mrs_bev_*); reference semantics are documented there.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BevCfg, OUT_COMPACT, OUT_REFERENCE


def _cfg(max_length, max_height, n0, n1, num_height, last):
    return BevCfg(int(max_length), int(max_height), int(n0), int(n1), int(num_height), int(last))


def _dev(t):
    if not t.is_cuda:
        raise _lib.MrsError("expected a device tensor (no CPU fallback)")
    return t.device.index or 0


def pack_scans(scans, device="cuda:0", planes=3):
    """Pack a list of [n_i, planes] (or already-SoA [planes*n_i]) host scans into the ABI's
    ragged layout: one float32 device tensor + int64 offsets (in points)."""
    flat, offs = [], [0]
    for s in scans:
        a = np.asarray(s, dtype=np.float32)
        if a.ndim == 2:
            a = a[:, :planes].T
        a = np.ascontiguousarray(a).reshape(-1)
        assert a.size % planes == 0
        flat.append(a)
        offs.append(offs[-1] + a.size // planes)
    xyz = torch.from_numpy(np.concatenate(flat) if flat else np.zeros(0, np.float32)).to(device)
    return xyz, torch.tensor(offs, dtype=torch.int64, device=device)


def polar_indices(xyz_soa, max_length, max_height, num_ring, num_sector, num_height):
    """A1: (ring, sector, height) int32 device tensors for one SoA scan [3n]."""
    d = _dev(xyz_soa)
    n = xyz_soa.numel() // 3
    r, s, h = (torch.empty(n, dtype=torch.int32, device=xyz_soa.device) for _ in range(3))
    cfg = _cfg(max_length, max_height, num_ring, num_sector, num_height, 1)
    _lib.check(_lib.load().mrs_bev_polar_indices(_lib.ctx(d), _lib.ptr(xyz_soa), n, C.byref(cfg),
                                                 _lib.ptr(r), _lib.ptr(s), _lib.ptr(h), _lib.current_stream(d)))
    return r, s, h


def cart_indices(xyz_soa, max_length, max_height, num_x, num_y, num_height):
    """A3: (ix, iy, ih) int32 device tensors for one SoA scan [3n]."""
    d = _dev(xyz_soa)
    n = xyz_soa.numel() // 3
    a, b, c = (torch.empty(n, dtype=torch.int32, device=xyz_soa.device) for _ in range(3))
    cfg = _cfg(max_length, max_height, num_x, num_y, num_height, 1)
    _lib.check(_lib.load().mrs_bev_cart_indices(_lib.ctx(d), _lib.ptr(xyz_soa), n, C.byref(cfg),
                                                _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.current_stream(d)))
    return a, b, c


def _batch(fn_name, xyz, offsets, cfg, layout, out_per_scan, out=None):
    d = _dev(xyz)
    batch = offsets.numel() - 1
    if out is None:
        out = torch.empty((batch, out_per_scan), dtype=torch.float32, device=xyz.device)
    fn = getattr(_lib.load(), fn_name)
    _lib.check(fn(_lib.ctx(d), _lib.ptr(xyz), _lib.ptr(offsets), batch, C.byref(cfg), layout,
                  _lib.ptr(out), _lib.current_stream(d)))
    return out


def polar_bev(xyz, offsets, max_length, max_height, num_ring, num_sector, num_height,
              enough_large=1, layout=OUT_COMPACT, out=None):
    """A1+A2.  COMPACT: [B, H, R, S] occupancy;  REFERENCE: [B, 3*R*S*H*enough_large]."""
    cfg = _cfg(max_length, max_height, num_ring, num_sector, num_height, enough_large)
    cells = num_ring * num_sector * num_height
    if layout == OUT_COMPACT:
        o = _batch("mrs_bev_polar_batch", xyz, offsets, cfg, layout, cells, out)
        return o.view(-1, num_height, num_ring, num_sector)
    return _batch("mrs_bev_polar_batch", xyz, offsets, cfg, layout, 3 * cells * enough_large, out)


def cart_bev(xyz, offsets, max_length, max_height, num_x, num_y, num_height,
             layout=OUT_COMPACT, out=None):
    """A3+A4.  COMPACT: [B, H, NX, NY] max-z;  REFERENCE: [B, 3*NX*NY*H]."""
    cfg = _cfg(max_length, max_height, num_x, num_y, num_height, 1)
    cells = num_x * num_y * num_height
    if layout == OUT_COMPACT:
        o = _batch("mrs_bev_cart_batch", xyz, offsets, cfg, layout, cells, out)
        return o.view(-1, num_height, num_x, num_y)
    return _batch("mrs_bev_cart_batch", xyz, offsets, cfg, layout, 3 * cells, out)


def feat_bev(pts, offsets, featsize, max_length, max_height, num_x, num_y, num_height=1,
             layout=OUT_COMPACT, out=None):
    """A5.  COMPACT: [B, F-3, NX, NY] planar (channels 3..F-1; [B, F-3, H, NX, NY] when num_height > 1);
    REFERENCE: [B, NX*NY*H*F]."""
    cfg = _cfg(max_length, max_height, num_x, num_y, num_height, featsize)
    cells = num_x * num_y * num_height
    if layout == OUT_COMPACT:
        o = _batch("mrs_bev_feat_batch", pts, offsets, cfg, layout, cells * (featsize - 3), out)
        return o.view(-1, featsize - 3, num_x, num_y) if num_height == 1 else o.view(-1, featsize - 3, num_height, num_x, num_y)
    return _batch("mrs_bev_feat_batch", pts, offsets, cfg, layout, cells * featsize, out)
