"""DiSCO descriptor + phase correlation over the C ABI (rows D1, D2).  Mirrors
LoopDetection/src/disco_ros/main.py (load_pc_infer :94-125, generate_DiSCO :84-90, phase_corr :260-272)
and models/DiSCO.py:315-334 (UNet bypassed, as in the ROS node)."""
import numpy as np
import torch

from . import _lib, bev

# disco_ros/main.py:483-485 (the node then overwrites num_height with max_height = 1: main.py:498)
NUM_RING = 40
NUM_SECTOR = 120
NUM_HEIGHT = 20


def _dev(t):
    if not t.is_cuda:
        raise _lib.MrsError("expected a device tensor (no CPU fallback)")
    return t.device.index or 0


def disco_from_bev(bev_occ, col=16):
    """bev_occ float32 [B,H,R,S] (device) -> (signature [B,4*col*col], spectrum complex64 [B,1,R,S])."""
    d = _dev(bev_occ)
    x = bev_occ.contiguous()
    B, H, R, S = x.shape
    sig = torch.empty((B, 4 * col * col), dtype=torch.float32, device=x.device)
    spec = torch.empty((B, 1, R, S, 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mrs_disco_descriptor(_lib.ctx(d), _lib.ptr(x), B, H, R, S, int(col), _lib.ptr(sig),
                                                _lib.ptr(spec), _lib.current_stream(d)))
    return sig, torch.view_as_complex(spec)


def disco_descriptors(xyz, offsets, num_ring=NUM_RING, num_sector=NUM_SECTOR, num_height=NUM_HEIGHT, col=16):
    """Batched load_pc_infer + generate_DiSCO: packed scans -> (signatures, spectra)."""
    occ = bev.polar_bev(xyz, offsets, 1, 1, num_ring, num_sector, num_height)
    return disco_from_bev(occ, col)


def phase_corr(a, b, num_sector=NUM_SECTOR, want_corr=False):
    """disco_ros/main.py:260-272 for P pairs: a, b complex64 [P,1,R,S] (device).  Returns yaw bins [P]
    (int32, = flat argmax % num_sector) and optionally the shifted correlation maps [P,R,S]."""
    d = _dev(a)
    a, b = a.contiguous(), b.contiguous()
    P, _, R, S = a.shape
    arg = torch.empty(P, dtype=torch.int32, device=a.device)
    corr = torch.empty((P, R, S), dtype=torch.float32, device=a.device) if want_corr else None
    _lib.check(_lib.load().mrs_disco_phase_corr(_lib.ctx(d), _lib.ptr(torch.view_as_real(a)),
                                                _lib.ptr(torch.view_as_real(b)), P, R, S, _lib.ptr(arg),
                                                _lib.ptr(corr) if want_corr else None, _lib.current_stream(d)))
    yaw = arg % num_sector
    return (yaw, corr) if want_corr else yaw


def calc_rel_ori(a, b):
    """GlobalManager::calcRelOri (global_manager.cpp:2719-2762), literal: a, b complex64 [P,R,S] or
    [P,1,R,S] (device).  Returns relative angles in degrees, float32 [P]."""
    d = _dev(a)
    a, b = a.contiguous(), b.contiguous()
    R, S = a.shape[-2:]
    P = a.numel() // (R * S)
    out = torch.empty(P, dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().mrs_disco_rel_ori_literal(_lib.ctx(d), _lib.ptr(torch.view_as_real(a)),
                                                     _lib.ptr(torch.view_as_real(b)), P, R, S, _lib.ptr(out),
                                                     _lib.current_stream(d)))
    return out


def signature_search(query, db):
    """Nearest signature (squared L2): query [Q,dim], db [N,dim] float32 device -> (index [Q], dist2 [Q])."""
    d = _dev(query)
    query, db = query.contiguous(), db.contiguous()
    Q, dim = query.shape
    idx = torch.empty(Q, dtype=torch.int32, device=query.device)
    d2 = torch.empty(Q, dtype=torch.float32, device=query.device)
    _lib.check(_lib.load().mrs_signature_search(_lib.ctx(d), _lib.ptr(query), Q, _lib.ptr(db), db.shape[0], dim,
                                                _lib.ptr(idx), _lib.ptr(d2), _lib.current_stream(d)))
    return idx, d2


def signature_knn(query, db, k):
    """The k nearest signatures, ascending (what the Mapping side asks its kd-tree for, global_manager.cpp:1002-1007):
    query [Q,dim], db [N,dim] float32 device -> (index [Q,k] int32, dist2 [Q,k]); rows past N hold -1 / inf."""
    d = _dev(query)
    query, db = query.contiguous(), db.contiguous()
    Q, dim = query.shape
    idx = torch.empty((Q, k), dtype=torch.int32, device=query.device)
    d2 = torch.empty((Q, k), dtype=torch.float32, device=query.device)
    _lib.check(_lib.load().mrs_signature_knn(_lib.ctx(d), _lib.ptr(query), Q, _lib.ptr(db), db.shape[0], dim, int(k),
                                             _lib.ptr(idx), _lib.ptr(d2), _lib.current_stream(d)))
    return idx, d2
