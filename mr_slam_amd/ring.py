"""RING / RING++ host logic over the C ABI: Radon plans, descriptor generation, rotation
correlation and translation solving.  Function names and argument meaning mirror
LoopDetection/src/RING_ros/util.py so the parity tests read like the reference's call sites.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib, bev
from ._lib import OUT_COMPACT

# RING_ros/config.py:7-11
NUM_RING = 120
NUM_SECTOR = 120
NUM_HEIGHT = 1
MAX_LENGTH = 1
MAX_HEIGHT = 1


def _dev(t):
    if not t.is_cuda:
        raise _lib.MrsError("expected a device tensor (no CPU fallback)")
    return t.device.index or 0


class RadonPlan:
    """Parallel-beam geometry bound to one device (C ABI mrs_radon_plan_*)."""

    def __init__(self, det_count, angles, det_spacing, height, width, device=0):
        ang = np.ascontiguousarray(np.asarray(angles, dtype=np.float32))
        self.n_angles, self.det, self.h, self.w = int(ang.size), int(det_count), int(height), int(width)
        self.device = device
        self._h = C.c_void_p()
        _lib.check(_lib.load().mrs_radon_plan_create(_lib.ctx(device), _lib.ptr(ang), self.n_angles, self.det,
                                                     C.c_float(det_spacing), self.h, self.w, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mrs_radon_plan_destroy(self._h)
        except Exception:
            pass

    def degenerate_count(self, reset=True):
        """Sinograms whose fused normalisation met std == 0 (blank / constant image) since the last reset: the reference
        raises there (fn.normalize, util.py:197), the kernel writes zeros and counts."""
        n = C.c_int32(0)
        _lib.check(_lib.load().mrs_radon_plan_degenerate_count(self._h, int(bool(reset)), C.byref(n)))
        return n.value

    OPT_FUSED_STAGGER_US, OPT_FUSED_PREFETCH, OPT_FUSED_GRID, OPT_FUSED_VARIANT, OPT_FUSED_SKIP = 1, 2, 3, 4, 5

    def set_option(self, option, value):
        """Tuning knobs of the fused descriptor kernel (mrs_radon_plan_set_option); results do not depend on them."""
        _lib.check(_lib.load().mrs_radon_plan_set_option(self._h, int(option), int(value)))

    def forward(self, img, raw=True, normalized=False):
        """img float32 [B,H,W] (device, contiguous) -> (sino [B,A,D] | None, sino_norm | None)."""
        d = _dev(img)
        assert img.dtype == torch.float32 and img.is_contiguous() and img.shape[-2:] == (self.h, self.w)
        B = img.numel() // (self.h * self.w)
        sino = torch.empty((B, self.n_angles, self.det), dtype=torch.float32, device=img.device) if raw else None
        norm = torch.empty((B, self.n_angles, self.det), dtype=torch.float32, device=img.device) if normalized else None
        _lib.check(_lib.load().mrs_radon_forward(self._h, _lib.ptr(img), B,
                                                 _lib.ptr(sino) if raw else None,
                                                 _lib.ptr(norm) if normalized else None, _lib.current_stream(d)))
        return sino, norm


_plans = {}
_plans_lock = threading.Lock()


def ring_plan(device=0, num_ring=NUM_RING, num_sector=NUM_SECTOR):
    """The geometry generate_RING builds (util.py:191-192): det_count = num_sector,
    angles = linspace(0, 2*pi, num_ring) (endpoint included), image num_ring x num_sector.
    One plan per (device, geometry), shared by the threads of the process (the reference's detectors run in rospy callback threads)."""
    key = (device, num_ring, num_sector)
    with _plans_lock:
        if key not in _plans:
            angles = np.linspace(0, 2 * np.pi, num_ring).astype(np.float32)
            _plans[key] = RadonPlan(num_sector, angles, 1.0, num_ring, num_sector, device)
        return _plans[key]


def normalize(x, group_len=None):
    """fn.normalize(t, mean=t.mean(), std=t.std()) per group of `group_len` floats
    (default: per leading-dim entry)."""
    d = _dev(x)
    x = x.contiguous()
    gl = int(group_len or x[0].numel())
    out = torch.empty_like(x)
    _lib.check(_lib.load().mrs_normalize_groups(_lib.ctx(d), _lib.ptr(x), _lib.ptr(out), x.numel() // gl, gl,
                                                _lib.current_stream(d)))
    return out


def fft_angle(x):
    """torch.fft.fft2(x, dim=-2, norm='ortho') of real [..., A, D] (util.py:198) -> complex64."""
    d = _dev(x)
    x = x.contiguous()
    A, D = x.shape[-2:]
    out = torch.empty(x.shape + (2,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mrs_fft_angle_r2c(_lib.ctx(d), _lib.ptr(x), x.numel() // (A * D), A, D, _lib.ptr(out),
                                             _lib.current_stream(d)))
    return torch.view_as_complex(out)


def forward_row_fft(x):
    """util.py:295-300 (magnitude only): |fft along the detector axis|, ortho."""
    d = _dev(x)
    x = x.contiguous()
    A, D = x.shape[-2:]
    out = torch.empty_like(x)
    _lib.check(_lib.load().mrs_fft_row_magnitude(_lib.ctx(d), _lib.ptr(x), x.numel() // (A * D), A, D,
                                                 _lib.ptr(out), _lib.current_stream(d)))
    return out


# Batches of at least this many scans take the single-launch kernel (mrs_ring_descriptors_batch) in ring_descriptors();
# None = always the two-call sequence.  Both give the same bits; below ~2 rounds of workgroups per compute unit the persistent
# kernel has nothing to overlap and the two stand-alone kernels (2 + 1 workgroups per compute unit) are as fast or faster.
FUSED_MIN_BATCH = 2048      # measured: 0.51 vs 0.54 ms per 1024 scans at 1024, 0.41 at 8192, 0.39 at 24 576 (one MI355X)


def ring_descriptors(xyz, offsets, num_ring=NUM_RING, num_sector=NUM_SECTOR, want_bev=False, fused=None):
    """Batched generate_RING front half (util.py:174-197): Cartesian BEV -> Radon -> normalise.
    Returns (bev | None, sinogram [B,A,D], normalised sinogram [B,A,D]).  fused: True / False picks the single-launch
    kernel / the two-call sequence (same bits), None decides by FUSED_MIN_BATCH."""
    d = _dev(xyz)
    auto = fused is None
    if auto:
        fused = FUSED_MIN_BATCH is not None and offsets.numel() - 1 >= FUSED_MIN_BATCH
    if fused:
        try:
            return ring_descriptors_fused(xyz, offsets, num_ring, num_sector, want_bev=want_bev, raw=True, normalized=True)
        except _lib.MrsError as e:
            # geometries whose two interleaved images do not fit the LDS (or with more than 16 rays per lane) only have the
            # two-call path: an automatic choice falls back to it, an explicit fused=True reports the refusal
            if not auto or "unsupported configuration" not in str(e):
                raise
    img = bev.cart_bev(xyz, offsets, MAX_LENGTH, MAX_HEIGHT, num_ring, num_sector, 1, layout=OUT_COMPACT)
    img = img.view(-1, num_ring, num_sector)
    sino, norm = ring_plan(d, num_ring, num_sector).forward(img, raw=True, normalized=True)
    return (img if want_bev else None), sino, norm


def ring_descriptors_fused(xyz, offsets, num_ring=NUM_RING, num_sector=NUM_SECTOR, want_bev=False, raw=True, normalized=True, out_norm=None):
    """The same in ONE launch (C ABI mrs_ring_descriptors_batch): a persistent workgroup per compute unit rasterises two scans
    straight into the Radon kernel's LDS tile and marches the rays; the BEV image reaches HBM only if want_bev.
    Bit-identical to ring_descriptors(fused=False)."""
    d = _dev(xyz)
    assert xyz.dtype == torch.float32 and xyz.is_contiguous() and offsets.dtype == torch.int64
    B = offsets.numel() - 1
    plan = ring_plan(d, num_ring, num_sector)
    dev = xyz.device
    img = torch.empty((B, num_ring, num_sector), dtype=torch.float32, device=dev) if want_bev else None
    sino = torch.empty((B, plan.n_angles, plan.det), dtype=torch.float32, device=dev) if raw else None
    norm = None
    if normalized:
        norm = out_norm if out_norm is not None else torch.empty((B, plan.n_angles, plan.det), dtype=torch.float32, device=dev)
        assert norm.is_contiguous() and norm.numel() == B * plan.n_angles * plan.det
    cfg = _lib.BevCfg(MAX_LENGTH, MAX_HEIGHT, num_ring, num_sector, 1, 1)
    _lib.check(_lib.load().mrs_ring_descriptors_batch(plan._h, _lib.ptr(xyz), _lib.ptr(offsets), B, C.byref(cfg),
                                                      _lib.ptr(img) if want_bev else None, _lib.ptr(sino) if raw else None,
                                                      _lib.ptr(norm) if normalized else None, _lib.current_stream(d)))
    return img, sino, norm


class DeviceMirror:
    """Device twins of HOST tensors, by identity -- what lets the UNCHANGED candidate loop of the nodes
        for idx in range(len(pc_candidates)): dist, angle = fast_corr(TIRING_current, TIRING_candidates[idx])      (main_RING.py:133-134)
    run through the drop-in without re-uploading two 115 KB host tensors per candidate (round 5: 4.4 k pairs/s, all of it PCIe round trips).
    The reference keeps its descriptors as CPU tensors in Python lists (generate_RING returns `.cpu()` copies, util.py:200); the mirror maps
    id(tensor) -> (tensor._version, weak reference, device value).  A hit needs the same object at the same version: an in-place torch
    operation on a cached tensor bumps `_version` and the twin is rebuilt at the next use; a collected tensor frees its slot through the weak
    reference's callback (and an id re-used by a new tensor can therefore never alias an old entry).  Writes that bypass torch's version
    counter (through a `.numpy()` view) are not seen -- the nodes never write to a stored descriptor.  `max_bytes` bounds the device memory
    held (oldest entries go first).  Thread-safe (the nodes' three callbacks share the lists)."""

    def __init__(self, max_bytes=8 << 30):
        self._d = {}
        self._lock = threading.RLock()           # re-entrant: a weak-reference callback (garbage collection) may fire on the thread that holds it
        self.max_bytes = int(max_bytes)
        self.bytes = 0
        self.hits = self.misses = 0

    def __len__(self):
        return len(self._d)

    def _evict(self, key, ref=None):
        with self._lock:
            ent = self._d.get(key)
            if ent is not None and (ref is None or ent[1] is ref):
                del self._d[key]
                self.bytes -= ent[3]

    def put(self, t, value):
        """register `value` (device tensor or tuple of them) as the twin of host tensor `t` at its current version"""
        import weakref
        key = id(t)
        nbytes = sum(v.numel() * v.element_size() for v in (value if isinstance(value, tuple) else (value,)) if isinstance(v, torch.Tensor))
        ref = weakref.ref(t, lambda r, k=key: self._evict(k, r))
        with self._lock:
            old = self._d.pop(key, None)
            if old is not None:
                self.bytes -= old[3]
            while self._d and self.bytes + nbytes > self.max_bytes:
                k0 = next(iter(self._d))               # insertion order: the oldest entry
                self.bytes -= self._d.pop(k0)[3]
            self._d[key] = (t._version, ref, value, nbytes)
            self.bytes += nbytes
        return value

    def get(self, t, build):
        """the device twin of host tensor `t`; `build(t)` makes it on a miss"""
        ent = self._d.get(id(t))
        if ent is not None and ent[0] == t._version and ent[1]() is t:
            self.hits += 1
            return ent[2]
        self.misses += 1
        return self.put(t, build(t))

    def clear(self):
        with self._lock:
            self._d.clear()
            self.bytes = 0
            self.hits = self.misses = 0


_mirror = DeviceMirror()


def device_mirror():
    """the process-wide mirror behind fast_corr / fast_corr_RINGplusplus (tests, memory accounting)"""
    return _mirror


def _cacheable(x):
    return isinstance(x, torch.Tensor) and not x.is_cuda and not x.requires_grad


def generate_RING(pc, device="cuda:0"):
    """util.py:174-200 for one pre-processed cloud [n,3]: returns (pc_bev [1,R,S] numpy,
    pc_RING [1,A,D] cpu tensor, pc_TIRING complex64 [1,A,D] cpu tensor)."""
    xyz, offs = bev.pack_scans([np.asarray(pc)[:, 0:3]], device)
    img, sino, norm = ring_descriptors(xyz, offs, want_bev=True)
    if not bool(norm.any()):
        # torchvision's fn.normalize at util.py:197 raises for a constant sinogram (blank scan).  The kernels write an all-zero
        # descriptor for such an image (a valid normalised sinogram has unit variance, so it can never be all zeros): the test is on
        # THIS call's output, not on the plan-wide counter, which other callers and threads of the cached plan share
        raise ValueError("std evaluated to zero after conversion to torch.float32, leading to division by zero.")
    tiring = fft_angle(norm)
    host = tiring.cpu()
    # the device copy this call already holds becomes the host tensor's twin (Hermitian by construction: the FFT of a real sinogram): fast_corr
    # never uploads it
    _mirror.put(host, ("half", tiring[:, :61].contiguous()[None]) if tuple(tiring.shape) == (1, 120, 120) else ("full", tiring))
    return img.cpu().numpy(), sino.cpu(), host


def corr_sweep(query, db, want_corr=False):
    """C1/C2 sweep: query [Q,C,A,D], db [N,C,A,D] normalised real descriptors (device).
    Returns (dist [Q,N] float32, angle [Q,N] int32[, corr [Q,N,A]])."""
    d = _dev(query)
    query, db = query.contiguous(), db.contiguous()
    Q, Cc, A, D = query.shape
    N = db.shape[0]
    assert db.shape[1:] == query.shape[1:]
    dist = torch.empty((Q, N), dtype=torch.float32, device=query.device)
    ang = torch.empty((Q, N), dtype=torch.int32, device=query.device)
    corr = torch.empty((Q, N, A), dtype=torch.float32, device=query.device) if want_corr else None
    _lib.check(_lib.load().mrs_ring_corr_sweep(_lib.ctx(d), _lib.ptr(query), Q, _lib.ptr(db), N, Cc, A, D,
                                               _lib.ptr(dist), _lib.ptr(ang),
                                               _lib.ptr(corr) if want_corr else None, _lib.current_stream(d)))
    return (dist, ang, corr) if want_corr else (dist, ang)


def corr_pairs(a, b, out=None):
    """Pairwise C1/C2: a, b [P,C,A,D] normalised real descriptors -> (dist [P], angle [P])."""
    d = _dev(a)
    a, b = a.contiguous(), b.contiguous()
    P, Cc, A, D = a.shape
    assert b.shape == a.shape
    dist, ang = out if out is not None else (torch.empty(P, dtype=torch.float32, device=a.device),
                                             torch.empty(P, dtype=torch.int32, device=a.device))
    _lib.check(_lib.load().mrs_ring_corr_pairs(_lib.ctx(d), _lib.ptr(a), _lib.ptr(b), P, Cc, A, D, _lib.ptr(dist),
                                               _lib.ptr(ang), None, _lib.current_stream(d)))
    return dist, ang


def _tiring_twin(x, device):
    """TIRING spectrum (complex64 [C,A,D], host or device) -> its device form for fast_corr: ("half", [1,1,61,120]) when it is ONE channel of the
    reference's 120 x 120 geometry and the spectrum is Hermitian along the angle axis -- every TIRING is: it is the FFT of a real sinogram
    (util.py:198), so rows 61..119 repeat rows 59..1 conjugated and the pair kernel on half spectra (11 us; the database format) gives
    fast_corr's numbers -- else ("full", [C,A,D]) for the general kernel (143 us).  The symmetry is CHECKED, once per tensor."""
    t = torch.as_tensor(x).to(device).to(torch.complex64).contiguous()
    if t.dim() == 3 and tuple(t.shape) == (1, 120, 120):       # one channel: fast_corr has no channel factor (util.py:369), the RING++ pair kernel has
        tol = 1e-5 * float(t.abs().max())
        if float((t[:, 61:] - t[:, 1:60].flip(1).conj()).abs().max()) <= tol and float(t[:, (0, 60)].imag.abs().max()) <= tol:
            return ("half", t[:, :61].contiguous()[None])
    return ("full", t)


def _full_of(tw):
    if tw[0] == "full":
        return tw[1]
    h = tw[1][0]
    return torch.cat([h, h[:, 1:60].flip(1).conj()], 1).contiguous()


def fast_corr(a, b, device="cuda:0", want_corr=False):
    """util.py:362-374 on TIRING spectra a, b (complex64 [C,A,D], host or device).
    Returns (dist, angle) as numpy scalars like the reference.  Host TENSORS keep a device twin (DeviceMirror): the node's loop over its stored
    descriptors costs one small launch and one 8-byte read-back per candidate, no upload after a tensor's first visit."""
    ta = _mirror.get(a, lambda t: _tiring_twin(t, device)) if _cacheable(a) else _tiring_twin(a, device)
    tb = _mirror.get(b, lambda t: _tiring_twin(t, device)) if _cacheable(b) else _tiring_twin(b, device)
    dv = ta[1].device
    out = torch.empty(2, dtype=torch.float32, device=dv)               # (dist, angle bits): one read-back instead of two
    if ta[0] == "half" and tb[0] == "half" and ta[1].shape == tb[1].shape and not want_corr:
        corr_pairs_fft(ta[1][0], tb[1][0], out=(out[0:1], out[1:2].view(torch.int32)))
        h = out.cpu().numpy()
        return h[0], h[1:2].view(np.int32)[0]
    fa, fb = _full_of(ta), _full_of(tb)
    Cc, A, D = fa.shape
    d = _dev(fa)
    corr = torch.empty(A, dtype=torch.float32, device=dv) if want_corr else None
    _lib.check(_lib.load().mrs_ring_corr_spectra(_lib.ctx(d), _lib.ptr(torch.view_as_real(fa)),
                                                 _lib.ptr(torch.view_as_real(fb)), 1, Cc, A, D, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(out.data_ptr() + 4), _lib.ptr(corr) if want_corr else None,
                                                 _lib.current_stream(d)))
    h = out.cpu().numpy()
    res = (h[0], h[1:2].view(np.int32)[0])
    return res + (corr.cpu().numpy(),) if want_corr else res


def _ringpp_spec(t, device):
    """RING++ TIRING magnitudes float32 [C,120,120] -> half spectrum of the jointly normalised channels (what fast_corr_RINGplusplus recomputes
    for both arguments at every comparison, util.py:339-343)"""
    return half_spectrum(normalize(t.to(device=device, dtype=torch.float32).contiguous()[None]))


def fast_corr_RINGplusplus(a, b, device="cuda:0"):
    """util.py:337-358 on RING++ TIRING magnitudes a, b (float32 [C,A,D]).  Host tensors keep their normalised half spectrum on the device
    (DeviceMirror): per candidate of the node's loop (main_RINGplusplus.py:131-134) one pair launch, nothing uploaded or re-normalised."""
    if tuple(torch.as_tensor(a).shape[-2:]) == (120, 120):      # FFT-domain kernel (specialised for the reference geometry)
        sa = _mirror.get(a, lambda t: _ringpp_spec(t, device)) if _cacheable(a) else _ringpp_spec(torch.as_tensor(a), device)
        sb = _mirror.get(b, lambda t: _ringpp_spec(t, device)) if _cacheable(b) else _ringpp_spec(torch.as_tensor(b), device)
        out = torch.empty(2, dtype=torch.float32, device=sa.device)
        corr_pairs_fft(sa, sb, out=(out[0:1], out[1:2].view(torch.int32)))
        h = out.cpu().numpy()
        return h[0], h[1:2].view(np.int32)[0]
    a = torch.as_tensor(a, dtype=torch.float32).to(device).contiguous()
    b = torch.as_tensor(b, dtype=torch.float32).to(device).contiguous()
    dist, ang = corr_sweep(normalize(a[None]), normalize(b[None]))
    return dist.cpu().numpy()[0, 0], ang.cpu().numpy()[0, 0]


def solve_translation(query, positive, rot_angle, device="cuda:0", want_shifts=False, least_squares=False):
    """util.py:388-423: query, positive float32 [C,H,W]; returns (x, y, error) -- by default the numbers the reference returns.

    The 120 row correlations and their integer shifts always run on the GPU.  The final 120 x 2 solve:
    default = the reference's call, `solve_overdetermined_linear_system(A, b, method='svd')` (util.py:415, 488-506), i.e. the product
    `v.t() @ s_inv @ u.t() @ b` as written.  torch.svd already returns V, so this is the least-squares solution turned by an
    orthogonal matrix that depends on the SVD routine's sign / ordering choices (A is almost isotropic, both singular values
    ~ sqrt(H/2)): it is formed on the host with the same torch.svd call the reference makes on CPU tensors, on the GPU-computed
    shifts, and equals the reference-run numbers of tests/golden/ref_corr.npz.  main_RING.py:177-178 consumes exactly these x, y.
    least_squares=True returns the pseudo-inverse solution instead (the reference's own `method='pinv'` branch), evaluated in
    the kernel.

    Scope of the parity claim: the default equals the reference run on CPU tensors (torch.svd -> LAPACK gesdd).  The reference runs
    torch.svd on its `device`; on a CUDA device another SVD backend may pick other signs / another order for the two nearly equal
    singular vectors, and `v.t() @ ...` (instead of `v @ ...`) then yields a differently turned (x, y).  The default is therefore not
    a property of the data; callers that need the well-defined answer use least_squares=True, callers that compare with the
    reference's logged numbers keep the default."""
    q = torch.as_tensor(query, dtype=torch.float32).to(device).contiguous()
    p = torch.as_tensor(positive, dtype=torch.float32).to(device).contiguous()
    Cc, H, W = q.shape
    d = _dev(q)
    angles = torch.from_numpy(np.linspace(0, 2 * np.pi, H).astype(np.float32)).to(q.device)
    rot = torch.tensor([float(rot_angle)], dtype=torch.float32, device=q.device)
    res = torch.empty(3, dtype=torch.float32, device=q.device)
    sh = torch.empty(H, dtype=torch.float32, device=q.device) if (want_shifts or not least_squares) else None
    _lib.check(_lib.load().mrs_ring_solve_translation(_lib.ctx(d), _lib.ptr(q), _lib.ptr(p), 1, Cc, H, W,
                                                      _lib.ptr(angles), _lib.ptr(rot), _lib.ptr(res),
                                                      _lib.ptr(sh) if sh is not None else None, _lib.current_stream(d)))
    if not least_squares:
        b = sh.cpu()
        ang = torch.FloatTensor(np.linspace(0, 2 * np.pi, H).astype(np.float32)) + rot_angle
        A = torch.stack([torch.cos(ang), torch.sin(ang)], dim=1)
        u, s, v = torch.svd(A, some=False)
        s_new = torch.zeros(A.shape)
        for i in range(len(s)):
            s_new[i, i] = 1 / s[i]
        sol = v.t() @ s_new.t() @ u.t() @ b.view(H, -1)
        err = torch.norm(torch.matmul(A, sol) - b.view(H, -1))
        out = (sol[0].numpy(), sol[1].numpy(), err.numpy())
    else:
        r = res.cpu().numpy()
        out = (r[0:1].copy(), r[1:2].copy(), r[2])
    return out + (sh.cpu().numpy(),) if want_shifts else out


def rotate_bev(bev_img, angle):
    """util.py:67-70: rotate a [C,H,W] (or [N,C,H,W] with one angle per N) BEV by `angle` radians
    (torchvision rotate defaults: nearest, about the centre, zero fill)."""
    d = _dev(bev_img)
    x = bev_img.contiguous()
    single = x.dim() == 3
    if single:
        x = x[None]
    N, Cc, H, W = x.shape
    ang = torch.as_tensor(np.atleast_1d(np.asarray(angle, dtype=np.float64)) * 180.0 / np.pi, dtype=torch.float32).to(x.device)
    assert ang.numel() == N
    out = torch.empty_like(x)
    _lib.check(_lib.load().mrs_rotate_nearest(_lib.ctx(d), _lib.ptr(x), N * Cc, Cc, H, W, _lib.ptr(ang), _lib.ptr(out),
                                              _lib.current_stream(d)))
    return out[0] if single else out


def solve_translation_bev(a, b, want_corr=False, num_ring=NUM_RING, num_sector=NUM_SECTOR):
    """util.py:427-450 for P pairs: a, b float32 [P,C,H,W] (device) or [C,H,W].
    Returns (y, x, -max) per pair exactly like the reference's return order."""
    single = a.dim() == 3
    if single:
        a, b = a[None], b[None]
    d = _dev(a)
    a, b = a.contiguous(), b.contiguous()
    P, Cc, H, W = a.shape
    arg = torch.empty(P, dtype=torch.int32, device=a.device)
    mx = torch.empty(P, dtype=torch.float32, device=a.device)
    corr = torch.empty((P, H, W), dtype=torch.float32, device=a.device) if want_corr else None
    _lib.check(_lib.load().mrs_bev_translation(_lib.ctx(d), _lib.ptr(a), _lib.ptr(b), P, Cc, H, W, _lib.ptr(arg),
                                               _lib.ptr(mx), _lib.ptr(corr) if want_corr else None,
                                               _lib.current_stream(d)))
    arg = arg.cpu().numpy()
    idx_x, idx_y = arg // W, arg % W
    x = idx_x - num_sector // 2
    y = num_ring // 2 - idx_y
    neg = -mx.cpu().numpy()
    out = (y[0], x[0], neg[0]) if single else (y, x, neg)
    return out + (corr,) if want_corr else out


def ringpp_descriptors(points, offsets, k=30, num_ring=NUM_RING, num_sector=NUM_SECTOR):
    """Batched generate_RINGplusplus (util.py:204-250), everything on the device:
    kNN + eigen features (row N1) -> 9-channel feature BEV -> Radon of the 6 feature channels ->
    |FFT along the detector axis|.  points: float32 device tensor [N, s>=3] of pre-processed clouds,
    offsets: host int64 [B+1].  Returns (bev [B,6,R,S], ring [B,6,A,D], tiring [B,6,A,D])."""
    from . import pointfeat
    d = _dev(points)
    planes = pointfeat.point_features(points, offsets, k, want=("planes",))["planes"]
    offs_dev = torch.as_tensor(np.asarray(offsets, dtype=np.int64)).to(points.device)
    fb = bev.feat_bev(planes, offs_dev, 9, MAX_LENGTH, MAX_HEIGHT, num_ring, num_sector, 1, layout=OUT_COMPACT)
    B = fb.shape[0]
    sino, _ = ring_plan(d, num_ring, num_sector).forward(fb.reshape(B * 6, num_ring, num_sector))
    sino = sino.view(B, 6, num_ring, num_sector)
    return fb, sino, forward_row_fft(sino)


def generate_RINGplusplus(pc, device="cuda:0"):
    """util.py:204-250 for one pre-processed cloud [n,3]: (pc_bev_tensor [6,R,S] device,
    pc_RING [6,A,D] cpu, pc_TIRING [6,A,D] cpu)."""
    pts = torch.from_numpy(np.ascontiguousarray(np.asarray(pc, dtype=np.float32)[:, 0:3])).to(device)
    fb, sino, tiring = ringpp_descriptors(pts, np.array([0, pts.shape[0]], np.int64))
    host = tiring[0].cpu()
    if tuple(host.shape[-2:]) == (120, 120):
        _mirror.put(host, _ringpp_spec(tiring[0], device))     # fast_corr_RINGplusplus' device form of this descriptor, made once, here
    return fb[0], sino[0].cpu(), host


def half_spectrum(norm):
    """Half TIRING (first 61 angle-frequency rows, ortho) of normalised sinograms [..., 120, 120]:
    the database format of the FFT-domain sweep."""
    d = _dev(norm)
    x = norm.contiguous()
    A, D = x.shape[-2:]
    n = x.numel() // (A * D)
    out = torch.empty(x.shape[:-2] + (A // 2 + 1, D, 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mrs_ring_half_spectrum(_lib.ctx(d), _lib.ptr(x), n, A, D, _lib.ptr(out), _lib.current_stream(d)))
    return torch.view_as_complex(out)


def half_spectrum_f16(norm, want_f32=True):
    """half_spectrum plus its fp16 replica [..., 61, 120, 2] float16 (the multi-GPU exchange format).
    Returns (spec complex64 | None, spec16 float16)."""
    d = _dev(norm)
    x = norm.contiguous()
    A, D = x.shape[-2:]
    n = x.numel() // (A * D)
    shape = x.shape[:-2] + (A // 2 + 1, D, 2)
    out = torch.empty(shape, dtype=torch.float32, device=x.device) if want_f32 else None
    out16 = torch.empty(shape, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().mrs_ring_half_spectrum_f16(_lib.ctx(d), _lib.ptr(x), n, A, D, _lib.ptr(out) if want_f32 else None,
                                                      _lib.ptr(out16), _lib.current_stream(d)))
    return (torch.view_as_complex(out) if want_f32 else None), out16


def spectrum_corr_pairs(norm, cand_spec, want_f32=True, want_f16=False, out=None):
    """half_spectrum(norm) and corr_pairs_fft(that, cand_spec) in one launch.  norm float32 [P,120,120], cand_spec
    complex64 [P,61,120].  Returns (spec | None, spec16 | None, dist [P], angle [P])."""
    d = _dev(norm)
    x, c = norm.contiguous(), cand_spec.contiguous()
    P = x.shape[0]
    assert x.shape[-2:] == (120, 120) and c.shape == (P, 61, 120)
    spec = torch.empty((P, 61, 120, 2), dtype=torch.float32, device=x.device) if want_f32 else None
    spec16 = torch.empty((P, 61, 120, 2), dtype=torch.float16, device=x.device) if want_f16 else None
    dist, ang = out if out is not None else (torch.empty(P, dtype=torch.float32, device=x.device),
                                             torch.empty(P, dtype=torch.int32, device=x.device))
    _lib.check(_lib.load().mrs_ring_spectrum_corr_pairs(_lib.ctx(d), _lib.ptr(x), _lib.ptr(torch.view_as_real(c)), P, 120, 120,
                                                        _lib.ptr(spec) if want_f32 else None, _lib.ptr(spec16) if want_f16 else None,
                                                        _lib.ptr(dist), _lib.ptr(ang), _lib.current_stream(d)))
    return (torch.view_as_complex(spec) if want_f32 else None), spec16, dist, ang


def spectrum_corr_pairs_db(norm, db_spec, cand_index, want_f32=True, want_f16=False, out=None, spec_out=None):
    """spectrum_corr_pairs with the candidate of pair i = row cand_index[i] (int32 device tensor) of a database of half
    spectra: complex64 [N,61,120] or the fp16 replica format [N,61,120,2] float16.  spec_out (optional): complex64
    [P,61,120] tensor that receives the new half spectra (e.g. the database slot they belong to)."""
    d = _dev(norm)
    x, db, idx = norm.contiguous(), db_spec.contiguous(), cand_index.contiguous()
    P = x.shape[0]
    assert x.shape[-2:] == (120, 120) and idx.dtype == torch.int32 and idx.numel() == P
    f16 = db.dtype == torch.float16
    assert (f16 and db.shape[1:] == (61, 120, 2)) or (db.dtype == torch.complex64 and db.shape[1:] == (61, 120))
    if spec_out is not None:
        assert spec_out.dtype == torch.complex64 and spec_out.shape == (P, 61, 120) and spec_out.is_contiguous()
        spec, want_f32 = torch.view_as_real(spec_out), True
    else:
        spec = torch.empty((P, 61, 120, 2), dtype=torch.float32, device=x.device) if want_f32 else None
    spec16 = torch.empty((P, 61, 120, 2), dtype=torch.float16, device=x.device) if want_f16 else None
    dist, ang = out if out is not None else (torch.empty(P, dtype=torch.float32, device=x.device),
                                             torch.empty(P, dtype=torch.int32, device=x.device))
    _lib.check(_lib.load().mrs_ring_spectrum_corr_pairs_db(_lib.ctx(d), _lib.ptr(x), _lib.ptr(db if f16 else torch.view_as_real(db)),
                                                           int(f16), int(db.shape[0]), _lib.ptr(idx), P, 120, 120,
                                                           _lib.ptr(spec) if want_f32 else None, _lib.ptr(spec16) if want_f16 else None,
                                                           _lib.ptr(dist), _lib.ptr(ang), _lib.current_stream(d)))
    return (torch.view_as_complex(spec) if want_f32 else None), spec16, dist, ang


def corr_sweep_fft(query_spec, db_spec, want_corr=False):
    """C1 sweep on half spectra: query_spec [Q,61,120] complex64, db_spec [N,61,120] complex64 or its fp16
    replica [N,61,120,2] float16 (device)."""
    d = _dev(query_spec)
    q, db = query_spec.contiguous(), db_spec.contiguous()
    Q, N = q.shape[0], db.shape[0]
    dist = torch.empty((Q, N), dtype=torch.float32, device=q.device)
    ang = torch.empty((Q, N), dtype=torch.int32, device=q.device)
    corr = torch.empty((Q, N, 120), dtype=torch.float32, device=q.device) if want_corr else None
    if db.dtype == torch.float16:
        assert db.shape[1:] == (61, 120, 2)
        fn, dbp = _lib.load().mrs_ring_corr_fft_sweep_f16, _lib.ptr(db)
    elif db.dim() == 4:        # RING++: [N, C, 61, 120]
        assert db.dtype == torch.complex64 and q.dim() == 4 and q.shape[1] == db.shape[1]
        _lib.check(_lib.load().mrs_ring_corr_fft_sweep_mc(_lib.ctx(d), _lib.ptr(torch.view_as_real(q)), Q,
                                                          _lib.ptr(torch.view_as_real(db)), N, int(db.shape[1]), _lib.ptr(dist),
                                                          _lib.ptr(ang), _lib.ptr(corr) if want_corr else None,
                                                          _lib.current_stream(d)))
        return (dist, ang, corr) if want_corr else (dist, ang)
    else:
        assert db.dtype == torch.complex64
        fn, dbp = _lib.load().mrs_ring_corr_fft_sweep, _lib.ptr(torch.view_as_real(db))
    _lib.check(fn(_lib.ctx(d), _lib.ptr(torch.view_as_real(q)), Q, dbp, N, _lib.ptr(dist), _lib.ptr(ang),
                  _lib.ptr(corr) if want_corr else None, _lib.current_stream(d)))
    return (dist, ang, corr) if want_corr else (dist, ang)


TILED_ENTRY_FLOATS = 58624 // 4     # MRS_RING_TILED_ENTRY_BYTES: one DMA-tiled database entry


def spec_to_tiled(spec):
    """[n,61,120] (RING) or [n,C,61,120] (RING++) complex64 half spectra -> the DMA-tiled database format of the one-query sweep: float32
    [n + 1, C * 14656], n entries of C planes of 58 624 B and one entry of zero slack behind them (include/mrslam_hip.h: mrs_ring_spec_to_tiled)."""
    d = _dev(spec)
    x = spec.contiguous()
    assert x.dtype == torch.complex64 and x.shape[-2:] == (61, 120) and x.dim() in (3, 4)
    n = x.shape[0]
    planes = x.numel() // (61 * 120)
    out = torch.zeros((n + 1, (planes // n) * TILED_ENTRY_FLOATS), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mrs_ring_spec_to_tiled(_lib.ctx(d), _lib.ptr(torch.view_as_real(x)), planes, _lib.ptr(out), _lib.current_stream(d)))
    return out


def corr_sweep_fft_tiled(query_spec, tiled, n_db=None):
    """One query ([C,61,120] / [1,C,61,120] / [61,120] complex64, row layout) against a DMA-tiled database (spec_to_tiled): (dist [n], angle [n]),
    bit-identical to corr_sweep_fft(query, db)[...][0]."""
    d = _dev(query_spec)
    q = query_spec.contiguous()
    channels = tiled.shape[1] // TILED_ENTRY_FLOATS
    assert q.dtype == torch.complex64 and q.numel() == channels * 61 * 120 and tiled.dtype == torch.float32 and tiled.is_contiguous()
    n = int(n_db if n_db is not None else tiled.shape[0] - 1)
    assert 0 < n < tiled.shape[0], "the tiled array needs one entry of slack behind the last one"
    dist = torch.empty(n, dtype=torch.float32, device=q.device)
    ang = torch.empty(n, dtype=torch.int32, device=q.device)
    _lib.check(_lib.load().mrs_ring_corr_fft_sweep_tiled(_lib.ctx(d), _lib.ptr(torch.view_as_real(q)), _lib.ptr(tiled), n, int(channels), _lib.ptr(dist), _lib.ptr(ang),
                                                         _lib.current_stream(d)))
    return dist, ang


def corr_sweep_fft_tiled_q(query_specs, tiled, n_db=None):
    """Q queries ([Q,C,61,120] / [Q,61,120] complex64, row layout) against a DMA-tiled database in one call (mrs_ring_corr_fft_sweep_tiled_q: the
    one-query LDS-DMA pipeline, the queries' workgroups grouped per XCD): (dist [Q,n], angle [Q,n]), bit-identical to corr_sweep_fft(queries, db)."""
    d = _dev(query_specs)
    q = query_specs.contiguous()
    channels = tiled.shape[1] // TILED_ENTRY_FLOATS
    assert q.dtype == torch.complex64 and q.numel() % (channels * 61 * 120) == 0 and tiled.dtype == torch.float32 and tiled.is_contiguous()
    nq = q.numel() // (channels * 61 * 120)
    n = int(n_db if n_db is not None else tiled.shape[0] - 1)
    assert 0 < n < tiled.shape[0], "the tiled array needs one entry of slack behind the last one"
    dist = torch.empty((nq, n), dtype=torch.float32, device=q.device)
    ang = torch.empty((nq, n), dtype=torch.int32, device=q.device)
    _lib.check(_lib.load().mrs_ring_corr_fft_sweep_tiled_q(_lib.ctx(d), _lib.ptr(torch.view_as_real(q)), int(nq), _lib.ptr(tiled), n, int(channels),
                                                           _lib.ptr(dist), _lib.ptr(ang), _lib.current_stream(d)))
    return dist, ang


def corr_sweep_fft_blocks(spec_pool, query_rows, db_first, n_db, out=None, check=False):
    """Several C1 sweeps in one launch: query q = entry query_rows[q] of spec_pool ([E,61,120] complex64) against the n_db entries that start
    at entry db_first[q] (int64 device tensors).  Returns (dist [Q,n_db], angle [Q,n_db]); bit-identical to corr_sweep_fft per query.
    The kernel trusts the indices (they live on the device): check=True verifies them against the pool first (one host synchronisation)."""
    d = _dev(spec_pool)
    if check:
        E = spec_pool.shape[0]
        assert int(query_rows.min()) >= 0 and int(query_rows.max()) < E and int(db_first.min()) >= 0 and int(db_first.max()) + int(n_db) <= E, \
            "query row or database block outside the pool"
    assert spec_pool.dtype == torch.complex64 and spec_pool.is_contiguous() and spec_pool.shape[1:] == (61, 120)
    qr, df = query_rows.contiguous(), db_first.contiguous()
    assert qr.dtype == torch.int64 and df.dtype == torch.int64 and qr.numel() == df.numel()
    Q = qr.numel()
    dist, ang = out if out is not None else (torch.empty((Q, n_db), dtype=torch.float32, device=spec_pool.device),
                                             torch.empty((Q, n_db), dtype=torch.int32, device=spec_pool.device))
    _lib.check(_lib.load().mrs_ring_corr_fft_sweep_blocks(_lib.ctx(d), _lib.ptr(torch.view_as_real(spec_pool)), _lib.ptr(qr), Q, _lib.ptr(df), int(n_db),
                                                          _lib.ptr(dist), _lib.ptr(ang), _lib.current_stream(d)))
    return dist, ang


def corr_pairs_fft(a_spec, b_spec, out=None):
    """Pairwise C1/C2 on half spectra [P,61,120] (or RING++ [P,C,61,120]) complex64 -> (dist [P], angle [P])."""
    d = _dev(a_spec)
    a, b = a_spec.contiguous(), b_spec.contiguous()
    P = a.shape[0]
    dist, ang = out if out is not None else (torch.empty(P, dtype=torch.float32, device=a.device),
                                             torch.empty(P, dtype=torch.int32, device=a.device))
    if a.dim() == 4:
        assert a.shape == b.shape
        _lib.check(_lib.load().mrs_ring_corr_fft_pairs_mc(_lib.ctx(d), _lib.ptr(torch.view_as_real(a)),
                                                          _lib.ptr(torch.view_as_real(b)), P, int(a.shape[1]), _lib.ptr(dist),
                                                          _lib.ptr(ang), None, _lib.current_stream(d)))
        return dist, ang
    _lib.check(_lib.load().mrs_ring_corr_fft_pairs(_lib.ctx(d), _lib.ptr(torch.view_as_real(a)),
                                                   _lib.ptr(torch.view_as_real(b)), P, _lib.ptr(dist), _lib.ptr(ang),
                                                   None, _lib.current_stream(d)))
    return dist, ang
