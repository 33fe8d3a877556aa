"""Node-shaped twin of the LoopDetection nodes' candidate loop: a device-resident, append-as-you-go descriptor database and
`detect_loop_icp` with the reference's own signatures.

The reference (RING_ros/main_RING.py:126-238, main_RINGplusplus.py:126-236, disco_ros/main.py:276-321) keeps one Python list of
descriptors per robot, appends one entry per callback and scores the new scan against EVERY entry of the other robots' lists in a Python
loop -- one torch FFT correlation per stored entry.  Here the list has a twin on the device (C ABI `mrs_loopdb_*`, csrc/loopdb.hip): an append
writes one slot, and the whole candidate loop is ONE sweep launch that returns the reference's `idxs / dists / angles` lists.

Two ways in, both leave the node's callbacks, message handling and ICP untouched:

  * `detect_loop_icp = node.bind_detect_loop_icp(globals(), "ring")` after the node's own definition (one line):
    the node's OWN function keeps running (its glue, prints, `loopinfo.txt` lines and messages are its own code), only the names through which
    it scores the candidates are answered by one sweep of the list's device twin; the candidate lists stay plain Python lists -- their
    twins are kept in step by identity (entries appended since the last call are uploaded, nothing is re-uploaded);
  * no edit at all: `compat.install()` alone -- `fast_corr` keeps device twins of the host tensors it is handed (ring.DeviceMirror);
  * `TIRING1 = node.DescriptorList("ring")`: a `list` whose `append` also writes the device slot (no per-call bookkeeping at all).
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib

KIND = {"ring": 0, "ringpp": 1, "disco": 2}
FORM_HOST, FORM_DEVICE, FORM_DEVICE_SPEC = 0, 1, 2


def _as_arg(x, dtype, shape_tail):
    """descriptor -> (pointer, on_device flag, keep-alive object); accepts torch tensors (host / device) and numpy arrays"""
    if isinstance(x, torch.Tensor):
        t = x.detach()
        if t.dtype != dtype:
            t = t.to(dtype)
        t = t.contiguous()
        assert tuple(t.shape[-len(shape_tail):]) == tuple(shape_tail), (tuple(t.shape), shape_tail)
        if t.is_complex():
            t = torch.view_as_real(t)
        return C.c_void_p(t.data_ptr()), t.is_cuda, t
    a = np.ascontiguousarray(x, dtype={torch.complex64: np.complex64, torch.float32: np.float32}[dtype])
    assert tuple(a.shape[-len(shape_tail):]) == tuple(shape_tail), (a.shape, shape_tail)
    return C.c_void_p(a.ctypes.data), False, a


class LoopDatabase:
    """Device-resident descriptor list of one robot (mrs_loopdb).

    kind "ring"  : entries = what generate_RING returns as pc_TIRING (complex64 [1,120,120], util.py:198) or half spectra [1,61,120];
    kind "ringpp": entries = generate_RINGplusplus' pc_TIRING (float32 [C,120,120], util.py:247-250) or half spectra [C,61,120] of the
                   normalised channels;
    `query(descriptor, threshold)` = the loop of main_RING.py:133-140 as one sweep: (idxs, dists, angles) of the entries under the
    threshold, in index order."""

    def __init__(self, kind="ring", channels=None, device=0, capacity=1024):
        assert kind in ("ring", "ringpp")
        self.kind, self.device = kind, int(device)
        self.channels = 1 if kind == "ring" else int(channels or 6)
        self._h = C.c_void_p()
        lib = _lib.load()
        _lib.check(lib.mrs_loopdb_create(_lib.ctx(self.device), KIND[kind], self.channels, int(capacity), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mrs_loopdb_destroy(self._h)
        except Exception:
            pass

    def __len__(self):
        n = C.c_int32(0)
        _lib.check(_lib.load().mrs_loopdb_size(self._h, C.byref(n)))
        return n.value

    def _descriptor(self, x):
        """(pointer, form, keep-alive)"""
        C_ = self.channels
        is_spec = (isinstance(x, torch.Tensor) and x.is_cuda and x.is_complex() and tuple(x.shape[-2:]) == (61, 120))
        if is_spec:
            p, _, keep = _as_arg(x, torch.complex64, (C_, 61, 120) if x.dim() >= 3 else (61, 120))
            return p, FORM_DEVICE_SPEC, keep
        if self.kind == "ring":
            p, dev, keep = _as_arg(x, torch.complex64, (120, 120))
        else:
            p, dev, keep = _as_arg(x, torch.float32, (C_, 120, 120))
        return p, (FORM_DEVICE if dev else FORM_HOST), keep

    def _stream(self, form):
        return _lib.current_stream(self.device) if form != FORM_HOST else None

    def append(self, descriptor):
        p, form, keep = self._descriptor(descriptor)
        _lib.check(_lib.load().mrs_loopdb_append(self._h, p, form, 1, self._stream(form)))

    def extend_spectra(self, spectra):
        """`spectra`: device half spectra [n, C, 61, 120] (or [n, 61, 120] for RING) -- a batch producer's output, appended in one call"""
        assert spectra.is_cuda and spectra.dtype == torch.complex64 and spectra.is_contiguous()
        n = spectra.shape[0]
        _lib.check(_lib.load().mrs_loopdb_append(self._h, C.c_void_p(torch.view_as_real(spectra).data_ptr()), FORM_DEVICE_SPEC, int(n),
                                                 _lib.current_stream(self.device)))

    def query(self, descriptor, threshold, want_all=False):
        """-> (idxs int32[m], dists float32[m], angles int32[m]) with dist < threshold, in index order
        (+ (all_dists, all_angles) over every entry with want_all).
        Thread-safe: the node's callbacks query one twin from concurrent rospy threads (callback1 and callback3 both score against TIRING2) and
        ctypes drops the GIL during the call, so every call owns its output arrays; the entries scored are the n the C side reports (an
        append from another thread between sizing the arrays and the sweep can neither overflow nor truncate them silently)."""
        p, form, keep = self._descriptor(descriptor)
        lib = _lib.load()
        cap = max(len(self), 1) + 64                       # room for entries another thread appends before the sweep runs
        while True:
            idx, dist, ang = np.empty(cap, np.int32), np.empty(cap, np.float32), np.empty(cap, np.int32)
            alld = np.empty(cap, np.float32) if want_all else None
            alla = np.empty(cap, np.int32) if want_all else None
            cnt, n = C.c_int32(0), C.c_int32(0)
            _lib.check(lib.mrs_loopdb_query(self._h, p, form, C.c_float(threshold), int(cap), _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(ang),
                                            C.byref(cnt), int(cap) if want_all else 0, _lib.ptr(alld) if want_all else None,
                                            _lib.ptr(alla) if want_all else None, C.byref(n), self._stream(form)))
            if cnt.value <= cap and n.value <= cap:
                break
            cap = max(cnt.value, n.value) + 64             # the list grew by more than the slack meanwhile: once more, with room
        m = cnt.value
        out = (idx[:m], dist[:m], ang[:m])
        return out + (alld[:n.value], alla[:n.value]) if want_all else out

    def query_multi(self, descriptors):
        """Several new descriptors against every entry in ONE sweep (mrs_loopdb_query_multi): `descriptors` = a list / stacked tensor of
        descriptors in one of the forms `query` takes -> (dists float32 [Q, n], angles int32 [Q, n]); row q carries the bits of
        query(descriptors[q], want_all=True)."""
        if isinstance(descriptors, (list, tuple)):
            descriptors = torch.stack([torch.as_tensor(d) for d in descriptors])
        q = descriptors
        nq = int(q.shape[0])
        is_spec = q.is_cuda and q.is_complex() and tuple(q.shape[-2:]) == (61, 120)
        if is_spec:
            t = q.to(torch.complex64).contiguous()
            assert t.numel() == nq * self.channels * 61 * 120
            form = FORM_DEVICE_SPEC
        elif self.kind == "ring":
            t = q.to(torch.complex64).contiguous()
            assert t.numel() == nq * 120 * 120
            form = FORM_DEVICE if t.is_cuda else FORM_HOST
        else:
            t = q.to(torch.float32).contiguous()
            assert t.numel() == nq * self.channels * 120 * 120
            form = FORM_DEVICE if t.is_cuda else FORM_HOST
        ptr = C.c_void_p((torch.view_as_real(t) if t.is_complex() else t).data_ptr())
        lib = _lib.load()
        cap = max(len(self), 1) + 64
        while True:
            alld, alla = np.empty((nq, cap), np.float32), np.empty((nq, cap), np.int32)
            n = C.c_int32(0)
            _lib.check(lib.mrs_loopdb_query_multi(self._h, ptr, form, nq, int(cap), _lib.ptr(alld), _lib.ptr(alla), C.byref(n), self._stream(form)))
            if n.value <= cap:
                break
            cap = n.value + 64
        return alld[:, :n.value], alla[:, :n.value]

    def device_entries(self):
        """(device pointer, n, floats per entry) of the stored entries (tests)"""
        p, n, ef = C.c_void_p(), C.c_int32(0), C.c_int64(0)
        _lib.check(_lib.load().mrs_loopdb_device_entries(self._h, C.byref(p), None, C.byref(n), C.byref(ef)))
        return p.value, n.value, ef.value


class DiscoDatabase:
    """DiSCO twin: `DiSCO<k>` (1024-d signatures) + `FFT<k>` (complex64 [1,1,40,120] spectra) of one robot on the device;
    `query(signature, spectrum)` = disco_ros/main.py:284-291 (nearest signature + phase_corr of the winner) in two launches."""

    def __init__(self, device=0, capacity=1024):
        self.device = int(device)
        self._h = C.c_void_p()
        _lib.check(_lib.load().mrs_loopdb_create(_lib.ctx(self.device), KIND["disco"], 1, int(capacity), C.byref(self._h)))

    __del__ = LoopDatabase.__del__
    __len__ = LoopDatabase.__len__

    def _args(self, signature, spectrum):
        ps, dev_s, k1 = _as_arg(torch.as_tensor(signature).reshape(-1), torch.float32, (1024,))
        pf, dev_f, k2 = _as_arg(spectrum, torch.complex64, (40, 120))
        if dev_s != dev_f:       # one side on the host: bring the signature (4 KB) where the spectrum is
            ps, dev_s, k1 = _as_arg(k1.to(k2.device) if isinstance(k1, torch.Tensor) else torch.from_numpy(k1).to(k2.device), torch.float32, (1024,))
        return ps, pf, int(dev_s), (k1, k2)

    def append(self, signature, spectrum):
        ps, pf, dev, keep = self._args(signature, spectrum)
        _lib.check(_lib.load().mrs_loopdb_append_disco(self._h, ps, pf, dev, _lib.current_stream(self.device) if dev else None))

    def query(self, signature, spectrum, num_sector=120):
        """-> (index, squared distance, yaw bin) of the nearest stored signature; index -1 for an empty database"""
        ps, pf, dev, keep = self._args(signature, spectrum)
        idx, d2, arg = C.c_int32(-1), C.c_float(0), C.c_int32(0)
        _lib.check(_lib.load().mrs_loopdb_query_disco(self._h, ps, pf, dev, C.byref(idx), C.byref(d2), C.byref(arg),
                                                      _lib.current_stream(self.device) if dev else None))
        return idx.value, d2.value, arg.value % num_sector


class DescriptorList(list):
    """`TIRING1 = DescriptorList("ring")` instead of `TIRING1 = []`: a list (the node indexes it for the matched entry) whose
    append also writes the device slot."""

    def __init__(self, kind="ring", channels=None, device=0, capacity=1024):
        super().__init__()
        self.db = LoopDatabase(kind, channels, device, capacity)

    def append(self, descriptor):
        super().append(descriptor)
        self.db.append(descriptor)


# device twins of PLAIN Python lists, by identity: the node's `TIRING2.append(...)` stays as it is, entries that arrived since the last
# call are uploaded at the next one (one 58 KB copy each for RING)
_twins = {}
_twins_lock = threading.Lock()


def twin_of(candidates, kind, channels=None, device=0):
    if isinstance(candidates, DescriptorList):
        return candidates.db
    with _twins_lock:
        ent = _twins.get(id(candidates))
        if ent is None or ent[0]() is not candidates:
            class _Ref:          # plain lists cannot be weak-referenced: remember the object itself (the node's lists live as long as the node)
                def __init__(self, o): self.o = o
                def __call__(self): return self.o
            ent = (_Ref(candidates), LoopDatabase(kind, channels, device))
            _twins[id(candidates)] = ent
        db = ent[1]
        for i in range(len(db), len(candidates)):
            db.append(candidates[i])
        return db


def _disco_twin(sigs, ffts, device=0):
    with _twins_lock:
        ent = _twins.get(("disco", id(sigs), id(ffts)))
        if ent is None or ent[0] is not sigs or ent[1] is not ffts:
            ent = (sigs, ffts, DiscoDatabase(device))
            _twins[("disco", id(sigs), id(ffts))] = ent
        db = ent[2]
        for i in range(len(db), len(sigs)):
            db.append(sigs[i], ffts[i])
        return db


# ---------------------------------------------------------------------------------------------------------------------------------------
# Binding the twin to a node.  Rounds 1-5 restated the ~55 lines of host glue that follow the candidate loop (top-1 pick, the two rotation
# hypotheses, BEV -> lidar frame, ICP gate, loopinfo.txt line, Loops message: main_RING.py:147-236) with the reference's own names; a fix in
# the node's glue would have silently diverged from that copy.  Now the NODE'S OWN detect_loop_icp runs, unchanged, from its own
# code object; only the names through which it reaches the candidates are bound to the twin for the duration of the call:
#   ring   : `fast_corr`               answers entry idx of a table filled by ONE sweep of the device-resident list before the function starts;
#   ringpp : `fast_corr_RINGplusplus`  likewise;
#   disco  : `KDTree` / `phase_corr`   answer from the two-launch query (nearest signature + phase correlation of the winner).
# Everything else -- cfg, f, pub, Loop, Loops, get_pose_msg_from_homo_matrix, fast_gicp, solve_translation, every print -- is looked up in the
# node's LIVE globals at the moment it is used (names the node defines after the binding line, like `f` and `pub` in its __main__ block, resolve
# too).  The node's Python loop over the candidates remains (~0.25 us per entry: a table read and a comparison); LoopDatabase.query is the
# form without it.
import types


class _Overlay(dict):
    """globals of the re-bound function: the few substituted names live here, every other lookup falls through to the node's own globals"""

    def __init__(self, base):
        super().__init__()
        self._base = base

    def __missing__(self, key):
        return self._base[key]


def bind_detect_loop_icp(ns, kind="ring", device="cuda:0", **overrides):
    """-> detect_loop_icp with the node's own signature and behaviour for `kind` ("ring": main_RING.py:126, "ringpp":
    main_RINGplusplus.py:126, "disco": disco_ros/main.py:276).  `ns`: the node's globals() (it must hold the node's detect_loop_icp, or pass
    detect_loop_icp=...); keyword overrides replace further names the function uses."""
    fn = overrides.pop("detect_loop_icp", None) or ns["detect_loop_icp"]
    ov = _Overlay(fn.__globals__)                            # == ns when the node passes globals()
    ov.update(overrides)
    import inspect
    sig = inspect.signature(fn)
    tls = threading.local()                                  # the table of the call in flight: the node's callbacks run on concurrent threads
    dev_index = torch.device(device).index or 0

    def rebind():
        return types.FunctionType(fn.__code__, ov, fn.__name__, fn.__defaults__, fn.__closure__)

    if kind in ("ring", "ringpp"):
        name = "fast_corr" if kind == "ring" else "fast_corr_RINGplusplus"
        own = ov[name] if name in overrides else None

        def from_table(a, b):
            t = tls.table
            i = t[0]
            t[0] = i + 1
            if i < len(t[1]) and b is t[3][i] and a is t[4]:
                return t[1][i], t[2][i]
            return (own or fn.__globals__[name])(a, b)       # not the loop's pattern: the node's own function answers
        ov[name] = from_table
        node_fn = rebind()

        def detect_loop_icp(*args, **kwargs):
            a = list(sig.bind(*args, **kwargs).arguments.values())      # (..., TIRING_current = a[4], ..., pc_candidates = a[6], ..., TIRING_candidates = a[8])
            TIRING_current, pc_candidates, TIRING_candidates = a[4], a[6], a[8]
            alld = alla = ()
            if len(pc_candidates):
                channels = int(torch.as_tensor(TIRING_current).shape[0]) if kind == "ringpp" else None
                db = twin_of(TIRING_candidates, kind, channels=channels, device=dev_index)
                _, _, _, alld, alla = db.query(TIRING_current, -1.0, want_all=True)       # every entry's score, one sweep
                alla = alla.astype(np.int64)                # the reference's angle is an int64 (util.py:372)
            tls.table = [0, alld, alla, TIRING_candidates, TIRING_current]
            try:
                return node_fn(*args, **kwargs)
            finally:
                tls.table = None
        detect_loop_icp.__signature__ = sig
        detect_loop_icp.__wrapped__ = fn
        return detect_loop_icp

    assert kind == "disco", kind

    class _TwinTree:
        """`KDTree(np.array(DiSCO_candidates)).query(DiSCO_current.reshape(1, -1), k=1)` (main.py:284-285) from the query made before the call"""

        def __init__(self, data, *a, **k):
            pass

        def query(self, x, k=1, **kw):
            idx, d2, _ = tls.q
            return np.array([[np.sqrt(max(d2, 0.0))]]), np.array([[idx]])

    def phase_from_query(FFT_candidate, fft_current, *a, **k):
        return np.array(tls.q[2]), None                       # main.py:260-272 returns (angle % num_sector as a 0-d array, corr)
    ov["KDTree"] = _TwinTree
    ov["phase_corr"] = phase_from_query
    node_fn = rebind()

    def detect_loop_icp(*args, **kwargs):
        a = list(sig.bind(*args, **kwargs).arguments.values())          # (..., DiSCO_current = a[3], fft_current = a[4], ..., DiSCO_candidates = a[7], FFT_candidates = a[8])
        DiSCO_current, fft_current, DiSCO_candidates, FFT_candidates = a[3], a[4], a[7], a[8]
        if len(DiSCO_candidates) > 1:
            db = _disco_twin(DiSCO_candidates, FFT_candidates, dev_index)
            num_sector = getattr(ov["cfg"], "num_sector", 120)
            tls.q = db.query(np.asarray(DiSCO_current, np.float32).reshape(-1), torch.as_tensor(fft_current).reshape(40, 120), num_sector)
        try:
            return node_fn(*args, **kwargs)
        finally:
            tls.q = None
    detect_loop_icp.__signature__ = sig
    detect_loop_icp.__wrapped__ = fn
    return detect_loop_icp
