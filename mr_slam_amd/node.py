"""Node-shaped twin of the LoopDetection nodes' candidate loop: a device-resident, append-as-you-go descriptor database and
`detect_loop_icp` with the reference's own signatures.

The reference (RING_ros/main_RING.py:126-238, main_RINGplusplus.py:126-236, disco_ros/main.py:276-321) keeps one Python list of
descriptors per robot, appends one entry per callback and scores the new scan against EVERY entry of the other robots' lists in a Python
loop -- one torch FFT correlation per stored entry.  Here the list has a twin on the device (C ABI `mrs_loopdb_*`, csrc/loopdb.hip): an append
writes one slot, and the whole candidate loop is ONE sweep launch that returns the reference's `idxs / dists / angles` lists.

Two ways in, both leave the node's callbacks, message handling and ICP untouched:

  * `detect_loop_icp = node.bind_detect_loop_icp(globals(), "ring")` after the node's own definition (one line):
    same signature, same prints, same `loopinfo.txt` lines and published messages; the candidate lists stay plain Python lists -- their
    device twins are kept in step by identity (entries appended since the last call are uploaded, nothing is re-uploaded);
  * `TIRING1 = node.DescriptorList("ring")`: a `list` whose `append` also writes the device slot (no per-call bookkeeping at all).
"""
import ctypes as C
import threading
import time
import weakref

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
# helpers the nodes define themselves (main_RING.py:60-104, util.py:51-82, 253-260, 378-385); used when the twin is bound without a node
def euler2rot(roll, pitch, yaw):
    R_x = np.array([[1, 0, 0], [0, np.cos(roll), -np.sin(roll)], [0, np.sin(roll), np.cos(roll)]])
    R_y = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
    R_z = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    return np.dot(R_z, np.dot(R_y, R_x))


def getSE3(x, y, yaw):
    R = np.eye(4)
    R[:3, :3] = euler2rot(0, 0, yaw)
    R[:3, 3] = np.array([x, y, 0])
    return R


def calculate_row_shift(shift, num_ring=120):
    return -shift if shift < num_ring // 2 else shift - num_ring


def fast_gicp(source, target, max_correspondence_distance=1.0, init_pose=np.eye(4)):
    """main_RING.py:81-104 on the drop-in pygicp"""
    from .compat import pygicp
    source = pygicp.downsample(source, 0.2)
    target = pygicp.downsample(target, 0.2)
    gicp = pygicp.FastGICP()
    gicp.set_input_target(target)
    gicp.set_input_source(source)
    gicp.set_num_threads(4)
    gicp.set_max_correspondence_distance(max_correspondence_distance)
    gicp.align(initial_guess=init_pose)
    fitness = gicp.get_fitness_score(1.0)
    return fitness, gicp.get_final_transformation()


class _Cfg:
    """RING_ros/config.py values the loop reads"""
    num_ring = 120
    num_sector = 120
    dist_threshold = 0.48
    icp_max_distance = 5.0
    icp_fitness_score = 0.22


class LoopResult:
    """What one detect_loop_icp call decided (the reference prints / publishes it; returned as well for callers that want it)"""
    __slots__ = ("idxs", "dists", "angles", "idx_matched", "dist", "init_pose", "fitness", "transform", "accepted", "id0", "id1")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)
        self.accepted = False


def bind_detect_loop_icp(ns=None, kind="ring", device="cuda:0", **overrides):
    """-> detect_loop_icp with the reference's signature for `kind` ("ring": main_RING.py:126, "ringpp": main_RINGplusplus.py:126,
    "disco": disco_ros/main.py:276).  `ns`: the node's globals() -- cfg, f (loopinfo file), pub, Loop, Loops, robotid_to_key,
    get_pose_msg_from_homo_matrix, fast_gicp, getSE3 are taken from it when present, so messages, file and ICP are the node's own; the
    candidate loop, the translation solve and (RING++) the BEV rotation / correlation run on the device.  Keyword overrides replace any of them."""
    ns = dict(ns or {})
    ns.update(overrides)
    from . import preprocess, ring
    cfg = ns.get("cfg", _Cfg)
    out = ns.get("print", print)
    _getSE3 = ns.get("getSE3", getSE3)
    _fast_gicp = ns.get("fast_gicp", fast_gicp)
    _key = ns.get("robotid_to_key", preprocess.robotid_to_key)
    dev_index = torch.device(device).index or 0

    def publish(res, robotid_current, idx_current, robotid_candidate, idx_matched, loop_transform, tag=""):
        """main_RING.py:206-236: invert, pose message, loopinfo.txt line, Loops message"""
        f, pub, Loop, Loops = ns.get("f"), ns.get("pub"), ns.get("Loop"), ns.get("Loops")
        pose_of = ns.get("get_pose_msg_from_homo_matrix")
        res.id0 = _key(robotid_current) + idx_current + 1
        res.id1 = _key(robotid_candidate) + idx_matched + 1
        loop_transform = np.linalg.inv(loop_transform)
        res.transform = loop_transform
        if pose_of is not None:
            pose = pose_of(loop_transform)
            line = [robotid_current, idx_current, robotid_candidate, idx_matched, pose.position.x, pose.position.y, pose.position.z,
                    pose.orientation.x, pose.orientation.y, pose.orientation.z, pose.orientation.w]
            if f is not None:
                f.write(' '.join(str(i) for i in line))
                f.write("\n")
            if Loop is not None and Loops is not None and pub is not None:
                Loop_msgs = Loops()
                Loop_msg = Loop()
                Loop_msg.id0, Loop_msg.id1, Loop_msg.pose = res.id0, res.id1, pose
                Loop_msgs.Loops.append(Loop_msg)
                pub.publish(Loop_msgs)
        out(tag + "Loop detected between id ", res.id0, " and id ", res.id1)

    def finish(res, robotid_current, idx_current, pc_current, robotid_candidate, pc_matched, trans_x, trans_y, rot_yaw):
        """main_RING.py:189-238 from the (x, y, yaw) estimate on"""
        trans_x_bev, trans_y_bev = -trans_y, trans_x
        trans_x_lidar, trans_y_lidar = -trans_x_bev, -trans_y_bev
        init_pose = np.linalg.inv(_getSE3(trans_x_lidar, trans_y_lidar, rot_yaw))
        res.init_pose = init_pose
        out("Loop detected.")
        out("Estimated translation: x: {}, y: {}, rotation: {}".format(trans_x_lidar, trans_y_lidar, rot_yaw))
        times = time.time()
        icp_fitness_score, loop_transform = _fast_gicp(pc_current, pc_matched, max_correspondence_distance=cfg.icp_max_distance, init_pose=init_pose)
        timee = time.time()
        res.fitness = icp_fitness_score
        out("ICP fitness score:", icp_fitness_score)
        out("ICP processed time:", timee - times, 's')
        if icp_fitness_score < cfg.icp_fitness_score and robotid_current != robotid_candidate:
            out("\033[32mICP fitness score is less than threshold, accept the loop.\033[0m")
            res.accepted = True
            publish(res, robotid_current, idx_current, robotid_candidate, res.idx_matched, loop_transform)
        else:
            out("\033[31mICP fitness score is larger than threshold, reject the loop.\033[0m")
        return res

    def detect_ring(robotid_current, idx_current, pc_current, RING_current, TIRING_current,
                    robotid_candidate, pc_candidates, RING_candidates, TIRING_candidates):
        res = LoopResult()
        db = twin_of(TIRING_candidates, "ring", device=dev_index)
        assert len(db) >= len(pc_candidates)
        RING_idxs, RING_dists, RING_angles = db.query(TIRING_current, cfg.dist_threshold)
        keep = RING_idxs < len(pc_candidates)              # `for idx in range(len(pc_candidates))`
        RING_idxs, RING_dists, RING_angles = RING_idxs[keep], RING_dists[keep], RING_angles[keep]
        res.idxs, res.dists, res.angles = RING_idxs, RING_dists, RING_angles
        if len(RING_dists) == 0:
            out("No loop detected.")
            return res
        idx_top1 = np.argsort(RING_dists)[0]
        dist = RING_dists[idx_top1]
        out("Top {} RING distance: ".format(1), dist)
        angle_matched = int(RING_angles[idx_top1])
        angle_matched_extra = angle_matched - cfg.num_ring // 2
        angle_matched_rad = angle_matched * 2 * np.pi / cfg.num_ring
        angle_matched_extra_rad = angle_matched_extra * 2 * np.pi / cfg.num_ring
        row_shift = calculate_row_shift(angle_matched, cfg.num_ring)
        row_shift_extra = calculate_row_shift(angle_matched_extra, cfg.num_ring)
        idx_matched = int(RING_idxs[idx_top1])
        res.idx_matched, res.dist = idx_matched, dist
        pc_matched = pc_candidates[idx_matched]
        RING_matched = torch.as_tensor(RING_candidates[idx_matched])
        RING_matched_shifted = torch.roll(RING_matched, row_shift, dims=1)
        RING_matched_shifted_extra = torch.roll(RING_matched, row_shift_extra, dims=1)
        x, y, error = ring.solve_translation(RING_current, RING_matched_shifted, angle_matched_rad, device)
        x_extra, y_extra, error_extra = ring.solve_translation(RING_current, RING_matched_shifted_extra, angle_matched_extra_rad, device)
        if error < error_extra:
            trans_x, trans_y, rot_yaw = x / cfg.num_sector * 140., y / cfg.num_ring * 140., angle_matched_rad
        else:
            trans_x, trans_y, rot_yaw = x_extra / cfg.num_sector * 140., y_extra / cfg.num_ring * 140., angle_matched_extra_rad
        return finish(res, robotid_current, idx_current, pc_current, robotid_candidate, pc_matched, trans_x, trans_y, rot_yaw)

    def detect_ringpp(robotid_current, idx_current, pc_current, bev_current, TIRING_current,
                      robotid_candidate, pc_candidates, bev_candidates, TIRING_candidates):
        res = LoopResult()
        channels = int(torch.as_tensor(TIRING_current).shape[0])
        db = twin_of(TIRING_candidates, "ringpp", channels=channels, device=dev_index)
        idxs, dists, angles = db.query(TIRING_current, cfg.dist_threshold)
        keep = idxs < len(pc_candidates)
        idxs, dists, angles = idxs[keep], dists[keep], angles[keep]
        res.idxs, res.dists, res.angles = idxs, dists, angles
        if len(dists) == 0:
            out("No loop detected.")
            return res
        idx_top1 = np.argsort(dists)[0]
        dist = dists[idx_top1]
        out("Top {} TIRING distance: ".format(1), dist)
        angle_matched = int(angles[idx_top1])
        angle_matched_extra = angle_matched - cfg.num_ring // 2
        angle_matched_rad = angle_matched * 2 * np.pi / cfg.num_ring
        angle_matched_extra_rad = angle_matched_extra * 2 * np.pi / cfg.num_ring
        idx_matched = int(idxs[idx_top1])
        res.idx_matched, res.dist = idx_matched, dist
        pc_matched = pc_candidates[idx_matched]
        bev_matched = torch.as_tensor(bev_candidates[idx_matched]).to(device)
        bev_cur = torch.as_tensor(bev_current).to(device)
        bev_current_rotated = ring.rotate_bev(bev_cur, angle_matched_rad)
        bev_current_rotated_extra = ring.rotate_bev(bev_cur, angle_matched_extra_rad)
        x, y, error = ring.solve_translation_bev(bev_current_rotated, bev_matched)
        x_extra, y_extra, error_extra = ring.solve_translation_bev(bev_current_rotated_extra, bev_matched)
        if error < error_extra:
            trans_x, trans_y, rot_yaw = x / cfg.num_sector * 140., y / cfg.num_ring * 140., angle_matched_rad
        else:
            trans_x, trans_y, rot_yaw = x_extra / cfg.num_sector * 140., y_extra / cfg.num_ring * 140., angle_matched_extra_rad
        return finish(res, robotid_current, idx_current, pc_current, robotid_candidate, pc_matched, trans_x, trans_y, rot_yaw)

    def detect_disco(robotid_current, idx_current, pc_current, DiSCO_current, fft_current,
                     robotid_candidate, pc_candidates, DiSCO_candidates, FFT_candidates):
        res = LoopResult()
        if len(DiSCO_candidates) <= 1:
            return res
        db = _disco_twin(DiSCO_candidates, FFT_candidates, dev_index)
        idx_top1_pc, d2, yaw_bin = db.query(np.asarray(DiSCO_current, np.float32).reshape(-1), torch.as_tensor(fft_current).reshape(40, 120))
        num_sector = getattr(cfg, "num_sector", 120)
        yaw_pc = (yaw_bin - num_sector // 2) / float(num_sector) * 360.
        pred_angle_rad = yaw_pc * np.pi / 180.
        init_pose_pc = _getSE3(0, 0, pred_angle_rad)
        res.idx_matched, res.dist, res.init_pose = idx_top1_pc, d2, init_pose_pc
        res.idxs, res.angles = np.array([idx_top1_pc]), np.array([yaw_bin])
        pc_matched_pc = pc_candidates[idx_top1_pc]
        fitness_pc, loop_transform = _fast_gicp(pc_current, pc_matched_pc, max_correspondence_distance=cfg.icp_max_distance, init_pose=init_pose_pc)
        res.fitness = fitness_pc
        out("fitness: ", fitness_pc)
        if fitness_pc < cfg.icp_fitness_score and robotid_current != robotid_candidate:
            out("ICP fitness score is less than threshold, accept the loop.")
            res.accepted = True
            publish(res, robotid_current, idx_current, robotid_candidate, idx_top1_pc, loop_transform, tag="DiSCO: ")
        else:
            out("DiSCO: ICP fitness score is larger than threshold, reject the loop.")
        return res

    return {"ring": detect_ring, "ringpp": detect_ringpp, "disco": detect_disco}[kind]
