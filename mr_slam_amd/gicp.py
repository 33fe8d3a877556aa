"""Batched GICP over the C ABI (rows G2-G6).  Host logic only; the kernels live in csrc/gicp.hip."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class GicpParams(C.Structure):
    _fields_ = [("k_correspondences", C.c_int32), ("max_iterations", C.c_int32),
                ("lm_max_iterations", C.c_int32), ("force_iterations", C.c_int32),
                ("max_correspondence_distance", C.c_double), ("rotation_epsilon", C.c_double),
                ("transformation_epsilon", C.c_double), ("lm_init_lambda_factor", C.c_double),
                ("voxel_resolution", C.c_double), ("voxel_neighbors", C.c_int32), ("reserved", C.c_int32),
                ("convergence_factor", C.c_double)]


def default_params():
    p = GicpParams()
    _lib.load().mrs_gicp_default_params(C.byref(p))
    return p


_tls = __import__("threading").local()


def _staging(points):
    """thread-local pinned float32 [>= points, 3] staging tensor, grown geometrically"""
    buf = getattr(_tls, "buf", None)
    if buf is None or buf.shape[0] < points:
        cap = max(1 << 16, 1 << int(points - 1).bit_length())
        buf = torch.empty((cap, 3), dtype=torch.float32).pin_memory()
        _tls.buf = buf
    return buf


class GicpBatch:
    """n_pairs independent (source, target) registrations advanced together on one GPU."""

    def __init__(self, n_pairs, device=0):
        self.n_pairs = int(n_pairs)
        self.device = device
        self._h = C.c_void_p()
        lib = _lib.load()
        lib.mrs_gicp_batch_last_nn_passes.restype = C.c_double
        lib.mrs_gicp_batch_last_searched_fraction.restype = C.c_double
        _lib.check(lib.mrs_gicp_batch_create(_lib.ctx(device), self.n_pairs, C.byref(self._h)))
        self.params = default_params()
        self._n = [None, None]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mrs_gicp_batch_destroy(self._h)
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        _lib.check(_lib.load().mrs_gicp_batch_set_params(self._h, C.byref(self.params)))

    def set_search(self, core):
        """1 (default): octree-cell leaves, per-query culling, certified neighbours; 2: without certificates; 3: round-4 kernel for the cold
        pass too; 0: the round-3 wave-shared traversal (A/B, cross-check)."""
        _lib.check(_lib.load().mrs_gicp_batch_set_search(self._h, int(core)))

    def _set(self, which, clouds):
        """clouds: list of [n_i, >=3] arrays (host) or a (device tensor [N, s], offsets) tuple."""
        if isinstance(clouds, tuple):
            pts, offs = clouds
            offs = np.ascontiguousarray(offs, dtype=np.int64)
        else:
            # host clouds (what the nodes hand over, float64 [n, 3] from pygicp.downsample): converted to float32 straight INTO a pinned staging
            # buffer (one pass instead of convert + concatenate + a pageable copy) and sent with one asynchronous copy; mrs_gicp_batch_set_clouds
            # synchronises the stream before it returns, so the buffer (one per thread: the callbacks run concurrently) is free again by then
            srcs = [np.asarray(c) for c in clouds]
            offs = np.zeros(len(srcs) + 1, np.int64)
            offs[1:] = np.cumsum([a.shape[0] for a in srcs])
            total = int(offs[-1])
            stage = _staging(total)
            view = stage.numpy()
            for a, lo, hi in zip(srcs, offs[:-1], offs[1:]):
                np.copyto(view[lo:hi], a[:, :3], casting="same_kind")
            pts = torch.empty((total, 3), dtype=torch.float32, device=f"cuda:{self.device}")
            pts.copy_(stage[:total], non_blocking=True)
        assert offs.size == self.n_pairs + 1
        pts = pts.contiguous()
        assert pts.is_cuda and pts.dtype == torch.float32
        _lib.check(_lib.load().mrs_gicp_batch_set_clouds(self._h, which, _lib.ptr(pts), int(pts.shape[1]),
                                                         _lib.ptr(offs), _lib.current_stream(self.device)))
        self._n[which] = offs

    def set_sources(self, clouds):
        self._set(0, clouds)

    def set_targets(self, clouds):
        self._set(1, clouds)

    def set_sources_from(self, store, ids, store_which=1):
        """Pair i's source := cloud ids[i] of `store` (a GicpBatch used as a submap store: set_targets(unique clouds) +
        compute_covariances(1)).  Sorted points, covariances and boxes are copied on the device, nothing is rebuilt."""
        self._set_from(0, store, ids, store_which)

    def set_targets_from(self, store, ids, store_which=1):
        self._set_from(1, store, ids, store_which)

    def _set_from(self, which, store, ids, store_which):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        assert ids.size == self.n_pairs
        _lib.check(_lib.load().mrs_gicp_batch_set_clouds_from(self._h, which, store._h, int(store_which), _lib.ptr(ids),
                                                              _lib.current_stream(self.device)))
        so = store._n[store_which]
        offs = np.zeros(self.n_pairs + 1, np.int64)
        offs[1:] = np.cumsum([so[i + 1] - so[i] for i in ids])
        self._n[which] = offs

    def compute_covariances(self, which, want_knn=False):
        knn = None
        if want_knn:
            knn = torch.empty((int(self._n[which][-1]), self.params.k_correspondences), dtype=torch.int32,
                              device=f"cuda:{self.device}")
        _lib.check(_lib.load().mrs_gicp_batch_compute_covariances(self._h, which, _lib.ptr(knn) if want_knn else None,
                                                                  _lib.current_stream(self.device)))
        return knn

    def covariances(self, which):
        """[N,3,3] float64 regularised covariances (host)."""
        n = int(self._n[which][-1])
        c6 = np.empty((n, 6), np.float64)
        _lib.check(_lib.load().mrs_gicp_batch_get_covariances(self._h, which, _lib.ptr(c6)))
        out = np.empty((n, 3, 3), np.float64)
        out[:, 0, 0], out[:, 0, 1], out[:, 0, 2] = c6[:, 0], c6[:, 1], c6[:, 2]
        out[:, 1, 0], out[:, 1, 1], out[:, 1, 2] = c6[:, 1], c6[:, 3], c6[:, 4]
        out[:, 2, 0], out[:, 2, 1], out[:, 2, 2] = c6[:, 2], c6[:, 4], c6[:, 5]
        return out

    def align(self, guesses=None):
        """Returns (T [P,4,4] float64, converged [P] bool, iterations [P] int32)."""
        P = self.n_pairs
        g = None
        if guesses is not None:
            g = np.ascontiguousarray(np.asarray(guesses, dtype=np.float64).reshape(P, 16))
        T = np.empty((P, 16), np.float64)
        conv = np.empty(P, np.int32)
        its = np.empty(P, np.int32)
        self.hessian = np.empty((P, 36), np.float64)
        _lib.check(_lib.load().mrs_gicp_batch_align(self._h, _lib.ptr(g) if g is not None else None, _lib.ptr(T),
                                                    _lib.ptr(conv), _lib.ptr(its), _lib.ptr(self.hessian),
                                                    _lib.current_stream(self.device)))
        self.nn_passes = float(_lib.load().mrs_gicp_batch_last_nn_passes(self._h))
        self.searched_fraction = float(_lib.load().mrs_gicp_batch_last_searched_fraction(self._h))
        return T.reshape(P, 4, 4), conv.astype(bool), its

    def linearize(self, poses, want_corr=False):
        P = self.n_pairs
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(P, 16))
        H = np.empty((P, 36), np.float64); b = np.empty((P, 6), np.float64); e = np.empty(P, np.float64)
        corr = torch.empty(int(self._n[0][-1]), dtype=torch.int32, device=f"cuda:{self.device}") if want_corr else None
        _lib.check(_lib.load().mrs_gicp_batch_linearize(self._h, _lib.ptr(poses), _lib.ptr(H), _lib.ptr(b), _lib.ptr(e),
                                                        _lib.ptr(corr) if want_corr else None,
                                                        _lib.current_stream(self.device)))
        return e, H.reshape(P, 6, 6), b, (corr.cpu().numpy() if want_corr else None)

    def profile(self, poses, reps=3):
        """HIP-event duration of every kernel of one outer iteration, launched alone at `poses` (mrs_gicp_batch_profile).
        Returns (dict of ms, dict of counts)."""
        P = self.n_pairs
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(P, 16))
        ms = np.zeros(8, np.float32); cnt = np.zeros(3, np.int64)
        _lib.check(_lib.load().mrs_gicp_batch_profile(self._h, _lib.ptr(poses), int(reps), _lib.ptr(ms), _lib.ptr(cnt),
                                                      _lib.current_stream(self.device)))
        names = ("linearize", "linearize_error_only", "search_round3_all", "certify", "search_round4_all", "knn_select", "cov_from_knn",
                 "certify_plus_worklist_1mm")
        return {n: float(v) for n, v in zip(names, ms)}, {"source_points": int(cnt[0]), "correspondences": int(cnt[1]), "worklist_queries_1mm": int(cnt[2])}

    def fitness(self, poses, max_range):
        P = self.n_pairs
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(P, 16))
        out = np.empty(P, np.float64)
        _lib.check(_lib.load().mrs_gicp_batch_fitness(self._h, _lib.ptr(poses), C.c_double(max_range), _lib.ptr(out),
                                                      _lib.current_stream(self.device)))
        return out
