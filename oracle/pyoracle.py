"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end to oracle/liboracle.so (the CPU restatements) and, when present,
oracle/_ref/libref_polar.so (the reference's own CPU polar rasteriser compiled in place).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None
DROP = np.int32(-2**31)


def build(quiet=True):
    """(Re)build liboracle.so and, if /root/reference exists, _ref/."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_occupied_fingerprint.restype = C.c_uint64
    return _LIB


def ref_polar():
    """The reference library, or None if it was never built (no /root/reference)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_polar.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


_REFS = {}


def ref_lib(name):
    """oracle/_ref/libref_<name>.so (the reference's own sources compiled by oracle/Makefile), or None if it was
    never built (no /root/reference at build time)."""
    if name not in _REFS:
        path = os.path.join(_HERE, "_ref", f"libref_{name}.so")
        _REFS[name] = C.CDLL(path) if os.path.exists(path) else None
    return _REFS[name]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def polar_height_fastpath_check(num_height, max_height, eps):
    """Every finite float z through the polar rasterisers' height fast path against the reference's float formula
    (oracle/fastpath_oracle.c).  Returns (accepted, mismatches, an offending z or 0.0)."""
    f = lib().mrs_polar_height_check
    f.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    acc, bad, v = C.c_uint64(), C.c_uint64(), C.c_float()
    f(int(num_height), int(max_height), float(eps), C.byref(acc), C.byref(bad), C.byref(v))
    return acc.value, bad.value, v.value


def cart_fastpath_check(bins, eps, max_length=1, bits_lo=1, bits_hi=0x3F800000, axis_form=False):
    """Walks every float with magnitude bits in [bits_lo, bits_hi] (default: every non-zero |v| <= 1, both signs) through the
    Cartesian rasterisers' fp32 fast path as the device evaluates it and compares with the reference's double formula
    (oracle/fastpath_oracle.c); axis_form: cart_axis()'s (v + 1.0f) * inv instead of the fma.  Returns (accepted, mismatches,
    max |g - q| in bins, an offending value or 0.0)."""
    f = lib().mrs_axispath_check if axis_form else lib().mrs_fastpath_check
    f.argtypes = [C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                  C.POINTER(C.c_double), C.POINTER(C.c_float)]
    acc, bad, worst, v = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_float()
    f(int(bins), int(max_length), float(eps), int(bits_lo), int(bits_hi), C.byref(acc), C.byref(bad), C.byref(worst), C.byref(v))
    return acc.value, bad.value, worst.value, v.value


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- polar BEV
def bev_polar_indices(xyz_soa, max_length, max_height, R, S, H):
    xyz = _f32(xyz_soa)
    n = xyz.size // 3
    ring = np.empty(n, np.int32); sector = np.empty(n, np.int32); height = np.empty(n, np.int32)
    valid = np.empty(n, np.uint8)
    lib().orc_bev_polar_indices(_p(xyz), n, max_length, max_height, R, S, H,
                                _p(ring), _p(sector), _p(height), _p(valid))
    return ring, sector, height, valid.astype(bool)


def bev_polar(xyz_soa, max_length, max_height, R, S, H, enough_large=1):
    xyz = _f32(xyz_soa)
    n = xyz.size // 3
    ring, sector, height, _ = bev_polar_indices(xyz, max_length, max_height, R, S, H)
    out = np.zeros(3 * R * S * H * enough_large, np.float32)
    lib().orc_bev_polar_scatter(_p(xyz), n, _p(ring), _p(sector), _p(height), R, S, H,
                                enough_large, _p(out))
    return out


def ref_bev_polar(xyz_soa, max_length, max_height, R, S, H, enough_large=1):
    r = ref_polar()
    if r is None:
        raise RuntimeError("oracle/_ref/libref_polar.so not built")
    xyz = _f32(xyz_soa).copy()
    n = xyz.size // 3
    out = np.zeros(3 * R * S * H * enough_large, np.float32)
    r.ref_polar_bev(_p(xyz), n, max_length, max_height, R, S, H, enough_large, _p(out))
    return out


def ref_bev_polar_indices(xyz_soa, max_length, max_height, R, S, H):
    r = ref_polar()
    if r is None:
        raise RuntimeError("oracle/_ref/libref_polar.so not built")
    xyz = _f32(xyz_soa).copy()
    n = xyz.size // 3
    ring = np.empty(n, np.int32); sector = np.empty(n, np.int32); height = np.empty(n, np.int32)
    r.ref_polar_indices(_p(xyz), n, max_length, max_height, R, S, H,
                        _p(ring), _p(sector), _p(height))
    return ring, sector, height


# ------------------------------------------------------------------------- Cartesian BEV
def bev_cart_indices(xyz_soa, max_length, max_height, NX, NY, H):
    xyz = _f32(xyz_soa)
    n = xyz.size // 3
    ix = np.empty(n, np.int32); iy = np.empty(n, np.int32); ih = np.empty(n, np.int32)
    valid = np.empty(n, np.uint8)
    lib().orc_bev_cart_indices(_p(xyz), n, max_length, max_height, NX, NY, H,
                               _p(ix), _p(iy), _p(ih), _p(valid))
    return ix, iy, ih, valid.astype(bool)


def bev_cart(xyz_soa, max_length, max_height, NX, NY, H):
    xyz = _f32(xyz_soa)
    n = xyz.size // 3
    ix, iy, ih, _ = bev_cart_indices(xyz, max_length, max_height, NX, NY, H)
    out = np.zeros(3 * NX * NY * H, np.float32)
    lib().orc_bev_cart_scatter(_p(xyz), n, _p(ix), _p(iy), _p(ih), NX, NY, H, _p(out))
    return out


def bev_feat(pts_cm, F, max_length, max_height, NX, NY, H):
    pts = _f32(pts_cm)
    n = pts.size // F
    out = np.zeros(NX * NY * H * F, np.float32)
    lib().orc_bev_feat(_p(pts), n, F, max_length, max_height, NX, NY, H, _p(out))
    return out


def ref_bev_cart(xyz_soa, max_length, max_height, NX, NY, H, enough_large=1):
    """The reference's voxelocc.GPUTransformer(...).transform(); .retreive() (generate_bev_cython_binary) on the host."""
    r = ref_lib("cart")
    if r is None:
        raise RuntimeError("oracle/_ref/libref_cart.so not built")
    xyz = _f32(xyz_soa).copy()
    out = np.zeros(3 * NX * NY * H * enough_large, np.float32)
    r.ref_cart_bev(_p(xyz), xyz.size // 3, max_length, max_height, NX, NY, H, enough_large, _p(out))
    return out


def ref_bev_cart_indices(xyz_soa, max_length, max_height, NX, NY, H):
    r = ref_lib("cart")
    if r is None:
        raise RuntimeError("oracle/_ref/libref_cart.so not built")
    xyz = _f32(xyz_soa).copy()
    n = xyz.size // 3
    ix = np.zeros(n, np.int32); iy = np.zeros(n, np.int32); ih = np.zeros(n, np.int32)
    r.ref_cart_indices(_p(xyz), n, max_length, max_height, NX, NY, H, _p(ix), _p(iy), _p(ih))
    return ix, iy, ih


def ref_bev_feat(pts_cm, F, max_length, max_height, NX, NY, H):
    """The reference's voxelfeat.GPUTransformer (generate_bev_pointfeat_cython), threads run in gid order."""
    r = ref_lib("feat")
    if r is None:
        raise RuntimeError("oracle/_ref/libref_feat.so not built")
    pts = _f32(pts_cm).copy()
    out = np.zeros(NX * NY * H * F, np.float32)
    r.ref_feat_bev(_p(pts), pts.size // F, max_length, max_height, NX, NY, H, F, _p(out))
    return out


def ref_point_features(pts, knn, eigens):
    """The reference's voxelfeat.GPUFeatureExtractor(pts, n, 13, k, knn, eigens).get_features() -> [n,13]."""
    r = ref_lib("feat")
    if r is None:
        raise RuntimeError("oracle/_ref/libref_feat.so not built")
    p = _f32(pts).reshape(-1).copy()
    knn = np.ascontiguousarray(knn, dtype=np.int32)
    e = _f32(eigens).reshape(-1).copy()
    n, k = knn.shape
    out = np.zeros(n * 13, np.float32)
    r.ref_point_features(_p(p), n, 13, k, _p(knn), _p(e), _p(out))
    return out.reshape(n, 13)


def occupied_fingerprint(out3):
    out3 = _f32(out3)
    cnt = C.c_int64(0)
    h = lib().orc_occupied_fingerprint(_p(out3), C.c_int64(out3.size // 3), C.byref(cnt))
    return int(h), int(cnt.value)


# ---------------------------------------------------------------------------------- Radon
def set_omp_threads(n):
    """threads of the OpenMP loops of the C restatements (Radon over images) from now on; OMP_NUM_THREADS is only read when libgomp starts"""
    lib()
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def radon_parallel(img, angles, det, spacing=1.0):
    """img [B,H,W] or [H,W] float32 -> sinogram [B,n_angles,det] (orc_radon_parallel)."""
    img = _f32(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[None]
    B, H, W = img.shape
    ang = _f32(angles)
    out = np.empty((B, ang.size, det), np.float32)
    lib().orc_radon_parallel_batch(_p(img), B, H, W, _p(ang), ang.size, det, C.c_float(spacing), _p(out))
    return out[0] if squeeze else out


_SYM = None


def ref_symbolic():
    """The reference's analytic Radon checker (torch-radon/src/symbolic.cpp), or None."""
    global _SYM
    if _SYM is None:
        path = os.path.join(_HERE, "_ref", "libref_symbolic.so")
        if not os.path.exists(path):
            return None
        _SYM = C.CDLL(path)
        _SYM.ref_sym_create.restype = C.c_void_p
    return _SYM


class RefSymbolicFunction:
    """Mirror of torch_radon_cuda.SymbolicFunction as the reference tests use it
    (torch-radon/tests/utils.py:17-32)."""

    def __init__(self, h, w):
        self._l = ref_symbolic()
        if self._l is None:
            raise RuntimeError("oracle/_ref/libref_symbolic.so not built")
        self._f = C.c_void_p(self._l.ref_sym_create(C.c_float(h), C.c_float(w)))

    def __del__(self):
        if getattr(self, "_f", None):
            self._l.ref_sym_destroy(self._f)

    def add_gaussian(self, k, cx, cy, a, b):
        self._l.ref_sym_add_gaussian(self._f, *(C.c_float(v) for v in (k, cx, cy, a, b)))

    def add_ellipse(self, k, cx, cy, r, a):
        self._l.ref_sym_add_ellipse(self._f, *(C.c_float(v) for v in (k, cx, cy, r, a)))

    def discretize(self, h, w):
        out = np.zeros((h, w), np.float32)
        self._l.ref_sym_discretize(self._f, _p(out), h, w)
        return out

    def forward(self, angles, det, spacing=1.0):
        ang = _f32(angles)
        out = np.zeros((ang.size, det), np.float32)
        self._l.ref_sym_forward(self._f, det, C.c_float(spacing), _p(ang), ang.size, _p(out))
        return out


# ----------------------------------------------------------------------------------- GICP
class Gicp:
    """ctypes handle on oracle/gicp_oracle.cpp (restated fast_gicp FastGICP; parity unpinned)."""

    def __init__(self, k=20, max_corr=1e300, max_iter=64, rot_eps=2e-3, trans_eps=5e-4, threads=None, conv_factor=10.0):
        L = lib()
        L.orc_gicp_create.restype = C.c_void_p
        L.orc_gicp_linearize.restype = C.c_double
        L.orc_gicp_fitness.restype = C.c_double
        self._l = L
        self._h = C.c_void_p(L.orc_gicp_create())
        self.set_params(k, max_corr, max_iter, rot_eps, trans_eps, threads or os.cpu_count() or 1)
        L.orc_gicp_set_conv_factor(self._h, C.c_double(conv_factor))

    @property
    def nn_passes(self):
        """update_correspondences calls of the last align (upstream: one per outer iteration)."""
        return int(self._l.orc_gicp_nn_passes(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.orc_gicp_destroy(self._h)

    def set_params(self, k, max_corr, max_iter, rot_eps, trans_eps, threads):
        self._l.orc_gicp_set_params(self._h, int(k), C.c_double(max_corr), int(max_iter), C.c_double(rot_eps),
                                    C.c_double(trans_eps), int(threads))

    def set_voxel(self, resolution, neighbors=1):
        """row G7: voxelised target (FastVGICP); resolution 0 switches back to GICP."""
        self._l.orc_gicp_set_voxel(self._h, C.c_double(resolution), int(neighbors))

    def set_source(self, pts):
        p = _f32(np.asarray(pts)[:, :3]); self._ns = p.shape[0]
        self._l.orc_gicp_set_source(self._h, _p(p), p.shape[0])

    def set_target(self, pts):
        p = _f32(np.asarray(pts)[:, :3]); self._nt = p.shape[0]
        self._l.orc_gicp_set_target(self._h, _p(p), p.shape[0])

    def covariances(self, which):
        n = self._nt if which else self._ns
        out = np.empty((n, 3, 3), np.float64)
        self._l.orc_gicp_covariances(self._h, int(which), _p(out))
        return out

    def align(self, guess=None, force_iters=0):
        g = np.ascontiguousarray(np.eye(4) if guess is None else guess, dtype=np.float64)
        out = np.empty((4, 4), np.float64)
        conv = self._l.orc_gicp_align(self._h, _p(g), _p(out), int(force_iters))
        return out, bool(conv), int(self._l.orc_gicp_iterations(self._h)), int(self._l.orc_gicp_lm_trials(self._h))

    def linearize(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        H = np.empty((6, 6), np.float64); b = np.empty(6, np.float64)
        corr = np.empty(self._ns, np.int32)
        e = self._l.orc_gicp_linearize(self._h, _p(T), _p(H), _p(b), _p(corr))
        return float(e), H, b, corr

    def fitness(self, T, max_range):
        T = np.ascontiguousarray(T, dtype=np.float64)
        return float(self._l.orc_gicp_fitness(self._h, _p(T), C.c_double(max_range)))


def knn(pts, k):
    p = _f32(np.asarray(pts)[:, :3])
    out = np.empty((p.shape[0], k), np.int32)
    lib().orc_knn(_p(p), p.shape[0], int(k), _p(out))
    return out


def pair_d2(src, T, tgt, idx):
    """float32 squared distance of every float-transformed source point to tgt[idx[i]] (same operation chain as the
    searches; -1 indices give inf): lets a test prove that two different neighbour indices are an exact tie."""
    s = _f32(np.asarray(src)[:, :3]); t = _f32(np.asarray(tgt)[:, :3])
    T = np.ascontiguousarray(T, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(s.shape[0], np.float32)
    lib().orc_pair_d2(_p(s), s.shape[0], _p(T), _p(t), _p(idx), _p(out))
    return out


def approx_voxel_grid(points, leaf):
    """pygicp.downsample(points, leaf) = pcl::ApproximateVoxelGrid restated (oracle/voxel_oracle.c): float64 [n,3] in,
    float64 [m,3] out (float precision, flush order)."""
    p = _f32(np.asarray(points)[:, :3])
    out = np.empty((p.shape[0], 3), np.float32)
    m = lib().orc_approx_voxel_grid(_p(p), p.shape[0], C.c_float(leaf), _p(out))
    return out[:m].astype(np.float64)


def se3_exp(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    T = np.empty((4, 4), np.float64)
    lib().orc_se3_exp(_p(a), _p(T))
    return T


# ------------------------------------------------------------------------------ elevation map
class ElevMap:
    """ctypes handle on oracle/elev_oracle.cpp (sequential restatement of gpu_process.cu; pinned to the reference source
    built for the host, RefElevMap below, by tests/test_oracle_elev.py)."""

    def __init__(self, length, resolution, mahal=2.0, obstacle=0.6):
        L = lib()
        L.orc_elev_create.restype = C.c_void_p
        self._l, self.L = L, int(length)
        self._h = C.c_void_p(L.orc_elev_create(self.L, C.c_float(resolution), C.c_float(mahal), C.c_float(obstacle)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.orc_elev_destroy(self._h)

    def move(self, pos3):
        p = _f32(pos3); c = np.zeros(2, np.float32); s = np.zeros(2, np.int32); a = np.zeros(2, np.float32)
        self._l.orc_elev_move(self._h, _p(p), _p(c), _p(s), _p(a))
        return c, s, a

    def process_points(self, x, y, z, T, lower, upper, min_r, beam_a, beam_c, sj, rv, csb, pmul, bskew):
        x, y, z = _f32(x).copy(), _f32(y).copy(), _f32(z).copy()
        n = x.size
        mi = np.empty(n, np.int32)
        var, xt, yt, zt = (np.empty(n, np.float32) for _ in range(4))
        a = [_f32(T).reshape(16), _f32(sj).reshape(3), _f32(rv).reshape(9), _f32(csb).reshape(9), _f32(pmul).reshape(3), _f32(bskew).reshape(9)]
        self._l.orc_elev_process_points(self._h, n, _p(x), _p(y), _p(z), _p(a[0]), C.c_double(lower), C.c_double(upper),
                                        C.c_float(min_r), C.c_float(beam_a), C.c_float(beam_c), _p(a[1]), _p(a[2]), _p(a[3]),
                                        _p(a[4]), _p(a[5]), _p(mi), _p(var), _p(xt), _p(yt), _p(zt))
        return dict(map_index=mi, x=x, y=y, z=z, var=var, x_ts=xt, y_ts=yt, z_ts=zt)

    def fuse(self, index, cr, cg, cb, inten, h, v):
        arrs = [np.ascontiguousarray(index, np.int32), np.ascontiguousarray(cr, np.int32), np.ascontiguousarray(cg, np.int32),
                np.ascontiguousarray(cb, np.int32), _f32(inten), _f32(h), _f32(v)]
        self._l.orc_elev_fuse(self._h, arrs[0].size, *[_p(a) for a in arrs])

    def mapvar_update(self, v):
        self._l.orc_elev_mapvar_update(self._h, C.c_float(v))

    def map_feature(self):
        n = self.L * self.L
        f = {k: np.zeros(n, np.float32) for k in ("elevation", "var", "rough", "slope", "traver", "intensity")}
        f["traver"][:] = -10
        c = {k: np.zeros(n, np.int32) for k in ("colorR", "colorG", "colorB")}
        self._l.orc_elev_map_feature(self._h, _p(f["elevation"]), _p(f["var"]), _p(c["colorR"]), _p(c["colorG"]), _p(c["colorB"]),
                                     _p(f["rough"]), _p(f["slope"]), _p(f["traver"]), _p(f["intensity"]))
        f.update(c)
        return f

    def raytracing(self):
        self._l.orc_elev_raytracing(self._h)

    def map_optmove(self, p, dh):
        a = np.zeros(2, np.float32)
        self._l.orc_elev_map_optmove(self._h, _p(_f32(p)), C.c_float(dh), _p(a))
        return a

    def map_closeloop(self, p, dh):
        self._l.orc_elev_map_closeloop(self._h, _p(_f32(p)), C.c_float(dh))

    def layer(self, which):
        out = np.empty(self.L * self.L, np.float32)
        self._l.orc_elev_get(self._h, int(which), _p(out))
        return out

    def frame(self):
        c = np.zeros(2, np.float32); s = np.zeros(2, np.int32)
        self._l.orc_elev_get_frame(self._h, _p(c), _p(s))
        return c, s


class RefElevMap:
    """The reference's own gpu_process.cu built for the host (oracle/_ref/libref_elev.so, oracle/ref_elev_shim.cpp), same
    methods as ElevMap.  The reference keeps one map per process in module-scope variables: one live instance at a time."""

    def __init__(self, length, resolution, mahal=2.0, obstacle=0.6):
        self._l, self.L = ref_lib("elev"), int(length)
        self._l.ref_elev_create(self.L, C.c_float(resolution), C.c_float(mahal), C.c_float(obstacle))

    def move(self, pos3):
        p = _f32(pos3); c = np.zeros(2, np.float32); s = np.zeros(2, np.int32); a = np.zeros(2, np.float32)
        self._l.ref_elev_move(_p(p), _p(c), _p(s), _p(a))
        return c, s, a

    def process_points(self, x, y, z, T, lower, upper, min_r, beam_a, beam_c, sj, rv, csb, pmul, bskew):
        x, y, z = _f32(x).copy(), _f32(y).copy(), _f32(z).copy()
        n = x.size
        mi = np.empty(n, np.int32)
        var, xt, yt, zt = (np.empty(n, np.float32) for _ in range(4))
        a = [_f32(T).reshape(16), _f32(sj).reshape(3), _f32(rv).reshape(9), _f32(csb).reshape(9), _f32(pmul).reshape(3), _f32(bskew).reshape(9)]
        self._l.ref_elev_process_points(n, _p(x), _p(y), _p(z), _p(a[0]), C.c_double(lower), C.c_double(upper), C.c_float(min_r),
                                        C.c_float(beam_a), C.c_float(beam_c), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]),
                                        _p(mi), _p(var), _p(xt), _p(yt), _p(zt))
        return dict(map_index=mi, x=x, y=y, z=z, var=var, x_ts=xt, y_ts=yt, z_ts=zt)

    def fuse(self, index, cr, cg, cb, inten, h, v):
        arrs = [np.ascontiguousarray(index, np.int32), np.ascontiguousarray(cr, np.int32), np.ascontiguousarray(cg, np.int32),
                np.ascontiguousarray(cb, np.int32), _f32(inten), _f32(h), _f32(v)]
        self._l.ref_elev_fuse(arrs[0].size, *[_p(a) for a in arrs])

    def mapvar_update(self, v):
        self._l.ref_elev_mapvar_update(C.c_float(v))

    def map_feature(self):
        n = self.L * self.L
        f = {k: np.zeros(n, np.float32) for k in ("elevation", "var", "rough", "slope", "traver", "intensity")}
        f["traver"][:] = -10
        c = {k: np.zeros(n, np.int32) for k in ("colorR", "colorG", "colorB")}
        self._l.ref_elev_map_feature(_p(f["elevation"]), _p(f["var"]), _p(c["colorR"]), _p(c["colorG"]), _p(c["colorB"]),
                                     _p(f["rough"]), _p(f["slope"]), _p(f["traver"]), _p(f["intensity"]))
        f.update(c)
        return f

    def raytracing(self):
        self._l.ref_elev_raytracing()

    def map_optmove(self, p, dh):
        a = np.zeros(2, np.float32)
        self._l.ref_elev_map_optmove(_p(_f32(p)), C.c_float(dh), _p(a))
        return a

    def map_closeloop(self, p, dh):
        self._l.ref_elev_map_closeloop(_p(_f32(p)), C.c_float(dh))

    def layer(self, which):
        out = np.empty(self.L * self.L, np.float32)
        self._l.ref_elev_get(int(which), _p(out))
        return out

    def frame(self):
        c = np.zeros(2, np.float32); s = np.zeros(2, np.int32)
        self._l.ref_elev_get_frame(_p(c), _p(s))
        return c, s


def ref_kdtree_knn(db, query, k):
    """The reference's own kd-tree (Mapping/src/global_manager/src/kdtree.cpp built in place): k nearest rows of db [n][dim]
    to query [dim] -> (index int64 [m], distance float32 [m] = Euclidean distance, nearest first), m <= k."""
    L = ref_lib("kdtree")
    db, query = _f32(db), _f32(query)
    idx = np.zeros(k, np.int64); dist = np.zeros(k, np.float32)
    L.ref_kdtree_knn.restype = C.c_int
    m = L.ref_kdtree_knn(_p(db), int(db.shape[0]), int(db.shape[1]), _p(query), int(k), _p(idx), _p(dist))
    m = min(int(m), k)
    return idx[:m], dist[:m]


def ref_calc_rel_ori(a, b):
    """GlobalManager::calcRelOri as the reference wrote it (global_manager.cpp:2719-2762 cut out by oracle/Makefile and built with
    stand-ins for FFTW and Eigen::VectorXf -> oracle/_ref/libref_relori.so): a, b complex [height][width] -> degrees."""
    L = ref_lib("relori")
    a, b = np.asarray(a), np.asarray(b)
    h, w = a.shape
    ra, ia, rb, ib = (_f32(x).reshape(-1) for x in (a.real, a.imag, b.real, b.imag))
    L.ref_calc_rel_ori.restype = C.c_float
    return float(L.ref_calc_rel_ori(_p(ra), _p(ia), _p(rb), _p(ib), int(h), int(w)))


def ref_radon_parallel(img, angles, det, spacing=1.0, weight_bits=8):
    """The reference's own Radon kernel (torch-radon/src/forward.cu:12-124, cut out at build time and run on the host with the
    texture fetch the CUDA guide documents: oracle/_ref/libref_radon.so).  weight_bits = 8: the texture unit's 1.8 fixed-point
    interpolation weights; 0: fp32 fractions.  img [B,H,W] or [H,W] -> sinogram [B,n_angles,det]."""
    L = ref_lib("radon")
    img = _f32(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[None]
    B, H, W = img.shape
    ang = _f32(angles)
    out = np.empty((B, ang.size, det), np.float32)
    L.ref_radon_set_weight_bits(int(weight_bits))
    L.ref_radon_parallel(_p(img), B, H, W, _p(ang), ang.size, det, C.c_float(spacing), _p(out))
    return out[0] if squeeze else out
