"""Elevation mapping over the C ABI (row N3): host-side mirror of the reference's libgpu.so functions
(Mapping/src/elevation_mapping_periodical/elevation_mapping/cuda/gpu_process.cu:938-1312)."""
import ctypes as C

import numpy as np

from . import _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class ElevationMap:
    def __init__(self, length, resolution, mahalanobis_threshold=2.0, obstacle_threshold=0.6, device=0):
        self.L = int(length)
        self._h = C.c_void_p()
        _lib.check(_lib.load().mrs_elev_create(_lib.ctx(device), self.L, C.c_float(resolution), C.c_float(mahalanobis_threshold),
                                               C.c_float(obstacle_threshold), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().mrs_elev_destroy(self._h)
        except Exception:
            pass

    def move(self, position3):
        p = _f(position3); c = np.zeros(2, np.float32); s = np.zeros(2, np.int32); a = np.zeros(2, np.float32)
        _lib.check(_lib.load().mrs_elev_move(self._h, _lib.ptr(p), _lib.ptr(c), _lib.ptr(s), _lib.ptr(a)))
        return c, s, a

    def process_points(self, x, y, z, transform, lower, upper, min_r, beam_a, beam_c, sensor_jacobian, rotation_variance,
                       c_sb_transpose, p_mul_c_bm_transpose, b_r_bs_skew):
        x, y, z = _f(x).copy(), _f(y).copy(), _f(z).copy()
        n = x.size
        mi = np.empty(n, np.int32)
        var, xt, yt, zt = (np.empty(n, np.float32) for _ in range(4))
        T = _f(transform).reshape(16)
        args = [_f(sensor_jacobian).reshape(3), _f(rotation_variance).reshape(9), _f(c_sb_transpose).reshape(9),
                _f(p_mul_c_bm_transpose).reshape(3), _f(b_r_bs_skew).reshape(9)]
        _lib.check(_lib.load().mrs_elev_process_points(self._h, n, _lib.ptr(x), _lib.ptr(y), _lib.ptr(z), _lib.ptr(T),
                                                       C.c_double(lower), C.c_double(upper), C.c_float(min_r), C.c_float(beam_a),
                                                       C.c_float(beam_c), *[_lib.ptr(a) for a in args], _lib.ptr(mi), _lib.ptr(var),
                                                       _lib.ptr(xt), _lib.ptr(yt), _lib.ptr(zt)))
        return dict(map_index=mi, x=x, y=y, z=z, var=var, x_ts=xt, y_ts=yt, z_ts=zt)

    def fuse(self, index, color_r, color_g, color_b, intensity, height, var):
        arrs = [_i(index), _i(color_r), _i(color_g), _i(color_b), _f(intensity), _f(height), _f(var)]
        _lib.check(_lib.load().mrs_elev_fuse(self._h, arrs[0].size, *[_lib.ptr(a) for a in arrs]))

    def mapvar_update(self, v):
        _lib.check(_lib.load().mrs_elev_mapvar_update(self._h, C.c_float(v)))

    def map_feature(self):
        n = self.L * self.L
        f = {k: np.empty(n, np.float32) for k in ("elevation", "var", "rough", "slope", "traver", "intensity")}
        c = {k: np.empty(n, np.int32) for k in ("colorR", "colorG", "colorB")}
        _lib.check(_lib.load().mrs_elev_map_feature(self._h, _lib.ptr(f["elevation"]), _lib.ptr(f["var"]), _lib.ptr(c["colorR"]),
                                                    _lib.ptr(c["colorG"]), _lib.ptr(c["colorB"]), _lib.ptr(f["rough"]),
                                                    _lib.ptr(f["slope"]), _lib.ptr(f["traver"]), _lib.ptr(f["intensity"])))
        f.update(c)
        return f

    def raytracing(self):
        _lib.check(_lib.load().mrs_elev_raytracing(self._h))

    def map_optmove(self, opt_p, height_update):
        a = np.zeros(2, np.float32)
        _lib.check(_lib.load().mrs_elev_map_optmove(self._h, _lib.ptr(_f(opt_p)), C.c_float(height_update), _lib.ptr(a)))
        return a

    def map_closeloop(self, update_position, height_update):
        _lib.check(_lib.load().mrs_elev_map_closeloop(self._h, _lib.ptr(_f(update_position)), C.c_float(height_update)))

    def layer(self, which):
        out = np.empty(self.L * self.L, np.float32)
        _lib.check(_lib.load().mrs_elev_get_layer(self._h, int(which), _lib.ptr(out)))
        return out

    def frame(self):
        c = np.zeros(2, np.float32); s = np.zeros(2, np.int32)
        _lib.check(_lib.load().mrs_elev_get_frame(self._h, _lib.ptr(c), _lib.ptr(s)))
        return c, s
