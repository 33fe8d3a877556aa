"""ctypes binding of libmrslam_hip.so (C ABI: include/mrslam_hip.h).

There is no CPU fallback: if the shared library is missing or no GPU is visible the
product path raises.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmrslam_hip.so")

MRS_OK = 0
OUT_REFERENCE = 0
OUT_COMPACT = 1
DROPPED = -2**31


class MrsError(RuntimeError):
    pass


class BevCfg(C.Structure):
    _fields_ = [("max_length", C.c_int32), ("max_height", C.c_int32), ("n0", C.c_int32),
                ("n1", C.c_int32), ("num_height", C.c_int32), ("enough_large", C.c_int32)]


_lib = None
_lock = threading.Lock()
_ctx = {}


def load():
    """Load the HIP library; raises MrsError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise MrsError(
                        f"{LIB_PATH} not found: build it with `make -C mr_slam_amd/csrc` "
                        "(or __graft_entry__.build()); there is no CPU fallback")
                lib = C.CDLL(LIB_PATH)
                lib.mrs_status_str.restype = C.c_char_p
                lib.mrs_last_error.restype = C.c_char_p
                _lib = lib
    return _lib


def check(status):
    if status != MRS_OK:
        lib = load()
        raise MrsError("%s: %s" % (lib.mrs_status_str(status).decode(), lib.mrs_last_error().decode()))


def ctx(device=0):
    """Per-process, per-device context handle (created on first use)."""
    lib = load()
    with _lock:
        h = _ctx.get(device)
        if h is None:
            h = C.c_void_p()
            check(lib.mrs_ctx_create(int(device), C.byref(h)))
            _ctx[device] = h
    return h


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array as c_void_p."""
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
