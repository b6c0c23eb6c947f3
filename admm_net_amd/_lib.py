"""ctypes binding of libadmmnet_hip.so (the C ABI of include/admmnet.h).

This is the stub a maintainer of the reference would add next to admm_net.py
(see INTEGRATION.md).  There is NO fallback: if the shared library is missing
or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadmmnet_hip.so")


class Cfg(ctypes.Structure):
    """struct admmnet_cfg (include/admmnet.h)."""
    _fields_ = [("M", c_int32), ("N", c_int32), ("L", c_int32), ("K", c_int32),
                ("has_head", c_int32), ("chunk", c_int32), ("reserved", c_int32 * 2)]


class AdmmNetError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol include/admmnet.h declares
SYMBOLS = {
    "admmnet_abi_version": (c_int32, []),
    "admmnet_last_error": (c_char_p, []),
    "admmnet_raw_weight_count": (c_int64, [POINTER(Cfg)]),
    "admmnet_packed_weight_count": (c_int64, [POINTER(Cfg)]),
    "admmnet_pack_weights": (c_int32, [POINTER(Cfg), c_void_p, c_void_p]),
    "admmnet_workspace_bytes": (c_int64, [POINTER(Cfg), c_int64]),
    "admmnet_forward_f32": (c_int32, [POINTER(Cfg), c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "admmnet_begin": (c_int32, [POINTER(Cfg), c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "admmnet_layer_front": (c_int32, [POINTER(Cfg), c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "admmnet_layer_back": (c_int32, [POINTER(Cfg), c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]),
    "admmnet_finish": (c_int32, [POINTER(Cfg), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "admmnet_layer_weight_offset": (c_int64, [POINTER(Cfg), c_int32]),
    "admmnet_glayer_workspace_bytes": (c_int64, [POINTER(Cfg), c_int64]),
    "admmnet_glayer_f32": (c_int32, [POINTER(Cfg), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "admmnet_eigh_workspace_bytes": (c_int64, [c_int32, c_int64]),
    "admmnet_eigh_c64": (c_int32, [c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_void_p, c_void_p]),
    "admmnet_vdvh_c64": (c_int32, [c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "admmnet_vhsv_f32": (c_int32, [c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "admmnet_spectrum_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "admmnet_spectrum_f64": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p,
                                       c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "admmnet_peak_search_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int64]),
    "admmnet_peak_search_f64": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                          c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64,
                                          c_void_p]),
    "admmnet_regional_maxima_f64": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "admmnet_synth_batch": (c_int32, [c_int64, c_int32, c_int32, c_int32, ctypes.c_uint64, c_double, c_double, c_double,
                                      c_double, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "admmnet_layer_back_pair": (c_int32, [POINTER(Cfg), c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]),
    "admmnet_profile_enable": (c_int32, [c_int32]),
    "admmnet_profile_read": (c_int32, [c_void_p, c_void_p, c_int32]),
    "admmnet_profile_dropped": (c_int64, []),
}


def load():
    """Load the HIP extension; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdmmNetError(
            f"{LIB_PATH} is missing: build it with `python -m admm_net_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.admmnet_abi_version() != 1:
        raise AdmmNetError("ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().admmnet_last_error()
        raise AdmmNetError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
