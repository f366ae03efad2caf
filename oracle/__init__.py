"""CPU oracles (test infrastructure only; never imported by the shipped package)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_C_LIB = None


def build_c(force=False):
    """Compile oracle/raster_c.c with gcc into oracle/_build/libisg_oracle.so."""
    out = os.path.join(_HERE, "_build", "libisg_oracle.so")
    src = os.path.join(_HERE, "raster_c.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return out


def load_c():
    """ctypes handle to the C oracle (built on demand when a compiler is present)."""
    global _C_LIB
    if _C_LIB is None:
        lib = ctypes.CDLL(build_c())
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.isg_oracle_shape_mask.restype = ctypes.c_int
        lib.isg_oracle_shape_mask.argtypes = [i32p, ctypes.c_int, ctypes.c_int, u8p]
        lib.isg_oracle_raster_iou.restype = ctypes.c_int
        lib.isg_oracle_raster_iou.argtypes = [i32p, i32p, i32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              i32p, i32p, u8p, u8p, ctypes.c_int]
        _C_LIB = lib
    return _C_LIB
