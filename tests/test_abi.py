"""The C-ABI library loads and exports exactly what include/isg.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT
from imagesegmentation_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "isg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert "isg_raster_iou" in names and "isg_version" in names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), "libisg.so does not export %s" % name


def test_python_binding_covers_the_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_text():
    lib = _lib.lib()
    assert lib.isg_version() >= 100
    assert isinstance(lib.isg_last_error(), bytes)


def test_argument_validation_happens_before_any_launch():
    lib = _lib.lib()
    # size out of range and null pointers are rejected on the host, no GPU needed
    assert lib.isg_raster_iou(None, None, None, None, 4, 16, 28, None, None, None, None, None) == -1
    assert b"null" in lib.isg_last_error()
    assert lib.isg_shape_mask(None, None, 1, 16, 4096, None, None, None) == -1


def test_upload_accepts_every_integer_and_float_dtype():
    """ops.upload on the CPU device (the pinned path needs a GPU): values and requested dtype survive, including
    the unsigned types torch has no name for."""
    import numpy as np
    import torch
    from imagesegmentation_amd import ops
    for np_dtype in (np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.float32, np.float64, np.bool_):
        a = (np.arange(12).reshape(3, 4) % 5).astype(np_dtype)
        t = ops.upload(a, "cpu", torch.int32)
        assert t.dtype == torch.int32 and np.array_equal(t.numpy(), a.astype(np.int32))
    assert ops.upload(np.zeros([0], np.int32), "cpu").numel() == 0
    assert ops.upload([[1, 2], [3, 4]], "cpu").tolist() == [[1, 2], [3, 4]]
