"""The CPU oracles against fixtures produced by the real reference (tests/golden/make_raster_fixtures.py)."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import raster
from conftest import unpack_masks

SIZES = (8, 28, 32, 224)


@pytest.mark.parametrize("size", SIZES)
def test_numpy_oracle_masks_match_reference(raster_golden, size):
    verts, nverts = raster_golden["verts_%d" % size], raster_golden["nverts_%d" % size]
    ref = unpack_masks(raster_golden["masks_%d" % size], size)
    for i in range(len(verts)):
        got = raster.create_shape_mask(verts[i, :nverts[i]], size)
        assert got.dtype == np.float64
        assert np.array_equal(got.astype(np.uint8), ref[i]), "polygon %d at size %d" % (i, size)


@pytest.mark.parametrize("size", (8, 28))
def test_loop_oracle_matches_reference(raster_golden, size):
    verts, nverts = raster_golden["verts_%d" % size], raster_golden["nverts_%d" % size]
    ref = unpack_masks(raster_golden["masks_%d" % size], size)
    for i in range(0, len(verts), 3):
        got = raster.create_shape_mask_loops(verts[i, :nverts[i]], size)
        assert np.array_equal(got.astype(np.uint8), ref[i])


@pytest.mark.parametrize("size", SIZES)
def test_c_oracle_masks_and_iou_match_reference(raster_golden, size):
    lib = oracle.load_c()
    verts = np.ascontiguousarray(raster_golden["verts_%d" % size])
    nverts = np.ascontiguousarray(raster_golden["nverts_%d" % size])
    n, vmax = verts.shape[0], verts.shape[1]
    vt = np.ascontiguousarray(np.roll(verts, -1, axis=0))
    nt = np.ascontiguousarray(np.roll(nverts, -1, axis=0))
    inter = np.zeros([n], np.int32)
    union = np.zeros([n], np.int32)
    mp = np.zeros([n, size, size], np.uint8)
    mt = np.zeros([n, size, size], np.uint8)
    i32p, u8p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8)
    out = lib.isg_oracle_raster_iou(verts.ctypes.data_as(i32p), nverts.ctypes.data_as(i32p),
                                    vt.ctypes.data_as(i32p), nt.ctypes.data_as(i32p), n, vmax, size,
                                    inter.ctypes.data_as(i32p), union.ctypes.data_as(i32p),
                                    mp.ctypes.data_as(u8p), mt.ctypes.data_as(u8p), 2)
    assert out == 0
    ref = unpack_masks(raster_golden["masks_%d" % size], size)
    assert np.array_equal(mp, ref)
    assert np.array_equal(mt, np.roll(ref, -1, axis=0))
    assert np.array_equal(inter, raster_golden["inter_%d" % size])
    assert np.array_equal(union, raster_golden["union_%d" % size])
    assert np.array_equal(inter / union, raster_golden["iou_%d" % size])  # float64, bit for bit


@pytest.mark.parametrize("size", (28, 224))
def test_numpy_oracle_iou_matches_reference(raster_golden, size):
    verts, nverts = raster_golden["verts_%d" % size], raster_golden["nverts_%d" % size]
    n = len(verts)
    for i in range(0, n, 7):
        j = (i + 1) % n
        iou = raster.calculate_iou(verts[i, :nverts[i]], verts[j, :nverts[j]], size)
        assert iou == raster_golden["iou_%d" % size][i]


def test_out_of_image_raises_like_reference(env_golden):
    assert env_golden["out_of_image_raises_IndexError"] == [True, True]
    for p in ([[0, 0], [8, 0], [8, 5]], [[3, 3], [3, 9], [5, 5]]):
        with pytest.raises(IndexError):
            raster.create_shape_mask(np.array(p), 8)
        with pytest.raises(IndexError):
            raster.create_shape_mask_loops(np.array(p), 8)


def test_known_answers():
    # boundary-inclusive rectangle (skimage test_polygon_rectangle: corners are filled)
    m = raster.create_shape_mask(np.array([[1, 1], [6, 1], [6, 5], [1, 5]]), 8)
    expect = np.zeros([8, 8])
    expect[1:6, 1:7] = 1
    assert np.array_equal(m, expect)
    # skimage docstring triangle (draw.py:480-496): r = [1, 2, 8], c = [1, 7, 4] on a 10x10 canvas
    tri = raster.create_shape_mask(np.array([[1, 1], [7, 2], [4, 8]]), 10)
    rows = ["0000000000", "0100000000", "0011111100", "0011111000", "0001111000", "0001110000", "0000110000",
            "0000100000", "0000100000", "0000000000"]
    assert np.array_equal(tri, np.array([[int(ch) for ch in r] for r in rows], dtype=np.float64))
