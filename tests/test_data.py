"""Generator and batch builder against the reference's recorded outputs and batch contract."""
import numpy as np

from imagesegmentation_amd import data
from oracle import raster as oracle_raster


def test_generator_reproduces_reference_polygons(env_golden):
    for key, rec in env_golden["generator"].items():
        size, seed = [int(s) for s in key.split("_")]
        s = size / 28.0
        np.random.seed(seed)
        for want_v, want_area in zip(rec["verts"], rec["areas"]):
            v, image = data.create_valid_polygon(size, 10, 10, 5 * s, 100 * s, 20,
                                                 shape_mask_fn=oracle_raster.create_shape_mask)
            assert np.round(v * size).astype(int).tolist() == want_v
            assert int(np.count_nonzero(image)) // 3 == want_area


def test_inline_asserts_of_the_reference_generator():
    # reference generate_polygons.py:59-61
    lst = [1, 2, 3, 4, 5, 6, 7]
    assert len(list(data.iterate_in_ntuples(lst, n=3, offset=-1))) == len(lst)
    assert abs(data.get_angle(np.array([0, 0]), np.array([1, 1]), np.array([2, 2])) - 180) < 0.01


def _dataset(end_token, hist=2, T=10, P=28, n=12):
    return data.PolygonDataset.synthetic(n, P, P, T, hist, seed=3, end_token=end_token,
                                         shape_mask_fn=oracle_raster.create_shape_mask)


def test_batch_layout_matches_reference_contract():
    for end_token in (False, True):
        ds = _dataset(end_token)
        np.random.seed(5)
        d, images, h, t, verts = ds.get_batch_for_rnn(batch_size=6)
        assert d.dtype == np.int32 and images.shape == (6, 28, 28, 3) and h.shape == (6, 10, 28, 28, 2)
        assert t.shape == (6, 10, 2) and t.dtype == np.int32
        for b in range(6):
            v = verts[b]
            n = len(v)
            assert d[b] == n - 2 + 1
            for step in range(d[b]):
                frames = oracle_raster.create_history(v, step, 2, 28)
                assert np.array_equal(h[b, step], frames)
                expect = v[(step + 1) % n] if not (end_token and step == d[b] - 1) else [28, 0]
                assert t[b, step].tolist() == list(expect)
            assert not h[b, d[b]:].any() and not t[b, d[b]:].any()
        # the point form renders to the same frames
        np.random.seed(5)
        d2, images2, pts, t2, verts2 = ds.get_batch_points(batch_size=6)
        assert np.array_equal(d, d2) and np.array_equal(t, t2) and np.array_equal(images, images2)
        dense = np.zeros_like(h)
        for b in range(6):
            for step in range(10):
                for i in range(2):
                    x, y = pts[b, step, i]
                    if x >= 0:
                        dense[b, step, y, x, i] = 1
        assert np.array_equal(dense, h)


def test_raw_sample_rolls_vertices():
    ds = _dataset(False)
    np.random.seed(1)
    images, verts = ds.raw_sample(4)
    assert images.shape == (4, 28, 28, 3) and len(verts) == 4
    for img, v in zip(images, verts):
        assert v.shape[1] == 2 and img.shape == (28, 28, 3)
        # the image is the fill of the same polygon (28-grid vertices are exact at image_size 28)
        assert np.array_equal(img[:, :, 0], oracle_raster.create_shape_mask(v, 28))
