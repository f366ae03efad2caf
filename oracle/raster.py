"""CPU oracle for the polygon rasteriser and the IoU reward.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package imports this module: it exists so that tests, the
smoke check and the CPU-baseline leg of the benchmark have an independent checker for
the HIP rasteriser.

What it restates
----------------
* ``util.create_shape_mask``                       (reference util.py:60-65)
* ``PolygonEnv._calculate_iou``                    (reference polyrl/polygon_env.py:48-53,
                                                    polyrl/polygon_env_end_token.py:49-54)
* ``util.create_point_mask`` / ``create_history``  (reference util.py:44-57)

The arithmetic of ``create_shape_mask`` lives in a third-party dependency that the
reference does not vendor or pin: ``skimage.draw.polygon`` -> Cython ``_draw._polygon``
-> ``_shared.geometry.point_in_polygon``.  The behaviour pinned here is that of
scikit-image 0.18.3 (the version importable next to the reference when the golden
fixtures under tests/golden/ were generated): a pixel is set when its integer centre is
inside by crossing parity, or lies on an edge, or coincides with a vertex; rows are
scanned from max(0, min y) to max y and columns from max(0, min x) to max x.

Pinning: tests/test_oracle_raster.py checks every function here against fixtures
produced by the real reference code (tests/golden/make_raster_fixtures.py).
"""
import numpy as np


def point_in_polygon(xs, ys, x, y):
    """Scalar restatement of skimage's ``point_in_polygon`` for integer inputs.

    Returns 0 outside, 1 inside, 2 vertex, 3 edge (skimage/_shared/geometry.pxd).
    For integer coordinates the floating-point quotient of the original has exactly the
    sign of numerator*denominator, so the sign form below is exact.
    """
    n = len(xs)
    r_cross = 0
    l_cross = 0
    x1 = int(xs[n - 1]) - x
    y1 = int(ys[n - 1]) - y
    for i in range(n):
        x0 = int(xs[i]) - x
        y0 = int(ys[i]) - y
        if x0 == 0 and y0 == 0:
            return 2
        if (y0 > 0) != (y1 > 0):
            num = x0 * y1 - x1 * y0
            den = y1 - y0
            if (num > 0 and den > 0) or (num < 0 and den < 0):
                r_cross += 1
        if (y0 < 0) != (y1 < 0):
            num = x0 * y1 - x1 * y0
            den = y1 - y0
            if (num > 0 and den < 0) or (num < 0 and den > 0):
                l_cross += 1
        x1, y1 = x0, y0
    if (r_cross & 1) != (l_cross & 1):
        return 3
    return 1 if (r_cross & 1) else 0


def create_shape_mask_loops(vertices, image_size):
    """Pure-Python pixel loop; only for small cases (reference util.py:60-65)."""
    vertices = np.asarray(vertices).astype(np.int64).reshape(-1, 2)
    xs, ys = vertices[:, 0], vertices[:, 1]
    mask = np.zeros([image_size, image_size])
    minr, maxr = max(0, int(ys.min())), int(ys.max())
    minc, maxc = max(0, int(xs.min())), int(xs.max())
    for r in range(minr, maxr + 1):
        for c in range(minc, maxc + 1):
            if point_in_polygon(xs, ys, c, r):
                mask[r, c] = 1  # IndexError beyond the image, exactly as util.py:64
    return mask


def create_shape_mask(vertices, image_size):
    """Vectorised integer restatement of ``util.create_shape_mask`` (util.py:60-65).

    vertices: int array [V, 2] of (x, y).  Returns float64 [image_size, image_size].
    Raises IndexError when a covered pixel falls outside the image, as the reference's
    ``mask[rr, cc] = 1`` does.
    """
    vertices = np.asarray(vertices).astype(np.int64).reshape(-1, 2)
    xs, ys = vertices[:, 0], vertices[:, 1]
    mask = np.zeros([image_size, image_size])
    minr, maxr = max(0, int(ys.min())), int(ys.max())
    minc, maxc = max(0, int(xs.min())), int(xs.max())
    if maxr < minr or maxc < minc:
        return mask
    rr = np.arange(minr, maxr + 1)[:, None, None]
    cc = np.arange(minc, maxc + 1)[None, :, None]
    x0 = xs[None, None, :] - cc
    y0 = ys[None, None, :] - rr
    x1 = np.roll(xs, 1)[None, None, :] - cc
    y1 = np.roll(ys, 1)[None, None, :] - rr
    vertex = ((x0 == 0) & (y0 == 0)).any(axis=2)
    s = np.sign(x0 * y1 - x1 * y0) * np.sign(y1 - y0)
    r_cross = (((y0 > 0) != (y1 > 0)) & (s > 0)).sum(axis=2)
    l_cross = (((y0 < 0) != (y1 < 0)) & (s < 0)).sum(axis=2)
    filled = vertex | ((r_cross & 1) != (l_cross & 1)) | ((r_cross & 1) == 1)
    r_idx, c_idx = np.nonzero(filled)
    mask[r_idx + minr, c_idx + minc] = 1
    return mask


def iou_counts(predicted_vertices, true_vertices, size):
    """(intersection, union) pixel counts (polygon_env.py:49-52)."""
    predicted_mask = create_shape_mask(np.array(predicted_vertices), size)
    true_mask = create_shape_mask(true_vertices, size)
    intersection = int(np.count_nonzero(np.logical_and(predicted_mask, true_mask)))
    union = int(np.count_nonzero(np.logical_or(predicted_mask, true_mask)))
    return intersection, union


def calculate_iou(predicted_vertices, true_vertices, size):
    """``PolygonEnv._calculate_iou`` (polygon_env.py:48-53): Python int / int."""
    intersection, union = iou_counts(predicted_vertices, true_vertices, size)
    return intersection / union


def create_point_mask(point, size):
    """reference util.py:54-57"""
    mask = np.zeros([size, size])
    mask[point[1], point[0]] = 1
    return mask


def create_history(vertices, end_idx, history_length, image_size):
    """reference util.py:44-51"""
    num_vertices = len(vertices)
    history_mask = np.zeros([image_size, image_size, history_length])
    for i in range(history_length):
        x, y = vertices[(end_idx - history_length + i + 1) % num_vertices]
        history_mask[y, x, i] = 1
    return history_mask
