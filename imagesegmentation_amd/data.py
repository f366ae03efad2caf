"""Synthetic polygons and teacher-forcing batches with the reference's recipe and batch layout.

* ``create_valid_polygon`` / ``reduce_polygon`` / ``get_angle``: the generator recipe of reference
  generate_polygons.py:11-54 (random points -> convex hull -> drop near-straight vertices -> fill,
  rejection on area and edge lengths), including its quirks: the area is counted over the
  3-channel image, and ``reduce_polygon`` keeps iterating over the vertex list it started a pass with
  while deleting from the shrinking copy.  The lossy JPEG round trip of the reference's dataset
  (generate_polygons.py:75, Dataset.py:68) is skipped: images stay clean {0, 1}.
* ``PolygonDataset``: ``get_batch_for_rnn`` and ``raw_sample`` with the reference's tuple layout
  (reference Dataset.py:104-162, EndTokenDataset.py:104-164), plus ``get_batch_points`` which hands
  the history out as vertex points [B, T, hist, 2] instead of dense one-hot frames, so that the
  frames are rendered on the GPU and a 224 x 224 batch does not move half a gigabyte of float64.
"""
import numpy as np
import scipy.spatial

from .util import create_history, iterate_in_ntuples


def get_angle(a, b, c):
    """Angle a-b-c in degrees (reference generate_polygons.py:33-41)."""
    ba, bc = a - b, c - b
    cosine = np.dot(ba, bc) / (np.linalg.norm(ba) * np.linalg.norm(bc))
    return np.degrees(np.arccos(cosine))


def reduce_polygon(verts, eps):
    """Deletes vertices whose interior angle is within eps of 180 degrees
    (reference generate_polygons.py:44-54).  As there, a pass enumerates the vertex array it started
    with while deletions hit the current, shorter array."""
    again = True
    while again:
        again = False
        for i, (a, b, c) in enumerate(iterate_in_ntuples(verts, n=3, offset=-1)):
            if abs(180 - get_angle(a, b, c)) < eps:
                verts = np.delete(verts, i, 0)
                again = True
    return verts


def create_valid_polygon(image_size, shape_complexity, min_area, min_distance, max_distance, reduction_tolerance,
                         shape_mask_fn=None):
    """(vertices / image_size, image [S, S, 3]) of one accepted polygon; draws from ``np.random``
    exactly as reference generate_polygons.py:11-30 does, so equal seeds give equal polygons."""
    if shape_mask_fn is None:
        from .util import create_shape_mask as shape_mask_fn
    while True:
        try:
            points = np.random.randint(0, image_size, [shape_complexity, 2])
            hull = scipy.spatial.ConvexHull(points)
            poly_verts = np.array([(points[s, 0], points[s, 1]) for s in hull.vertices])
            poly_verts = reduce_polygon(poly_verts, eps=reduction_tolerance)
            image = np.stack([shape_mask_fn(poly_verts, image_size) for _ in range(3)], axis=2)
        except Exception:
            continue
        area = np.count_nonzero(image)  # over 3 channels, as in the reference
        distances = [np.linalg.norm(b - a) for a, b in iterate_in_ntuples(poly_verts, n=2)]
        if area > min_area and min(distances) > min_distance and max(distances) < max_distance:
            break
    return poly_verts / image_size, image


def generate_polygons(count, image_size, seed, shape_mask_fn=None):
    """``count`` polygons with the reference's settings (generate_polygons.py:72-74); the edge-length
    thresholds 5 / 100 scale with image_size / 28.  Returns (images [N,S,S,3] float64, list of
    normalised vertex arrays)."""
    np.random.seed(seed)
    s = image_size / 28.0
    images, verts = [], []
    for _ in range(count):
        v, img = create_valid_polygon(image_size, 10, 10, 5 * s, 100 * s, 20, shape_mask_fn)
        images.append(img)
        verts.append(v)
    return np.stack(images).astype(np.float64), verts


class PolygonDataset:
    """In-memory dataset with the reference's batch contract (Dataset.py:95-170).

    ``vertices``: object array of integer [V, 2] arrays on the prediction grid
    (``floor(normalised * prediction_size)``, Dataset.py:73).  ``end_token=True`` reproduces
    EndTokenDataset: the target of the last valid step is (prediction_size, 0) (its :150)."""

    def __init__(self, images, vertices, max_timesteps, prediction_size, history_length, end_token=False):
        self._images = images
        self._vertices = vertices
        self._max_timesteps = max_timesteps
        self._image_size = self._images.shape[1]
        self._prediction_size = prediction_size
        self._history_length = history_length
        self._end_token = end_token

    @classmethod
    def synthetic(cls, count, image_size, prediction_size, max_timesteps, history_length, seed=0, end_token=False,
                  shape_mask_fn=None):
        images, verts = generate_polygons(count, image_size, seed, shape_mask_fn)
        vs = np.empty([count], dtype=object)
        for i, v in enumerate(verts):
            vs[i] = np.floor(v * prediction_size).astype(int)
        return cls(images, vs, max_timesteps, prediction_size, history_length, end_token)

    def __len__(self):
        assert self._images.shape[0] == self._vertices.shape[0]
        return self._images.shape[0]

    @property
    def image_size(self):
        return self._image_size

    def _sequence(self, poly_verts, start_idx=None, dense=True):
        n = len(poly_verts)
        hist, T, P = self._history_length, self._max_timesteps, self._prediction_size
        start_idx = np.random.randint(n) if start_idx is None else start_idx
        poly_verts = np.roll(poly_verts, start_idx, axis=0)
        histories = np.zeros([T, P, P, hist], dtype=np.uint16) if dense else np.full([T, hist, 2], -1, np.int32)
        targets = np.zeros([T, 2], dtype=np.int32)
        idx = 0
        for idx in range(min(n - hist + 1, T)):
            if dense:
                histories[idx] = create_history(poly_verts, idx, hist, P)
            else:
                for i in range(hist):
                    histories[idx, i] = poly_verts[(idx - hist + i + 1) % n]
            targets[idx, :] = np.array(poly_verts[(idx + 1) % n])
        if self._end_token:
            targets[idx, :] = np.array([P, 0])
        return n - hist + 1, histories, targets, poly_verts

    def _batch(self, batch_size, start_idx, dense):
        hist, T, P = self._history_length, self._max_timesteps, self._prediction_size
        batch_indices = np.random.choice(self._images.shape[0], batch_size, replace=False)
        batch_d = np.zeros([batch_size], dtype=np.int32)
        batch_images = self._images[batch_indices]
        batch_h = np.zeros([batch_size, T, P, P, hist]) if dense else np.zeros([batch_size, T, hist, 2], np.int32)
        batch_t = np.zeros([batch_size, T, 2], dtype=np.int32)
        batch_vertices = self._vertices[batch_indices]
        for i, vertices in enumerate(batch_vertices):
            batch_d[i], batch_h[i], batch_t[i], batch_vertices[i] = self._sequence(vertices, start_idx, dense)
        return batch_d, batch_images, batch_h, batch_t, batch_vertices

    def get_batch_for_rnn(self, batch_size=8, start_idx=None):
        """(durations [B], images [B,S,S,3], histories [B,T,P,P,hist], targets [B,T,2], vertices)."""
        return self._batch(batch_size, start_idx, True)

    def get_batch_points(self, batch_size=8, start_idx=None):
        """Same tuple with histories as int32 points [B,T,hist,2] ((-1,-1) = empty frame)."""
        return self._batch(batch_size, start_idx, False)

    def raw_sample(self, batch_size):
        """(images, vertices) with a random start vertex per polygon (Dataset.py:152-162)."""
        batch_indices = np.random.choice(self._images.shape[0], batch_size, replace=False)
        batch_images = self._images[batch_indices]
        batch_verts = np.empty([batch_size], dtype=object)
        for i, p in enumerate(self._vertices[batch_indices]):
            batch_verts[i] = np.roll(p, np.random.randint(len(p)), axis=0)
        return batch_images, batch_verts
