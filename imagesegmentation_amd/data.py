"""Synthetic polygons and teacher-forcing batches: the reference's recipe, random-draw order and batch
contract, built around index arithmetic that runs the same way on the host (NumPy) and on the GPU.

What is a contract here (checked against outputs recorded from the reference, tests/test_data.py):

* generator (reference generate_polygons.py:11-54): ONE ``np.random.randint(0, S, [10, 2])`` per attempt ->
  convex hull -> near-straight vertices removed -> filled mask x 3 channels -> accepted when the area (counted
  over the three channels) and the shortest / longest edge pass.  Equal seeds give equal polygons.  The
  vertex removal applies each pass's deletions BY POSITION to an array that shrinks while the positions stay
  those of the pass's start (the reference deletes from a rebound copy while it enumerates the original); a
  position that has fallen off the end is an ``IndexError``, which like any failure of an attempt means
  "draw again".  The lossy JPEG round trip of the reference's files (generate_polygons.py:75, Dataset.py:68)
  is skipped: images stay clean {0, 1}.
* batches (reference Dataset.py:104-162, EndTokenDataset.py:104-164): ``np.random.choice`` of the samples,
  then one ``np.random.randint(n)`` start vertex per sample in batch order; tuple ``(durations, images,
  histories, targets, vertices)``; ``duration = n - hist + 1``; frame i of step t shows rolled vertex
  ``t - hist + i + 1``; target of step t is rolled vertex ``t + 1``; with an end token the target of the
  last filled step is ``(P, 0)`` (EndTokenDataset.py:150).

``PolygonDataset`` keeps every polygon in one padded vertex table, so a batch is a gather:
``sequence_points`` (host) and ``DeviceBatchBuilder`` (isg_batch_points, one launch) evaluate the same
``(t, i, start) -> vertex`` formula.  Dense one-hot frames [B, T, P, P, hist] - the reference's feed - are
rendered from the points only when asked for (``get_batch_for_rnn``).
"""
import numpy as np
import scipy.spatial

from .util import iterate_in_ntuples  # noqa: F401  (part of this module's surface, as in the reference's `from util import *`)


# ---- generator -------------------------------------------------------------------------------------------------
def corner_angles(ring):
    """Angle in degrees at every vertex of a closed ring [V, 2], between its two neighbours."""
    ring = np.asarray(ring)
    to_prev = np.roll(ring, 1, axis=0) - ring
    to_next = np.roll(ring, -1, axis=0) - ring
    dots = (to_prev * to_next).sum(axis=1)
    norms = np.sqrt((to_prev * to_prev).sum(axis=1)) * np.sqrt((to_next * to_next).sum(axis=1))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.degrees(np.arccos(dots / norms))


def get_angle(a, b, c):
    """Angle a-b-c in degrees (reference generate_polygons.py:33-41)."""
    return float(corner_angles(np.stack([np.asarray(a), np.asarray(b), np.asarray(c)]))[1])


def reduce_polygon(verts, eps):
    """Removes vertices whose corner is within ``eps`` degrees of straight, pass after pass until none is
    (reference generate_polygons.py:44-54), with the reference's by-position deletion (module docstring)."""
    ring = np.asarray(verts)
    while True:
        straight = np.flatnonzero(np.abs(180 - corner_angles(ring)) < eps)
        if straight.size == 0:
            return ring
        survivors = list(range(len(ring)))
        for position in straight:
            if position >= len(survivors):
                raise IndexError("index %d is out of bounds for axis 0 with size %d" % (position, len(survivors)))
            del survivors[position]
        ring = ring[survivors]


def _attempt(image_size, shape_complexity, reduction_tolerance, shape_mask_fn):
    """One draw of the generator: (ring [V, 2] int, mask [S, S]); raises when the draw is degenerate."""
    cloud = np.random.randint(0, image_size, [shape_complexity, 2])
    ring = cloud[scipy.spatial.ConvexHull(cloud).vertices]
    ring = reduce_polygon(ring, eps=reduction_tolerance)
    return ring, shape_mask_fn(ring, image_size)


def create_valid_polygon(image_size, shape_complexity, min_area, min_distance, max_distance, reduction_tolerance,
                         shape_mask_fn=None):
    """(vertices / image_size, image [S, S, 3]) of the first accepted draw; consumes ``np.random`` exactly as
    reference generate_polygons.py:11-30 does."""
    if shape_mask_fn is None:
        from .util import create_shape_mask as shape_mask_fn
    while True:
        try:
            ring, mask = _attempt(image_size, shape_complexity, reduction_tolerance, shape_mask_fn)
        except Exception:  # degenerate hull, a vertex dropped off the ring, a pixel outside the image: draw again
            continue
        edges = np.roll(ring, -1, axis=0) - ring
        lengths = np.sqrt((edges * edges).sum(axis=1))
        area = 3 * int(np.count_nonzero(mask))  # the reference counts over the 3-channel image
        if area > min_area and lengths.min() > min_distance and lengths.max() < max_distance:
            return ring / image_size, np.repeat(mask[:, :, None], 3, axis=2)


def generate_polygons(count, image_size, seed, shape_mask_fn=None):
    """``count`` polygons with the reference's settings (generate_polygons.py:72-74); the edge-length
    thresholds 5 / 100 scale with image_size / 28.  Returns (images [N,S,S,3] float64, list of
    normalised vertex arrays)."""
    np.random.seed(seed)
    unit = image_size / 28.0
    drawn = [create_valid_polygon(image_size, 10, 10, 5 * unit, 100 * unit, 20, shape_mask_fn) for _ in range(count)]
    return np.stack([image for _, image in drawn]).astype(np.float64), [ring for ring, _ in drawn]


# ---- batches ----------------------------------------------------------------------------------------------------
def sequence_points(table, counts, indices, starts, max_timesteps, history_length, prediction_size, end_token):
    """The teacher-forcing sequences of samples ``indices`` rolled to start vertices ``starts``, as a gather from
    the padded vertex ``table`` int [N, vcap, 2] (``counts`` [N] vertices each).

    np.roll(v, s) puts v[(k - s) mod n] at position k, so rolled vertex k of a sample is table row
    ``(k - s) mod n``.  Returns (durations int32 [B], points int32 [B, T, hist, 2] with (-1, -1) for frames of
    steps that are not filled, targets int32 [B, T, 2])."""
    indices = np.asarray(indices, np.int64)
    T, hist = int(max_timesteps), int(history_length)
    n = counts[indices].astype(np.int64)[:, None, None]
    s = np.asarray(starts, np.int64)[:, None, None]
    t = np.arange(T)[None, :, None]
    i = np.arange(hist)[None, None, :]
    durations = (n - hist + 1)[:, 0, 0]
    filled = t < np.minimum(durations, T)[:, None, None]  # steps the reference's loop writes
    row = indices[:, None, None]
    points = table[row, ((t - hist + i + 1) % n - s) % n].astype(np.int32)
    points[~np.broadcast_to(filled, points.shape[:3])] = -1
    targets = table[row[:, :, 0], (((t + 1) % n - s) % n)[:, :, 0]].astype(np.int32)
    targets[~filled[:, :, 0]] = 0
    if end_token:
        last = np.maximum(np.minimum(durations, T) - 1, 0)
        targets[np.arange(len(indices)), last] = (prediction_size, 0)
    return durations.astype(np.int32), points, targets


def frames_from_points(points, prediction_size, dtype=np.float64):
    """Dense one-hot frames [B, T, P, P, hist] (index [y, x]) of int points [B, T, hist, 2]."""
    B, T, hist, _ = points.shape
    frames = np.zeros([B, T, prediction_size, prediction_size, hist], dtype=dtype)
    b, t, i = np.nonzero(points[..., 0] >= 0)
    frames[b, t, points[b, t, i, 1], points[b, t, i, 0], i] = 1
    return frames


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
        self._counts = np.array([len(v) for v in vertices], np.int32)
        self._table = np.zeros([len(vertices), max(int(self._counts.max(initial=1)), 1), 2], np.int32)
        for k, v in enumerate(vertices):
            self._table[k, :len(v)] = v

    @classmethod
    def synthetic(cls, count, image_size, prediction_size, max_timesteps, history_length, seed=0, end_token=False,
                  shape_mask_fn=None):
        images, rings = generate_polygons(count, image_size, seed, shape_mask_fn)
        vertices = np.empty([count], dtype=object)
        for k, ring in enumerate(rings):
            vertices[k] = np.floor(ring * prediction_size).astype(int)
        return cls(images, vertices, max_timesteps, prediction_size, history_length, end_token)

    def __len__(self):
        assert self._images.shape[0] == self._vertices.shape[0]
        return self._images.shape[0]

    @property
    def image_size(self):
        return self._image_size

    # The random part of a batch, in the reference's order: which samples, then each one's start vertex.
    def draw(self, batch_size, start_idx=None):
        indices = np.random.choice(self._images.shape[0], batch_size, replace=False)
        if start_idx is None:
            starts = np.array([np.random.randint(self._counts[k]) for k in indices], np.int64).reshape(-1)
        else:
            starts = np.full([batch_size], start_idx, np.int64)
        return indices, starts

    def rolled_vertices(self, indices, starts):
        out = np.empty([len(indices)], dtype=object)
        for k, (sample, start) in enumerate(zip(indices, starts)):
            out[k] = np.roll(self._vertices[sample], int(start), axis=0)
        return out

    def get_batch_points(self, batch_size=8, start_idx=None):
        """(durations [B], images [B,S,S,3], points int32 [B,T,hist,2] with (-1,-1) = empty frame, targets
        [B,T,2], vertices): the reference's tuple with the history as vertex points."""
        indices, starts = self.draw(batch_size, start_idx)
        durations, points, targets = sequence_points(self._table, self._counts, indices, starts, self._max_timesteps,
                                                     self._history_length, self._prediction_size, self._end_token)
        return durations, self._images[indices], points, targets, self.rolled_vertices(indices, starts)

    def get_batch_for_rnn(self, batch_size=8, start_idx=None):
        """(durations [B], images [B,S,S,3], histories [B,T,P,P,hist], targets [B,T,2], vertices)."""
        durations, images, points, targets, vertices = self.get_batch_points(batch_size, start_idx)
        return durations, images, frames_from_points(points, self._prediction_size), targets, vertices

    def raw_sample(self, batch_size):
        """(images, vertices) with a random start vertex per polygon (Dataset.py:152-162)."""
        indices, starts = self.draw(batch_size)
        return self._images[indices], self.rolled_vertices(indices, starts)


class DeviceBatchBuilder:
    """The batch builder with the dataset resident on the GPU: images, the padded vertex table and the counts are
    uploaded once; a batch is one gather of the images plus ONE isg_batch_points launch that writes durations,
    history points, targets and the rolled polygons.  Nothing of size B x T x P x P is ever built, and the only
    host work per batch is the random draw (the reference's ``np.random`` calls, so that equal seeds give the
    batches ``PolygonDataset.get_batch_points`` gives) or none at all with ``device_draw=True``."""

    def __init__(self, dataset, device, image_dtype=None):
        import torch
        from . import ops
        self.dataset = dataset
        self.device = torch.device(device)
        images = torch.as_tensor(dataset._images)
        self.images = images.to(self.device, dtype=image_dtype if image_dtype is not None else images.dtype)
        if len(dataset._counts) == 0 or int(dataset._counts.min()) < 1:
            raise ValueError("every polygon of the dataset needs at least one vertex")
        self.table = ops.upload(dataset._table, self.device, torch.int32).contiguous()
        self.counts = ops.upload(dataset._counts, self.device, torch.int32).contiguous()
        self._generator = None

    def build(self, indices, starts):
        """indices / starts: host arrays or device int tensors [B]."""
        import torch
        from . import ops
        ds = self.dataset
        indices = ops.upload(np.asarray(indices), self.device, torch.int32) if not torch.is_tensor(indices) \
            else indices.to(self.device, torch.int32)
        starts = ops.upload(np.asarray(starts), self.device, torch.int32) if not torch.is_tensor(starts) \
            else starts.to(self.device, torch.int32)
        durations, points, targets, rolled, nverts = ops.batch_points(
            self.table, self.counts, indices, starts, ds._max_timesteps, ds._history_length, ds._prediction_size,
            ds._end_token)
        images = self.images.index_select(0, indices.long())
        return dict(durations=durations, images=images, points=points, targets=targets, vertices=rolled,
                    nverts=nverts)

    def get_batch_points(self, batch_size=8, start_idx=None, device_draw=False, seed=0):
        """``device_draw``: samples and start vertices are drawn on the GPU (its own generator, seeded once with
        ``seed``): no host work per batch, not the reference's random stream."""
        if not device_draw:
            return self.build(*self.dataset.draw(batch_size, start_idx))
        import torch
        if self._generator is None:
            self._generator = torch.Generator(device=self.device)
            self._generator.manual_seed(int(seed))
        n = self.images.shape[0]
        indices = torch.randperm(n, device=self.device, generator=self._generator)[:batch_size].to(torch.int32)
        u = torch.rand([batch_size], device=self.device, generator=self._generator)
        counts = self.counts.index_select(0, indices.long())
        starts = torch.minimum((u * counts).to(torch.int32), counts - 1) if start_idx is None else \
            torch.full([batch_size], int(start_idx), dtype=torch.int32, device=self.device)
        return self.build(indices, starts)
