"""Scalar polygon environments with the reference's reset/step contract.

Both reference variants share everything but the termination rule, so they are one class here with
the rule as a constructor flag; ``polygon_env.PolygonEnv`` and ``polygon_env_end_token.PolygonEnv``
bind the flag.  The terminal reward (mask IoU) is computed on the GPU by ``util.calculate_iou``.

Contract kept from the reference (polyrl/polygon_env.py:8-53, polyrl/polygon_env_end_token.py:8-54):
  * ``dataset`` duck type: ``raw_sample(batch_size=1) -> (images, vertices)``, ``_history_length``,
    ``_prediction_size``;
  * ``reset() -> float64 [P, P, 3 + hist]``: image channels then history frames, oldest first;
  * ``step([x, y]) -> (state, reward, done)``; stepping a finished episode raises
    ``NotImplementedError("Didn't reset a finished environment!")``;
  * public attributes ``image, vertices, predicted_vertices, histories, done``;
  * the history shift is ``np.roll(+1)`` on the frame axis followed by overwriting the last frame,
    reproduced literally (it is a true shift only for history lengths <= 2).
"""
import numpy as np

from .. import util


class ScalarPolygonEnv:
    def __init__(self, dataset, end_token):
        self.dataset = dataset
        self._end_token = bool(end_token)

    # -- helpers ---------------------------------------------------------------------------------
    def _make_state(self):
        return np.concatenate([self.image, self.histories], axis=2)

    def _push_history(self, action):
        size = self.dataset._prediction_size
        self.histories = np.roll(self.histories, axis=2, shift=1)
        self.histories[:, :, -1] = util.create_point_mask(action, size)

    def _calculate_iou(self):
        return util.calculate_iou(np.array(self.predicted_vertices), self.vertices, self.dataset._prediction_size)

    # -- reference surface -----------------------------------------------------------------------
    def reset(self):
        hist = self.dataset._history_length
        self.done = False
        images, polygons = self.dataset.raw_sample(batch_size=1)
        self.image, self.vertices = images[0], polygons[0]
        self.predicted_vertices = self.vertices[:hist].tolist()
        assert len(self.predicted_vertices) == hist
        self.histories = util.create_history(vertices=self.predicted_vertices, end_idx=hist - 1,
                                             history_length=hist, image_size=self.dataset._prediction_size)
        return self._make_state()

    def step(self, action):
        if self.done:
            raise NotImplementedError("Didn't reset a finished environment!")

        if self._end_token:
            if np.all(action == [self.dataset._prediction_size, 0]):
                # closing token: the polygon is closed with the first true vertex, history untouched
                self.predicted_vertices.append(self.vertices[0])
                self.done = True
                return self._make_state(), self._calculate_iou(), True
            self.predicted_vertices.append(action)
            self._push_history(action)
            return self._make_state(), 0, False

        self.predicted_vertices.append(action)
        self._push_history(action)
        if np.all(action == self.vertices[0]):
            self.done = True
            return self._make_state(), self._calculate_iou(), True
        return self._make_state(), 0, False
