"""``PolygonEnv`` whose episode ends on the action ``[prediction_size, 0]``
(reference polyrl/polygon_env_end_token.py:4-54)."""
from .envs import ScalarPolygonEnv


class PolygonEnv(ScalarPolygonEnv):
    def __init__(self, dataset):
        super().__init__(dataset, end_token=True)
