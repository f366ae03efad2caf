"""``PolygonEnv`` whose episode ends when the first true vertex is predicted again
(reference polyrl/polygon_env.py:4-53)."""
from .envs import ScalarPolygonEnv


class PolygonEnv(ScalarPolygonEnv):
    def __init__(self, dataset):
        super().__init__(dataset, end_token=False)
