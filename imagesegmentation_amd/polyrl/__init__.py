"""Polygon RL environments (reference polyrl/polygon_env.py, polyrl/polygon_env_end_token.py)."""
