"""Experiment: can one REINFORCE rollout (T policy steps + env steps) be captured in a HIP graph?"""
import time
import numpy as np
import torch
from imagesegmentation_amd import data
from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator

dev = torch.device("cuda:0")
N = 2048
ds = data.PolygonDataset.synthetic(64, image_size=28, prediction_size=28, max_timesteps=10, history_length=2, seed=1,
                                   end_token=True)
pol = ExperimentPolicyEstimator(28, 28, max_timesteps=10, end_token=True, compute_dtype=torch.bfloat16, device=dev)
runner = BatchedReinforce(pol, ds, N)
rs = np.random.RandomState(0)
idx = rs.randint(0, len(ds), [N])
images = ds._images[idx]
polygons = [ds._vertices[i] for i in idx]
ep = runner.rollout(images, polygons)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ep = runner.rollout(images, polygons)
torch.cuda.synchronize()
print("eager rollout ms", (time.perf_counter() - t0) / 3 * 1e3, float(ep["reward"].mean()))

env = runner.env
env.reset(images, polygons)
features = pol.encode(env.image)
state = pol.get_rnn_zero_state(N)
static_pts = env.history_points.clone()


def one_step():
    global state
    logits, state2 = pol.step_logits(features, env.history_points, state)
    a = pol.sample_actions(logits).to(torch.int32)
    env.step(a, compute_reward=False)
    return a, state2


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.no_grad():
        one_step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.no_grad():
        with torch.cuda.graph(g, capture_error_mode="relaxed"):
            a, st = one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay of one step: ms", (time.perf_counter() - t0) / 10 * 1e3)
except Exception as e:  # noqa: BLE001
    print("capture failed:", repr(e)[:500])
