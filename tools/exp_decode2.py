"""Diagnostic: where a B = 16 validate_iou spends its time (graph replay alone vs the host work around it)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagesegmentation_amd import data
from imagesegmentation_amd.polyrnn import experiments
dev = torch.device("cuda:0")
model = experiments.make("EndTokenDeep_2", image_size=28, compute_dtype=torch.bfloat16, device=dev, seed=0)
ds = data.PolygonDataset.synthetic(64, 28, 28, 10, 2, seed=50, end_token=True)
rs = np.random.RandomState(60)
B = 16
idx = rs.randint(0, len(ds), [B])
images = torch.as_tensor(ds._images[idx], device=dev).to(torch.bfloat16)
verts = [ds._vertices[i] for i in idx]
for _ in range(3):
    model.validate_iou(images, verts, summary_prefix=None)
g = next(iter(model._decode_graph_cache.values()))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    g.graph.replay()
b.record()
torch.cuda.synchronize()
print("graph replay alone: %.3f ms" % (a.elapsed_time(b) / 20))
t0 = time.perf_counter()
for _ in range(20):
    model.validate_iou(images, verts, summary_prefix=None)
torch.cuda.synchronize()
print("validate_iou: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    model.validate_iou(images, verts, summary_prefix=None)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
