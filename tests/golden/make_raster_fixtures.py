"""Generates tests/golden/raster_golden.npz and env_golden.json by RUNNING THE REFERENCE.

Run once, in the build container only, with the interpreter that has scikit-image 0.18.3:

    /opt/conda/bin/python3.9 -B -W ignore tests/golden/make_raster_fixtures.py

It imports the reference's own ``util``, ``generate_polygons``, ``polyrl.polygon_env`` and
``polyrl.polygon_env_end_token`` from /root/reference (read-only; -B keeps bytecode out of
it) and records inputs and outputs only.  The fixtures are data; no reference source is
stored.  The reference never travels to the GPU box - only these files do.
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402

import generate_polygons  # noqa: E402
import util  # noqa: E402
from polyrl import polygon_env, polygon_env_end_token  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
VMAX = 16


def pack_mask(mask):
    return np.packbits(mask.astype(np.uint8).reshape(-1))


def random_polygons(rs, size, count):
    """Mixed bag: arbitrary (self-intersecting) vertex lists, duplicates, collinear runs,
    1- and 2-vertex degenerate inputs, a few negative coordinates."""
    polys = []
    for t in range(count):
        kind = t % 10
        if kind == 0:
            v = rs.randint(0, size, [1, 2])
        elif kind == 1:
            v = rs.randint(0, size, [2, 2])
        elif kind == 2:  # collinear run on a horizontal / vertical / diagonal line
            a = rs.randint(0, size, [2])
            n = rs.randint(3, 6)
            d = [(1, 0), (0, 1), (1, 1), (1, -1)][rs.randint(4)]
            v = np.array([[a[0] + d[0] * k, a[1] + d[1] * k] for k in rs.permutation(n)])
            v = np.clip(v, 0, size - 1)
        elif kind == 3:  # duplicate vertex
            nv = rs.randint(3, 9)
            v = rs.randint(0, size, [nv, 2])
            v[rs.randint(nv)] = v[rs.randint(nv)]
        elif kind == 4:  # axis-aligned rectangle (boundary-inclusive rule)
            x0, x1 = sorted(rs.randint(0, size, [2]))
            y0, y1 = sorted(rs.randint(0, size, [2]))
            v = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
        elif kind == 5:  # partly negative coordinates (rows/cols start at max(0, min))
            nv = rs.randint(3, 7)
            v = rs.randint(-size // 2, size, [nv, 2])
        else:  # arbitrary vertex list, V up to 13 (history + max_timesteps + 1)
            nv = rs.randint(3, 14)
            v = rs.randint(0, size, [nv, 2])
        polys.append(v.astype(np.int64))
    return polys


def convex_polygons(seed, size, count):
    """The reference generator recipe (generate_polygons.py:72-74), thresholds scaled by size/28."""
    np.random.seed(seed)
    s = size / 28.0
    out = []
    for _ in range(count):
        verts, _ = generate_polygons.create_valid_polygon(
            image_size=size, shape_complexity=10, min_area=10, min_distance=5 * s, max_distance=100 * s,
            reduction_tolerance=20)
        out.append(np.floor(verts * size).astype(np.int64))
    return out


def main():
    rs = np.random.RandomState(1234)
    arrays = {}
    meta = {}
    for size, n_rand, n_convex in [(8, 300, 0), (28, 300, 150), (32, 200, 80), (224, 40, 30)]:
        polys = random_polygons(rs, size, n_rand)
        if n_convex:
            polys += convex_polygons(size, size, n_convex)
        n = len(polys)
        verts = np.zeros([n, VMAX, 2], np.int32)
        nverts = np.zeros([n], np.int32)
        masks = np.zeros([n, (size * size + 7) // 8], np.uint8)
        area = np.zeros([n], np.int32)
        for i, p in enumerate(polys):
            verts[i, :len(p)] = p
            nverts[i] = len(p)
            m = util.create_shape_mask(p, size)
            masks[i] = pack_mask(m)
            area[i] = int(np.count_nonzero(m))
        # IoU of consecutive pairs, exactly as polygon_env.py:49-53 computes it
        inter = np.zeros([n], np.int32)
        union = np.zeros([n], np.int32)
        iou = np.zeros([n], np.float64)
        for i in range(n):
            j = (i + 1) % n
            pm = util.create_shape_mask(polys[i], size)
            tm = util.create_shape_mask(polys[j], size)
            inter[i] = np.count_nonzero(np.logical_and(pm, tm))
            union[i] = np.count_nonzero(np.logical_or(pm, tm))
            iou[i] = inter[i] / union[i]
        arrays.update({"verts_%d" % size: verts, "nverts_%d" % size: nverts, "masks_%d" % size: masks,
                       "area_%d" % size: area, "inter_%d" % size: inter, "union_%d" % size: union,
                       "iou_%d" % size: iou})
        meta[str(size)] = {"count": n, "random": n_rand, "convex": n_convex}

    # out-of-image vertices: the reference raises IndexError (util.py:64)
    raised = []
    for p in ([[0, 0], [8, 0], [8, 5]], [[3, 3], [3, 9], [5, 5]]):
        try:
            util.create_shape_mask(np.array(p), 8)
            raised.append(False)
        except IndexError:
            raised.append(True)
    meta["out_of_image_raises_IndexError"] = raised

    # generator outputs for recorded seeds (validates the numpy-only generator restatement)
    gen = {}
    for size in (28, 224):
        s = size / 28.0
        for seed in (0, 1, 2):
            np.random.seed(seed)
            vs, areas = [], []
            for _ in range(8):
                verts, image = generate_polygons.create_valid_polygon(size, 10, 10, 5 * s, 100 * s, 20)
                vs.append(np.round(verts * size).astype(int).tolist())
                areas.append(int(np.count_nonzero(image)) // 3)
            gen["%d_%d" % (size, seed)] = {"verts": vs, "areas": areas}
    meta["generator"] = gen
    np.savez_compressed(os.path.join(HERE, "raster_golden.npz"), **arrays)

    # ---- environment traces -------------------------------------------------------------------
    class ReplayDataset:
        """raw_sample hands out recorded (image, vertices) pairs; duck-types Dataset.raw_sample."""

        def __init__(self, size, hist, samples):
            self._prediction_size = size
            self._history_length = hist
            self._samples = samples
            self._i = 0

        def raw_sample(self, batch_size):
            assert batch_size == 1
            image, verts = self._samples[self._i % len(self._samples)]
            self._i += 1
            vs = np.empty([1], dtype=object)
            vs[0] = verts
            return np.array([image]), vs

    def state_summary(state):
        nz = np.argwhere(state[:, :, 3:] != 0)
        return {"shape": list(state.shape), "hist_nonzero": nz.tolist(), "image_sum": float(state[:, :, :3].sum())}

    traces = []
    rs = np.random.RandomState(77)
    for size, hist in [(28, 2), (28, 1), (32, 2)]:
        polys = convex_polygons(100 + size + hist, size, 6)
        for k, p in enumerate(polys):
            image = np.stack([util.create_shape_mask(p, size)] * 3, axis=2)
            p = np.roll(p, rs.randint(len(p)), axis=0)
            for variant, env_mod in (("close", polygon_env), ("end_token", polygon_env_end_token)):
                ds = ReplayDataset(size, hist, [(image, p)])
                env = env_mod.PolygonEnv(ds)
                st = env.reset()
                tr = {"variant": variant, "size": size, "hist": hist, "vertices": p.tolist(),
                      "reset": state_summary(st), "steps": []}
                # action script: ground truth for even k, noisy for odd k
                actions = []
                nv = len(p)
                for t in range(hist, nv + 1):
                    a = p[t % nv].copy()
                    if k % 2 == 1 and t < nv:
                        a = np.clip(a + rs.randint(-3, 4, [2]), 0, size - 1)
                    actions.append([int(a[0]), int(a[1])])
                if variant == "end_token":
                    actions = actions[:-1] + [[size, 0]]
                for a in actions:
                    if env.done:
                        break
                    st, reward, done = env.step(a)
                    tr["steps"].append({"action": a, "reward": float(reward), "done": bool(done),
                                        "state": state_summary(st),
                                        "predicted_vertices": np.array(env.predicted_vertices).tolist()})
                # stepping a finished env raises NotImplementedError (polygon_env.py:32-33)
                if env.done:
                    try:
                        env.step(actions[-1])
                        tr["step_after_done"] = "no error"
                    except NotImplementedError as e:
                        tr["step_after_done"] = str(e)
                traces.append(tr)
    meta["env_traces"] = traces
    with open(os.path.join(HERE, "env_golden.json"), "w") as f:
        json.dump(meta, f)
    print("wrote fixtures:", {k: v for k, v in meta.items() if k not in ("env_traces", "generator")})


if __name__ == "__main__":
    main()
