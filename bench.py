#!/usr/bin/env python3
"""Benchmark of the two hot paths on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config rnnconv|endtokendeep2] [--scaling weak|strong]

Headline (the JSON line's metric/value): polygon sequences per second, forward + backward + optimiser,
of the RNNConvExperiment graph (encoder 32@14,32@5,32@3 -> 2 x ConvLSTM(16) -> head) on 224 x 224
synthetic polygons, batch 64 per GPU, teacher forced, T = 5.  The
reference's dense head (P*P*16 -> P*P) has 4e10 weights at P = 224 and cannot exist; as the reference
never runs above 28 x 28, the head here average-pools the recurrent output 8x to the 28 x 28 grid the
reference's head was built for (Model(head_size=28)).  ``--config endtokendeep2`` times the
EndTokenDeepExperiment_2 graph (7 encoder layers, 5 x ConvLSTM(16), T = 10, P*P+1 classes, 32 sequences per GPU,
28 x 28 unless --size says otherwise) in the same line shape.

Also reported in the same line: IoU rollouts per second of the rasteriser (16,384 rollouts at 224 x 224,
masks materialised), per-kernel rooflines measured with events on the launch stream, and CPU baselines
timed on the host cores in the same run (rank 0, N = 1 only).

One process per GPU (RCCL all-reduce of the flat gradient bucket per step; the rasteriser shards rollouts
with no collective).  `python bench.py --gpus N` with N > 1 starts its own N ranks through
torch.distributed.run before anything touches a GPU and exits with their status; started under a launcher
already (WORLD_SIZE set) it is one of the ranks.  ``--scaling weak`` (default) keeps the per-GPU work fixed,
``--scaling strong`` keeps the global batch / rollout counts fixed and splits them over the ranks.

Every rank walks ONE control flow (``run_legs``); whatever issues a collective (a training step, an RL update, a
fence, a timing reduction) runs on every rank or on none.  ``--launch-check`` walks the same function with
stand-in legs that do no GPU work but issue the same collectives through the same ``parallel`` helpers on gloo
host tensors, and checks before each collective that all ranks are about to issue the same one.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
HBM_PEAK_GBS = 8000.0


CONFIGS = {
    # name -> (experiment graph, default size, default sequences per GPU)
    "rnnconv": ("RNNConv", 224, 64),              # reference polyrnn/RNNConvExperiment.py
    "endtokendeep2": ("EndTokenDeep_2", 28, 32),  # reference polyrnn/EndTokenDeepExperiment_2.py:19-57,202-205,224-226
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--config", default="rnnconv", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch / --rollouts / --rl-envs / --a3c-envs are per GPU; strong: they are the "
                         "whole job's and are split over the ranks")
    ap.add_argument("--size", type=int, default=None, help="image side (default: the config's)")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default: the config's)")
    ap.add_argument("--rollouts", type=int, default=16384, help="IoU rollouts per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--rl-envs", type=int, default=2048, help="REINFORCE rollouts per GPU per step (0 = skip)")
    ap.add_argument("--no-miopen-find", action="store_true",
                    help="keep MIOpen's immediate-mode pick for the encoder convs instead of letting it time its solvers "
                         "during warm-up")
    ap.add_argument("--a3c-envs", type=int, default=1024, help="actor-critic environments per GPU (0 = skip)")
    ap.add_argument("--settle-seconds", type=float, default=0.0,
                    help="idle pause before the warm-up (diagnostic; see main)")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins the process group (ISG_DIST_BACKEND, default gloo) and walks "
                         "the whole control flow of a run with stand-in legs that issue the real collectives on host "
                         "tensors; rank 0 prints a JSON line - checks the multi-rank path on a machine without GPUs")
    ap.add_argument("--no-extra-runs", action="store_true",
                    help="skip the extra legs (full-length sequences, f32, fresh batches, the other config, decode)")
    ap.add_argument("--kernels-only", action="store_true",
                    help="launch each hot kernel 3 times and exit (the run the PMC counter passes profile)")
    args = ap.parse_args()
    _, size, batch = CONFIGS[args.config]
    args.size = size if args.size is None else args.size
    args.batch = batch if args.batch is None else args.batch
    return args


def make_batch(size, batch, T, hist, head_size, seed, device, dtype, end_token=False):
    """Synthetic polygons by the reference generator recipe, rasterised by the HIP rasteriser, laid out
    as the reference's teacher-forcing batch (history as vertex points, targets on the head grid)."""
    from imagesegmentation_amd import data, util
    cache = {}

    def mask_fn(verts, s):
        key = verts.tobytes()
        if key not in cache:
            cache.clear()
            cache[key] = util.create_shape_mask(verts, s)
        return cache[key]

    ds = data.PolygonDataset.synthetic(batch, size, size, T, hist, seed=seed, shape_mask_fn=mask_fn,
                                       end_token=end_token)
    np.random.seed(seed + 1)
    d, images, pts, targets, verts = ds.get_batch_points(batch)
    d = np.minimum(d, T).astype(np.int32)
    scale = size // head_size
    targets = targets // scale
    return dict(durations=d, images=torch.as_tensor(images, device=device).to(dtype).contiguous(),
                points=torch.as_tensor(pts, device=device), targets=torch.as_tensor(targets, device=device),
                vertices=verts, images_np=images, points_np=pts, targets_np=targets, dataset=ds)


def event_time_ms(fn, reps):
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / reps


def time_kernels(model, batch, dtype_name, reps=5, in_step=None):
    """Average launch duration of each ConvLSTM kernel at the workload's two layer shapes, with HIP events
    on the stream the kernels are launched on (torch's current stream): ``ms`` inside the training step
    (``in_step``, from ops.timeline) and ``ms_isolated`` from back-to-back launches on fresh buffers."""
    from imagesegmentation_amd import ops
    dt = model.compute_dtype
    dev = model.device
    B, P = batch["images"].shape[0], model.prediction_size
    F = 16
    out = []
    n_layers = len(model.rnn.cells)
    for l, cell in enumerate(model.rnn.cells):
        if l not in (0, 1, n_layers - 1):
            continue  # layers 1 .. L-2 are the same launch
        hist = model.history_length if l == 0 else 0  # layer 0 takes its history frames as points
        packed = cell.packed(hist=hist)
        cs, c_real = packed.cs, cell.input_to_state_kernel.shape[2]
        pts = torch.randint(0, P, [B, hist, 2], device=dev, dtype=torch.int32) if hist else None
        x = torch.randn([B, P, P, cs], device=dev).to(dt)
        h = (torch.randn([B, P, P, F], device=dev) * 0.5).to(dt)
        # what the forward step keeps for backward: the backward factors where the fused backward step exists, else
        # the gate activations for the three-kernel backward (four values per channel either way); with the fused
        # step the cell state is stored as bf16 (as the training step stores it)
        fused = ops.bwd_fused_supported(packed, B, P, P)
        layout = ops.SAVE_FACTORS if fused else ops.SAVE_GATES
        c_dt = dt if fused and ops.state_bf16_supported(packed, B, P, P) else torch.float32
        c = torch.randn([B, P, P, F], device=dev).to(c_dt)
        h2, c2 = torch.empty_like(h), torch.empty_like(c)
        gates = torch.zeros(ops.saved_shape(B, P, P, F, layout), dtype=dt, device=dev)
        dy = torch.empty([B, P, P, F, 4], dtype=dt, device=dev)
        dh = torch.randn([B, P, P, F], device=dev).to(dt)
        dc = torch.randn([B, P, P, F], device=dev).to(dt)  # carried in the activation dtype, as in the train step
        dcp = torch.empty_like(dc)
        dx = torch.empty([B, P, P, cs], dtype=dt, device=dev)
        dhp = torch.empty_like(h)
        dwx = torch.zeros_like(cell.input_to_state_kernel)
        dwh = torch.zeros_like(cell.hidden_to_state_kernel)
        db = torch.zeros_like(cell.bias)
        flops = 2.0 * B * P * P * 9 * (c_real + F) * 4 * F
        esz = 2 if dt == torch.bfloat16 else 4
        px = B * P * P
        # algorithmic HBM bytes per pixel: every operand read once, every result written once
        x_b, h_b, c_b, g_b = (cs if hist else c_real) * esz, F * esz, F * c.element_size(), 4 * F * esz
        kernels = [
            ("cell_fwd", lambda: ops.cell_fwd(x, h, c, packed, h_out=h2, c_out=c2, gates=gates, hist_points=pts,
                                              save_layout=layout), flops,
             px * (x_b + h_b + c_b + h_b + c_b + g_b)),
        ]
        top = l == len(model.rnn.cells) - 1
        pool = P // model.head_size if top else 1
        if fused:
            # one launch: the saved factors, c_prev, the step's h and the three incoming gradients in (the top
            # layer's comes from the pooled head grid), [x | h_prev] in, dc_prev / dh_prev / dx out (layer 0 adds to
            # its running dx)
            above = torch.randn([B, P // pool, P // pool, F], device=dev).to(dt)
            dcn, dhn = torch.empty_like(dc), torch.empty_like(dh)
            ws = ops.wgrad_workspace(cs, F, dev, cached=False)
            dx_ch = cs if hist else F
            dx_f = torch.zeros([B, P, P, dx_ch], dtype=dt, device=dev)
            acc_dx = l == 0
            dx_b = dx_ch * esz * (2 if acc_dx else 1)
            kernels.append(("bwd_fused", lambda: ops.bwd_fused(gates, c, h2, above, (P // pool) ** 2 * F, dh, dc, x, h, packed,
                                                                dcn, dhn, ws, ops.WGRAD_DEFER, dwx, dx=dx_f,
                                                                dx_channels=dx_ch, dx_accumulate=acc_dx, hist_points=pts,
                                                                pool=pool),
                            2.0 * flops, px * (g_b + c_b + h_b + h_b / (pool * pool) + 2 * h_b + x_b + h_b + 2 * h_b + dx_b)))
        else:
            kernels += [
                ("pointwise_bwd", lambda: ops.pointwise_bwd(gates, c, c2, dh, P * P * F, dh, dc, dy=dy, dc_prev=dcp),
                 0.0, px * (2 * g_b + 2 * h_b + 2 * c_b + 2 * h_b)),  # the dc carry is stored like dh
                ("dgrad", lambda: ops.dgrad(dy, packed, dhp, dx=dx, dx_channels=cs), flops, px * (g_b + x_b + h_b)),
                ("wgrad", lambda: ops.wgrad(x, h, dy, packed.c, dwx, dwh, db, accumulate=False, hist_points=pts, hist=hist),
                 flops, px * (x_b + h_b + g_b)),
            ]
        for name, fn, work, nbytes in kernels:
            ms_isolated = event_time_ms(fn, reps)
            # duration inside the training step (events around the same launch in two extra steps) when known:
            # that is the launch the step pays for, with its operands as warm or cold as the step leaves them
            ms = (in_step or {}).get("%s[layer%d]" % (name, l), ms_isolated)
            rec = {"kernel": "%s[layer%d C=%d]" % (name, l, c_real), "ms": ms, "ms_isolated": ms_isolated,
                   "flops": work, "bytes": nbytes,
                   "mfma_frac": work / ms / 1e9 / MFMA_PEAK_TFLOPS[dtype_name],
                   "hbm_frac": nbytes / ms / 1e6 / HBM_PEAK_GBS}
            # the roof that binds is the one the kernel sits closer to
            if rec["mfma_frac"] > rec["hbm_frac"]:
                rec.update(bound="mfma", achieved=work / ms / 1e9, peak=MFMA_PEAK_TFLOPS[dtype_name], unit="TFLOP/s",
                           frac=rec["mfma_frac"])
            else:
                rec.update(bound="hbm", achieved=nbytes / ms / 1e6, peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=rec["hbm_frac"])
            out.append(rec)
    return out


def bench_reinforce(n_envs, steps, device, rank, dtype):
    """Sampled rollouts + terminal IoU + one policy update per step: the reference's REINFORCE loop
    (end-token 2-layer net at 28x28, at most 10 sampled steps) for n_envs lock-stepped episodes."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator
    ds = data.PolygonDataset.synthetic(64, image_size=28, prediction_size=28, max_timesteps=10, history_length=2,
                                       seed=10 + rank, end_token=True)
    policy = ExperimentPolicyEstimator(28, 28, max_timesteps=10, end_token=True, compute_dtype=dtype, device=device,
                                       seed=0)
    # episodes are drawn from the device-resident copy of the dataset (64 polygons, reused with replacement
    # beyond 64 environments, each rolled to a random start vertex): a fresh draw per rollout, no host work
    runner = BatchedReinforce(policy, ds, n_envs, device_pool=True, pool_seed=20 + rank)
    runner.step()
    runner.rollout()  # (untimed like the step: a rollout by itself keeps nothing for an update - other kernel variants)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ep = runner.rollout()
    torch.cuda.synchronize()
    t_roll = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        ep, _ = runner.step()
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    return {"envs": n_envs, "size": 28, "max_steps": 10, "episode_source": "device pool",
            "rollouts_per_s_rollout_only": n_envs / t_roll,
            "rollouts_per_s_with_update": n_envs / t_step, "ms_per_step": t_step * 1e3,
            "mean_reward": float(ep["reward"].mean()), "mean_length": float(ep["lengths"].float().mean())}


def bench_a3c(n_envs, steps, device, rank, dtype):
    """Synchronous actor-critic updates (the a3c specification: 4 x conv32 -> ConvLSTM(32) -> policy / value heads
    at 32 x 32, rollouts of at most 10 steps) over n_envs lock-stepped environments: env-steps per second."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv, DevicePolygonPool
    ds = data.PolygonDataset.synthetic(64, image_size=32, prediction_size=32, max_timesteps=10, history_length=2,
                                       seed=30 + rank)
    env = BatchedPolygonEnv(ds, n_envs, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=device)
    trainer = a3c.A3C(env, compute_dtype=dtype, seed=0)
    # episodes are drawn from the device-resident copy of the dataset, a fresh draw per update (each polygon rolled to
    # a random start vertex), as in the REINFORCE leg: no host work at reset
    pool = DevicePolygonPool(ds, device, dtype=torch.float32, vcap=2 + a3c.A3C.TIMESTEP_LIMIT + 1, seed=40 + rank)
    # (no solver search for any library convolution a fallback path of this leg might still take)
    find = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    # untimed: the first update launches its acting loop step by step, the second captures it as a graph
    # (A3C.rollout_graph), from the third on it is replayed
    for _ in range(3):
        trainer.process(device_batch=pool.sample(n_envs))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env_steps = 0
    for _ in range(steps):
        trainer.process(device_batch=pool.sample(n_envs))
        env_steps += trainer.last_env_steps
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    torch.backends.cudnn.benchmark = find
    return {"envs": n_envs, "size": 32, "filters": 32, "episode_source": "device pool",
            "env_steps_per_s": env_steps / el, "ms_per_update": el / steps * 1e3,
            "loss": trainer.last["loss"]}


def csrc_hash():
    """Fingerprint of the kernel sources (csrc/*.hip, *.h): the counter passes are stamped with it."""
    import glob
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "imagesegmentation_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_pmc_traffic():
    """HBM bytes per launch from the committed counter passes (FETCH_SIZE and WRITE_SIZE need a profiler
    run each, so they cannot be read inside the timed run): {kernel label: bytes}.  Empty when the file is
    absent or was taken with other kernel sources than the ones in the tree (its csrc_hash stamp)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("csrc_hash") != csrc_hash():
            return {}
        return {k: v["hbm_bytes"] for k, v in rec["kernels"].items()}
    except (OSError, ValueError, KeyError):
        return {}


def cpu_baseline_convlstm(model, batch, size, T, hist, head_size, budget_s=12.0):
    """The same graph as stock torch CPU ops (oracle/torch_port.py) on the host cores, fp32, on a bounded
    sample of the workload (2 sequences per pass)."""
    from oracle import torch_port
    threads = torch.get_num_threads()
    n = 2
    tp = {"encoder": [dict(w=w.detach().cpu().float().requires_grad_(True), b=b.detach().cpu().float().requires_grad_(True))
                      for w, b in model.encoder],
          "rnn": [dict(w_x=c.input_to_state_kernel.detach().cpu().float().requires_grad_(True),
                       w_h=c.hidden_to_state_kernel.detach().cpu().float().requires_grad_(True),
                       bias=c.bias.detach().cpu().float().requires_grad_(True)) for c in model.rnn.cells],
          "dense": dict(w=model.dense_kernel.detach().cpu().float().requires_grad_(True),
                        b=model.dense_bias.detach().cpu().float().requires_grad_(True))}
    images = torch.as_tensor(batch["images_np"][:n]).float()
    pts = batch["points_np"][:n]
    hd = torch.zeros([n, T, size, size, hist])
    for b in range(n):
        for t in range(T):
            for i in range(hist):
                x, y = pts[b, t, i]
                if x >= 0:
                    hd[b, t, y, x, i] = 1
    dur = torch.as_tensor(batch["durations"][:n].astype(np.int64))
    labels = torch.as_tensor(batch["targets_np"][:n].astype(np.int64))
    labels = labels[..., 0] * head_size + labels[..., 1]
    done, t0 = 0, time.perf_counter()
    while True:
        logits, _ = torch_port.model_logits(tp, images, hd, dur, head_pool=size // head_size)
        loss = torch.nn.functional.cross_entropy(logits.reshape(n * T, -1), labels.reshape(-1))
        loss.backward()
        done += n
        el = time.perf_counter() - t0
        if el > budget_s or done >= 512:  # bounded sample: about budget_s seconds of CPU work
            break
    return {"value": done / el, "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": "%d sequences (batch %d per pass) of the same 224x224 RNNConv workload, fwd+bwd, fp32 torch CPU "
                      "ops, %.1f s" % (done, n, el)}


def rollout_polygons(n, size, seed):
    """Predicted (3..13 random vertices, as an untrained policy emits) and true (convex-ish) polygons."""
    rs = np.random.RandomState(seed)
    vmax = 16
    nvp = rs.randint(3, 14, [n]).astype(np.int32)
    vp = rs.randint(0, size, [n, vmax, 2]).astype(np.int32)
    nvt = rs.randint(3, 9, [n]).astype(np.int32)
    ang = np.sort(rs.uniform(0, 2 * np.pi, [n, vmax]), axis=1)
    rad = rs.uniform(0.2, 0.48, [n, 1]) * size
    ctr = size / 2.0
    vt = np.stack([ctr + rad * np.cos(ang), ctr + rad * np.sin(ang)], axis=2).astype(np.int32)
    return vp, nvp, vt, nvt


def bench_raster(n, size, steps, warmup, device, rank, do_cpu):
    from imagesegmentation_amd import util
    vp, nvp, vt, nvt = rollout_polygons(n, size, 100 + rank)
    t = lambda a: torch.from_numpy(a).to(device)
    VP, NVP, VT, NVT = t(vp), t(nvp), t(vt), t(nvt)
    fn = lambda: util.raster_iou(VP, NVP, VT, NVT, size, want_masks=True)
    for _ in range(max(warmup, 1)):
        fn()
    ms = event_time_ms(fn, steps)
    fn_counts = lambda: util.raster_iou(VP, NVP, VT, NVT, size)
    ms_counts = event_time_ms(fn_counts, steps)
    bytes_per = 2 * size * size + 2 * vp.shape[1] * 8 + 8 + 8
    res = {"rollouts_per_s": n / ms * 1e3, "ms_per_launch": ms, "rollouts": n, "size": size,
           "counts_only_rollouts_per_s": n / ms_counts * 1e3,
           "roofline": {"bound": "hbm", "achieved": n * bytes_per / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": n * bytes_per / ms / 1e6 / HBM_PEAK_GBS,
                        "traffic": load_pmc_traffic().get("raster_iou[masks]") if (n == 16384 and size == 224) else None,
                        "algorithmic_bytes": n * bytes_per, "kernel": "raster_iou_kernel (masks materialised)"}}
    if do_cpu:
        import oracle
        lib = oracle.load_c()
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        sample = 512
        i32p, u8p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8)
        inter, union = np.zeros([sample], np.int32), np.zeros([sample], np.int32)
        mp, mt = np.zeros([sample, size, size], np.uint8), np.zeros([sample, size, size], np.uint8)
        args = [np.ascontiguousarray(a[:sample]) for a in (vp, nvp, vt, nvt)]
        done, t0 = 0, time.perf_counter()
        while True:
            lib.isg_oracle_raster_iou(args[0].ctypes.data_as(i32p), args[1].ctypes.data_as(i32p),
                                      args[2].ctypes.data_as(i32p), args[3].ctypes.data_as(i32p), sample,
                                      vp.shape[1], size, inter.ctypes.data_as(i32p), union.ctypes.data_as(i32p),
                                      mp.ctypes.data_as(u8p), mt.ctypes.data_as(u8p), threads)
            done += sample
            el = time.perf_counter() - t0
            if el > 10.0:
                break
        gi, gu = fn_counts()
        ok = bool(np.array_equal(gi[:sample].cpu().numpy(), inter) and np.array_equal(gu[:sample].cpu().numpy(), union))
        res["cpu_baseline"] = {"value": done / el, "unit": "rollouts/s", "cores": threads, "kind": "port",
                               "sample": "%d rollouts (the first %d of the launch, repeated) through the C oracle with "
                                         "OpenMP, %.1f s; counts equal to the GPU's: %s" % (done, sample, el, ok)}
    return res


def note(msg):
    """Progress on stderr (stdout carries the one JSON line): a long silent run looks hung to a supervisor."""
    if int(os.environ.get("RANK", "0")) == 0:
        mem = torch.cuda.memory_allocated() / 2 ** 30 if torch.cuda.is_initialized() else 0.0
        print("[bench %6.1f s, %5.1f GiB held] %s" % (time.perf_counter() - _T0, mem, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def self_launch(n):
    """Starts n ranks of this script (same arguments) as children through torch.distributed.run and returns their
    exit status.  Called before this process has touched a GPU; rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Comm:
    """The bench's own collectives (fences and reductions of timings), on device tensors in a real run and on host
    tensors under --launch-check.  With one rank nothing is issued."""

    def __init__(self, world, device, sync):
        self.world, self.device, self.sync = world, device, sync

    def fence(self):
        self.sync()
        if self.world > 1:
            torch.distributed.barrier()
        self.sync()

    def _reduce(self, value, op):
        if self.world == 1:
            return float(value)
        t = torch.tensor([value], device=self.device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value):
        return self._reduce(value, torch.distributed.ReduceOp.MAX)

    def sum(self, value):
        return self._reduce(value, torch.distributed.ReduceOp.SUM)


def per_rank(args, world):
    """Work of one rank: the arguments themselves under weak scaling, their share under strong scaling."""
    n = world if args.scaling == "strong" else 1
    for name in ("batch", "rollouts", "rl_envs", "a3c_envs"):
        if getattr(args, name) % n != 0:
            raise SystemExit("--scaling strong: --%s %d does not divide over %d ranks" %
                             (name.replace("_", "-"), getattr(args, name), world))
    return dict(batch=args.batch // n, rollouts=args.rollouts // n, rl_envs=args.rl_envs // n,
                a3c_envs=args.a3c_envs // n)


class GpuWorkload:
    """The legs of a real run on this rank's GPU.  Methods that issue collectives say so; ``run_legs`` calls those on
    every rank."""

    def __init__(self, args, world, rank, local_rank):
        self.args, self.world, self.rank = args, world, rank
        # rehearsal on a box with fewer GPUs than ranks: ISG_DIST_BACKEND=gloo lets several ranks share a device
        backend = os.environ.get("ISG_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if not args.no_miopen_find:
            torch.backends.cudnn.benchmark = True
        self.device = device = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=device)
            else:
                torch.distributed.init_process_group(backend)
        self.comm = Comm(world, device, torch.cuda.synchronize)
        self.share = per_rank(args, world)
        self.dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        self.size = args.size
        self.head_size = 28 if args.size % 28 == 0 else args.size
        self.model, self.batch = self._make(CONFIGS[args.config][0], self.size, self.share["batch"], self.dtype, seed=rank)
        self.T, self.hist = self.model.max_timesteps, self.model.history_length

    def _make(self, graph, size, batch, dtype, seed):
        from imagesegmentation_amd.polyrnn import experiments
        head_size = 28 if size % 28 == 0 else size
        model = experiments.make(graph, image_size=size, compute_dtype=dtype, device=self.device, seed=0,
                                 head_size=head_size)
        bt = make_batch(size, batch, model.max_timesteps, model.history_length, head_size, seed=seed,
                        device=self.device, dtype=dtype, end_token=model.end_token)
        # small grids: a step is hundreds of launches of a few microseconds, the host is the bottleneck - replay the
        # step as a HIP graph (ISG_NO_STEP_GRAPH=1: launch kernel by kernel, for A/B runs)
        model.step_graphs = size <= 64 and not os.environ.get("ISG_NO_STEP_GRAPH")
        return model, bt

    def close(self):
        if self.world > 1:
            torch.distributed.destroy_process_group()

    # -- the training step (collectives: the early tail all-reduce from the autograd hook + the remainder) ----
    def train_step(self, model=None, batch=None, durations=None):
        m = self.model if model is None else model
        b = self.batch if batch is None else batch
        return m.train_step(b["images"], b["durations"] if durations is None else durations, b["points"], b["targets"])

    def describe(self):
        m = self.model
        return {"workload": "polyrnn %sExperiment graph, teacher-forced training step (fwd+bwd+clip+Adam), %dx%d "
                            "synthetic polygons, batch %d per GPU, T=%d, %dxConvLSTM(16), dense head on the %dx%d grid%s" %
                            (CONFIGS[self.args.config][0], self.size, self.size, self.share["batch"], self.T,
                             len(m.rnn.cells), self.head_size, self.head_size,
                             " (8x-pooled)" if self.head_size != self.size else ""),
                "mean_duration": float(np.mean(self.batch["durations"])),
                "step_as_graph": bool(self.model.step_graphs)}

    def allreduce_bytes(self):
        return int(self.model._flat_grad.numel() * 4)

    def conv_flops_per_sequence(self, model=None, durations=None):
        m = self.model if model is None else model
        d = self.batch["durations"] if durations is None else durations
        return 3.0 * sum(2.0 * self.size * self.size * 9 * (c.input_to_state_kernel.shape[2] + 16) * 64
                         for c in m.rnn.cells) * float(np.mean(d))

    def kernels_only(self):
        # every label exactly 3 launches, in the order tools/pmc_traffic.py expects
        time_kernels(self.model, self.batch, self.args.dtype, reps=2)
        from imagesegmentation_amd import util
        polys = [torch.from_numpy(a).to(self.device) for a in rollout_polygons(self.args.rollouts, self.size, 100 + self.rank)]
        for want_masks in (True, False):
            for _ in range(3):
                util.raster_iou(*polys, self.size, want_masks=want_masks)
        torch.cuda.synchronize()

    # -- legs without collectives ---------------------------------------------------------------------------
    def raster_leg(self, do_cpu):
        return bench_raster(self.share["rollouts"], self.size, max(self.args.steps, 5), self.args.warmup, self.device,
                            self.rank, do_cpu=do_cpu)

    def timeline_steps(self, n):
        """n more training steps (collectives as in any step) with events around every cell launch."""
        from imagesegmentation_amd import ops
        with ops.timeline() as tl:
            for _ in range(n):
                self.train_step()
        return tl.summary()

    def kernel_rooflines(self, in_step):
        args = self.args
        kernels = time_kernels(self.model, self.batch, args.dtype, in_step=in_step)
        # the committed counter passes were taken at the default workload only
        default_shape = (args.config == "rnnconv" and self.share["batch"] == 64 and self.size == 224 and
                         args.dtype == "bf16")
        traffic = load_pmc_traffic() if default_shape else {}
        for k in kernels:
            k["traffic"] = traffic.get(k["kernel"])
        # The dominant kernel = the GPU function with the most time in a step (launches per step x duration).  ``ms``
        # is the mean duration of the label's REGULAR launches inside a step: the two sequence-end instantiations
        # (out of the zero state; no incoming state gradient) move fewer bytes and are timed under labels of their own.
        dom = max(kernels, key=lambda k: k["ms"])
        roofline = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                    "frac": dom["frac"], "traffic": dom["traffic"], "kernel": dom["kernel"],
                    "symbol": {"bwd_fused": "convlstm_bwd_fused_kernel", "cell_fwd": "conv3x3_pp_kernel",
                               "pointwise_bwd": "lstm_pointwise_bwd_kernel", "dgrad": "conv3x3_pp_kernel",
                               "wgrad": "conv_wgrad_pp_kernel"}.get(dom["kernel"].split("[")[0]),
                    "launches_per_step": self.T, "ms_per_launch": dom["ms"], "algorithmic_bytes": dom["bytes"],
                    "flops": dom["flops"], "mfma_frac": dom["mfma_frac"], "hbm_frac": dom["hbm_frac"],
                    "instantiation": "regular (neither end of the sequence)",
                    "in_step_ms_by_label": {k: round(v, 4) for k, v in sorted(in_step.items())}}
        return kernels, roofline

    def encoder_rooflines(self, in_step):
        """The encoder's launches as timed inside the step (ops.timeline labels), each against both roofs: algorithmic
        bytes = every operand read once and every result written once, FLOPs = 2 x taps x Cin x Cout per pixel."""
        px = float(self.share["batch"]) * self.size * self.size
        peak = MFMA_PEAK_TFLOPS["bf16" if self.dtype == torch.bfloat16 else "f32"]
        out = []
        for label, ms in sorted(in_step.items()):
            name, _, shape = label.partition("[")
            if name not in ("conv_in_fwd", "conv_in_wgrad", "conv32", "conv32_masked", "conv32_wgrad", "relu_bwd_bias"):
                continue
            nums = [int(v) for v in shape.replace("]", "").replace("C=", "").replace("x", " ").split()]
            if name.startswith("conv_in"):
                ks, c = nums[0], self.model.encoder[0][0].shape[3]
                flops, nbytes = 2.0 * ks * ks * 3 * c * px, (8 + 2 * c) * px  # the image padded to 4 channels in
            elif name == "relu_bwd_bias":
                c = nums[0]
                flops, nbytes = 0.0, 3 * 2 * c * px
            else:
                ks, c = nums[0], nums[2]
                flops = 2.0 * ks * ks * c * c * px
                nbytes = (3 if name == "conv32_masked" else 2) * 2 * c * px
            out.append({"kernel": label, "ms": ms, "flops": flops, "bytes": nbytes,
                        "mfma_frac": flops / (ms * 1e-3) / (peak * 1e12), "hbm_frac": nbytes / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9)})
        return out

    def cpu_baseline(self):
        return cpu_baseline_convlstm(self.model, self.batch, self.size, self.T, self.hist, self.head_size)

    def decode_leg(self):
        """Autoregressive decode of the EndTokenDeep_2 net at 28 x 28 (reference polyrnn/Model.py:231-249,
        EndTokenDeepExperiment_2.py:150-170): milliseconds per decoded batch and per sequence, launches per vertex."""
        from imagesegmentation_amd import data
        from imagesegmentation_amd.polyrnn import experiments
        model = experiments.make("EndTokenDeep_2", image_size=28, compute_dtype=self.dtype, device=self.device, seed=0)
        ds = data.PolygonDataset.synthetic(64, 28, 28, model.max_timesteps, model.history_length, seed=50 + self.rank,
                                           end_token=True)
        out = {"net": "EndTokenDeep_2", "size": 28, "max_timesteps": model.max_timesteps}
        rs = np.random.RandomState(60 + self.rank)
        for B in (16, 2048):
            idx = rs.randint(0, len(ds), [B])
            images = torch.as_tensor(ds._images[idx], device=self.device).to(self.dtype)
            verts = [ds._vertices[i] for i in idx]
            res = {}
            for mode, graphs in (("graph", True), ("stepwise", False)):
                model.decode_graphs = graphs
                for _ in range(2):
                    model.validate_iou(images, verts, summary_prefix=None)
                torch.cuda.synchronize()
                reps = 5 if B <= 64 else 2
                t0 = time.perf_counter()
                for _ in range(reps):
                    iou = model.validate_iou(images, verts, summary_prefix=None)
                torch.cuda.synchronize()
                res[mode] = ((time.perf_counter() - t0) / reps * 1e3, iou, getattr(model, "decode_host_launches", None))
            ms, iou, launches = res["graph"]
            # ms_per_decode: validate_iou end to end (upload, decode loop as one graph launch, polygon assembly,
            # rasteriser + IoU, one copy back); *_stepwise: the same with every kernel launched from the host
            out["B%d" % B] = {"ms_per_decode": ms, "ms_per_sequence": ms / B, "sequences_per_s": B / ms * 1e3,
                              "host_launches_of_the_loop": launches, "ms_per_decode_stepwise": res["stepwise"][0],
                              "host_launches_of_the_loop_stepwise": res["stepwise"][2],
                              "same_iou_both_ways": bool(res["stepwise"][1] == iou), "mean_iou_untrained": iou}
        out["kernels_per_vertex"] = getattr(model, "decode_launches_per_step", None)
        return out

    def recurrence_leg(self):
        """The recurrence of the EndTokenDeep_2 net alone at 224 x 224 (reference polyrnn/EndTokenDeepExperiment_2.py:
        33-52: five ConvLSTM(16) layers, T = 10, two history frames), 32 sequences, forward + backward through
        ``dynamic_rnn`` on synthetic features, every sequence at full length - the net's dense head cannot exist at
        that size (50,177 x 802,816 weights), its recurrence can."""
        from imagesegmentation_amd.convLSTM import ConvLSTMCell, MultiRNNCell, SequenceInput, dynamic_rnn
        B, P, T, L, hist, C = 32, 224, 10, 5, 2, 16
        dev, dt = self.device, self.dtype
        gen = torch.Generator(device=dev)
        gen.manual_seed(90 + self.rank)
        multi = MultiRNNCell([ConvLSTMCell([P, P], 16, [3, 3], compute_dtype=dt, device=dev) for _ in range(L)])
        multi.build(C + hist, dev)
        params = [q for c in multi.cells for q in (c.input_to_state_kernel, c.hidden_to_state_kernel, c.bias)]
        feats = torch.randn([B, P, P, C], device=dev, generator=gen).to(dt).requires_grad_(True)
        pts = torch.randint(0, P, [B, T, hist, 2], device=dev, generator=gen, dtype=torch.int32)
        durations = torch.full([B], T, dtype=torch.int32, device=dev)
        d_out = (torch.randn([B, T, P, P, 16], device=dev, generator=gen) * 0.1).to(dt)

        def step():
            for q in params + [feats]:
                q.grad = None
            out, _ = dynamic_rnn(multi, SequenceInput(static=feats, hist_points=pts), sequence_length=durations,
                                 time_major=False, steps=T)
            out.backward(d_out)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        n = 4
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / n
        flops = 3.0 * T * P * P * sum(2 * 9 * ((C + hist if l == 0 else 16) + 16) * 64 for l in range(L))
        return {"workload": "EndTokenDeep_2 recurrence alone, 224x224, %d sequences, T=%d, L=%d, hist=%d, fwd+bwd" %
                            (B, T, L, hist), "ms_per_step": sec * 1e3, "sequences_per_s": B / sec,
                "launches": 2 * T * L, "ms_per_cell_launch_pair": sec * 1e3 / (T * L),
                "convlstm_gflop_per_sequence": flops / 1e9,
                "convlstm_mfma_frac": flops * B / sec / 1e12 / MFMA_PEAK_TFLOPS["bf16" if dt == torch.bfloat16 else "f32"]}

    # -- legs with collectives (training steps / RL updates inside) --------------------------------------------
    def reinforce_leg(self):
        return bench_reinforce(self.share["rl_envs"], 3, self.device, self.rank, self.dtype)

    def a3c_leg(self):
        return bench_a3c(self.share["a3c_envs"], 4, self.device, self.rank, self.dtype)

    def release(self):
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    def full_length_variant(self):
        return dict(durations=np.full_like(self.batch["durations"], self.T))

    def f32_variant(self):
        from imagesegmentation_amd.polyrnn import experiments
        m32 = experiments.make(CONFIGS[self.args.config][0], image_size=self.size, compute_dtype=torch.float32,
                               device=self.device, seed=0, head_size=self.head_size)
        return dict(model=m32, batch=dict(self.batch, images=self.batch["images"].float()))

    def other_config_variant(self):
        """The config this run is not on, at its own default shape (28 x 28, 32 sequences per GPU for EndTokenDeep_2;
        skipped when this run is EndTokenDeep_2 already)."""
        if self.args.config != "rnnconv":
            return None
        graph, size, batch = CONFIGS["endtokendeep2"]
        if self.args.scaling == "strong":  # the reference config's global batch of 256, split over the ranks
            batch = max(256 // self.world, 1)
        model, bt = self._make(graph, size, batch, self.dtype, seed=70 + self.rank)
        return dict(model=model, batch=bt, label="%s %dx%d, %d sequences per GPU, T=%d, L=%d" %
                    (graph, size, size, batch, model.max_timesteps, len(model.rnn.cells)), batch_size=batch)

    def fresh_batch_stepper(self):
        """A step that builds its batch first (reference polyrnn/EndTokenDeepExperiment_2.py:224-227 calls
        get_batch_for_rnn before every train): the dataset lives on the device, a batch is one gather of the images and
        one isg_batch_points launch; lengths stay on the device (every step of the sequence runs under the mask)."""
        from imagesegmentation_amd import data
        builder = data.DeviceBatchBuilder(self.batch["dataset"], self.device, image_dtype=self.dtype)
        scale, T, B, model = self.size // self.head_size, self.T, self.share["batch"], self.model

        def step():
            b = builder.get_batch_points(B, device_draw=True, seed=80 + self.rank)
            return model.train_step(b["images"], b["durations"].clamp(max=T), b["points"], b["targets"] // scale)
        return step


class StubWorkload:
    """--launch-check: the legs as stand-ins that do no GPU work and issue, through the same ``parallel`` helpers the
    real legs use, the collectives the real legs issue - on gloo host tensors.  A training step is a backward pass
    through a toy net whose last layer closes a flat gradient bucket (``parallel.EarlyTailReduce``: the tail
    all-reduced from the autograd hook, the remainder after the backward); an RL update is one
    ``parallel.allreduce_gradients_`` of a flat bucket.  Before every collective the ranks compare what they are
    about to issue and fail with a message when they differ."""

    def __init__(self, args, world, rank, local_rank):
        import datetime
        import torch.distributed as dist
        from imagesegmentation_amd import parallel
        self.args, self.world, self.rank = args, world, rank
        self.issued = 0
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("ISG_DIST_BACKEND", "gloo"), timeout=datetime.timedelta(seconds=60))
            self._guard(dist)
        if os.environ.get("ISG_BENCH_FAIL_RANK") == str(rank):
            raise SystemExit("rank %d asked to fail (ISG_BENCH_FAIL_RANK)" % rank)
        self.comm = Comm(world, "cpu", lambda: None)
        self.share = per_rank(args, world)
        self.T = 5
        g = torch.Generator().manual_seed(3)
        self._a = torch.nn.Parameter(torch.randn(5, 4, generator=g))
        self._b = torch.nn.Parameter(torch.randn(4, 3, generator=g))
        self._flat = torch.zeros([self._a.numel() + self._b.numel()])
        self._a.grad = self._flat[:20].view_as(self._a)
        self._b.grad = self._flat[20:].view_as(self._b)
        self._early = parallel.EarlyTailReduce(self._flat, [self._b], 20)
        self._x = torch.randn(7, 5, generator=torch.Generator().manual_seed(50 + rank))
        self._rl_flat = torch.zeros([16])
        self._parallel = parallel

    def _guard(self, dist):
        """Wraps all_reduce / barrier: the ranks first exchange a tag of the collective each is about to issue."""
        world = self.world
        gather = dist.all_gather_object

        def guarded(fn, name):
            def wrapper(*a, **k):
                tag = "%s#%d:%s" % (name, self.issued, a[0].numel() if a and torch.is_tensor(a[0]) else "")
                tags = [None] * world
                gather(tags, tag)
                if len(set(tags)) != 1:
                    raise RuntimeError("ranks disagree on the next collective: %s" % tags)
                self.issued += 1
                return fn(*a, **k)
            return wrapper
        dist.all_reduce = guarded(dist.all_reduce, "all_reduce")
        dist.barrier = guarded(dist.barrier, "barrier")
        self._gather = gather

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            counts = [None] * self.world
            self._gather(counts, self.issued)
            if len(set(counts)) != 1:
                raise RuntimeError("ranks issued different numbers of collectives: %s" % counts)
            dist.destroy_process_group()

    def train_step(self, model=None, batch=None, durations=None):
        time.sleep(0.001 * (self.rank + 1))
        self._flat.zero_()
        self._early.arm()
        loss = (torch.tanh(self._x @ self._a) @ self._b).square().sum()
        loss.backward()
        self._early.finish(average=True)
        return loss.detach()

    def _rl_update(self):
        self._parallel.allreduce_gradients_(self._rl_flat, average=False)

    def describe(self):
        return {"workload": "launch check (no GPU work)", "mean_duration": float(self.T)}

    def allreduce_bytes(self):
        return int(self._flat.numel() * 4)

    def conv_flops_per_sequence(self, model=None, durations=None):
        return 0.0

    def raster_leg(self, do_cpu):
        return {"ms_per_launch": 1.0 + self.rank, "rollouts": self.share["rollouts"], "rollouts_per_s": 0.0}

    def timeline_steps(self, n):
        for _ in range(n):
            self.train_step()
        return {}

    def kernel_rooflines(self, in_step):
        return None, None

    def encoder_rooflines(self, in_step):
        return []

    def cpu_baseline(self):
        return None

    def decode_leg(self):
        return None

    def recurrence_leg(self):
        return None

    def reinforce_leg(self):
        # bench_reinforce: one warm-up step, 1 + 3 rollouts (no exchange), 3 steps with the update
        for _ in range(1 + 3):
            self._rl_update()
        return {"ms_per_step": 1.0 + self.rank, "rollouts_per_s_rollout_only": 0.0, "envs": self.share["rl_envs"]}

    def a3c_leg(self):
        # bench_a3c: three untimed updates, 4 timed ones
        for _ in range(3 + 4):
            self._rl_update()
        return {"env_steps_per_s": 1.0, "envs": self.share["a3c_envs"]}

    def release(self):
        pass

    def full_length_variant(self):
        return {}

    def f32_variant(self):
        return {}

    def other_config_variant(self):
        return dict(label="stand-in", batch_size=32)

    def fresh_batch_stepper(self):
        return self.train_step


def run_legs(args, world, rank, w):
    """The run, identical on every rank.  Returns the JSON line's record (used on rank 0)."""
    comm, share = w.comm, w.share
    # (An earlier build launched kernels 10x slower for the first ~8 s of a process started right behind another
    # GPU-heavy one, and idled here by default.  It went away with the vendor convolutions - the encoder's solver
    # search ran inside the first steps of every process.  The switch stays for diagnosis.)
    if args.settle_seconds > 0:
        note("settling pause of %.0f s" % args.settle_seconds)
        comm.sync()
        time.sleep(args.settle_seconds)
    note("model and batch ready; warm-up")
    for _ in range(args.warmup):
        w.train_step()
    note("timing %d training steps" % args.steps)
    comm.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = w.train_step()
    comm.fence()
    elapsed = comm.max(time.perf_counter() - t0)
    ranks_seen = int(round(comm.sum(1.0)))
    seqs = share["batch"] * world * args.steps

    def time_steps(step, n, warm):
        """(seconds per step, MAX over ranks) of n calls of ``step`` after ``warm`` untimed ones, fenced like the
        headline region.  Every rank calls this with the same n and warm."""
        for _ in range(warm):
            step()
        comm.fence()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        comm.fence()
        return comm.max(time.perf_counter() - t0) / n

    note("training leg done: %.3f ms/step; rasteriser leg (with its CPU baseline unless disabled)" %
         (elapsed / args.steps * 1e3))
    # ---- rasteriser / IoU leg (sharded rollouts, no collective inside) ---------------------------------------
    raster = w.raster_leg(do_cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
    if world > 1:
        raster["ms_per_launch"] = comm.max(raster["ms_per_launch"])
        raster["rollouts"] = share["rollouts"] * world
        raster["rollouts_per_s"] = raster["rollouts"] / raster["ms_per_launch"] * 1e3

    rl = None
    if args.rl_envs > 0:
        note("REINFORCE leg")
        rl = w.reinforce_leg()
        if world > 1:
            rl["ms_per_step"] = comm.max(rl["ms_per_step"])
            rl["envs"] = share["rl_envs"] * world
            rl["rollouts_per_s_with_update"] = rl["envs"] / rl["ms_per_step"] * 1e3
            rl.pop("rollouts_per_s_rollout_only")

    ac = None
    if args.a3c_envs > 0:
        note("actor-critic leg")
        ac = w.a3c_leg()
        if world > 1:
            ac["env_steps_per_s"] = comm.sum(ac["env_steps_per_s"])
            ac["envs"] = share["a3c_envs"] * world

    kernels, roofline, cpu = None, None, None
    if not args.no_kernel_timing:
        note("per-kernel timing")
        in_step = w.timeline_steps(2)  # training steps: on every rank
        if os.environ.get("ISG_BENCH_INJECT") == "rank0_step" and rank == 0:
            w.train_step()  # fault injection for the launch check's own test: a step only rank 0 takes
        if rank == 0:
            kernels, roofline = w.kernel_rooflines(in_step)  # isolated launches, no collective
            if kernels is not None:
                kernels = kernels + w.encoder_rooflines(in_step)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("CPU baseline of the training step (bounded sample)")
        cpu = w.cpu_baseline()

    extra = None
    if not args.no_extra_runs:
        w.release()
        extra = {}
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        note("extra leg: every sequence at full length (durations = T)")
        full = w.full_length_variant()
        # (this leg follows a dozen seconds of CPU-only work: a few more warm-up steps bring the clocks back up)
        sec = time_steps(lambda: w.train_step(**full), max(args.steps // 2, 3), 6)
        extra["all_durations_T"] = {"value": share["batch"] * world / sec, "unit": "sequences/s", "ms_per_step": sec * 1e3,
                                    "dtype": args.dtype, "convlstm_mfma_frac_of_step":
                                        w.conv_flops_per_sequence(durations=full.get("durations")) * share["batch"] /
                                        sec / 1e12 / peak}
        note("extra leg: a fresh batch from the device-resident dataset in every step")
        fresh = w.fresh_batch_stepper()
        sec = time_steps(fresh, max(args.steps // 2, 3), 2)
        extra["fresh_batches"] = {"value": share["batch"] * world / sec, "unit": "sequences/s", "ms_per_step": sec * 1e3,
                                  "dtype": args.dtype, "batch_source": "data.DeviceBatchBuilder (one gather + one "
                                  "isg_batch_points launch per step; lengths stay on the device, all T steps run masked)"}
        del fresh
        if args.dtype == "bf16":
            note("extra leg: the same step in f32 (f32 MFMA, three-kernel backward)")
            find = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = False  # no solver search for a three-step leg
            v32 = w.f32_variant()
            sec = time_steps(lambda: w.train_step(**v32), 3, 1)
            extra["f32"] = {"value": share["batch"] * world / sec, "unit": "sequences/s", "ms_per_step": sec * 1e3,
                            "dtype": "f32", "convlstm_mfma_frac_of_step":
                                w.conv_flops_per_sequence(model=v32.get("model")) * share["batch"] / sec / 1e12 /
                                MFMA_PEAK_TFLOPS["f32"]}
            del v32
            torch.backends.cudnn.benchmark = find
            w.release()
        other = w.other_config_variant()
        if other is not None:
            note("extra leg: %s" % other["label"])
            label, bsz = other.pop("label"), other.pop("batch_size")
            sec = time_steps(lambda: w.train_step(**other), 10, 3)
            extra["endtokendeep2"] = {"value": bsz * world / sec, "unit": "sequences/s", "ms_per_step": sec * 1e3,
                                      "dtype": args.dtype, "workload": label, "global_batch": bsz * world,
                                      "step_as_graph": bool(getattr(other.get("model"), "step_graphs", False))}
            del other
            w.release()
        if rank == 0:
            note("extra leg: autoregressive decode latency")
            extra["decode"] = w.decode_leg()  # no collective
            if args.config == "rnnconv" and args.dtype == "bf16":
                note("extra leg: the five-layer recurrence of EndTokenDeep_2 alone at 224 x 224")
                w.release()
                extra["recurrence_224_L5"] = w.recurrence_leg()  # no collective
    comm.fence()  # nobody tears the process group down while another rank is still inside a leg

    return {
        "metric": "polygon seqs/sec fwd+bwd (ConvLSTM, 224²) & IoU rollouts/sec",
        "value": seqs / elapsed, "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": dict(w.describe(), global_batch=share["batch"] * world, parallelism="dp%d" % world),
        "ranks_seen": ranks_seen, "allreduce_bytes": w.allreduce_bytes(),
        "convlstm_mfma_frac_of_step": w.conv_flops_per_sequence() * share["batch"] * world / (elapsed / args.steps) /
        1e12 / (MFMA_PEAK_TFLOPS[args.dtype] * world),
        "loss": float(loss),
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "extra_runs": extra,
        "iou": raster, "reinforce": rl, "a3c": ac,
    }


def main():
    args = parse()
    if os.environ.get("ISG_BENCH_WATCHDOG"):  # dump every thread's stack and exit if the run takes this long
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ISG_BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match the %d ranks of the launcher" % (args.gpus, world))
    w = (StubWorkload if args.launch_check else GpuWorkload)(args, world, rank, local_rank)
    if args.kernels_only:
        w.kernels_only()
        return w.close()
    line = run_legs(args, world, rank, w)
    if args.launch_check:
        line.update(metric="launch check (no GPU work)", launch_check=True)
    w.close()  # (the launch check compares the ranks' collective counts here, before anything is printed)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
