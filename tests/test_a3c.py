"""Actor-critic pieces: returns/advantages on CPU against scipy.signal.lfilter (the call the reference
makes, polyrl/a3c/a3c.py:11-32); policy, loss and gradients on the GPU against the float64 CPU
restatement (oracle/a3c.py).  The reference's a3c code never ran and has no fixtures: the network
and loss checks are parity unpinned."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import a3c as oa  # noqa: E402


def test_discount_and_advantages_match_lfilter():
    from imagesegmentation_amd.polyrl import a3c
    rs = np.random.RandomState(0)
    for n in (1, 2, 10):
        x = rs.randn(n)
        assert np.allclose(a3c.discount(x, 0.99), oa.discount(x, 0.99), rtol=0, atol=1e-12)
    rewards, values = rs.rand(7), rs.randn(7)
    for boot in (0.0, 0.37):
        adv, ret = a3c.process_rollout(rewards, values, boot, 0.99, 1.0)
        radv, rret = oa.process_rollout(rewards, values, boot, 0.99, 1.0)
        assert np.allclose(adv, radv, atol=1e-12) and np.allclose(ret, rret, atol=1e-12)


def test_batched_advantages_match_per_rollout():
    from imagesegmentation_amd.polyrl import a3c
    rs = np.random.RandomState(1)
    N, T = 6, 10
    rewards, values, boot = rs.rand(N, T), rs.randn(N, T), rs.rand(N)
    lengths = np.array([10, 3, 1, 0, 7, 10], np.int32)
    for b in (None, boot):
        adv, ret = a3c.process_rollouts(torch.tensor(rewards), torch.tensor(values), torch.tensor(lengths), 0.99, 1.0,
                                        None if b is None else torch.tensor(b))
        for n in range(N):
            l = int(lengths[n])
            assert float(adv[n, l:].abs().sum()) == 0.0 and float(ret[n, l:].abs().sum()) == 0.0
            if l == 0:
                continue
            radv, rret = oa.process_rollout(rewards[n, :l], values[n, :l], 0.0 if b is None else b[n], 0.99, 1.0)
            assert np.allclose(adv[n, :l].numpy(), radv, atol=1e-12)
            assert np.allclose(ret[n, :l].numpy(), rret, atol=1e-12)


def test_loss_matches_restatement_on_cpu_tensors():
    from imagesegmentation_amd.polyrl import a3c
    g = torch.Generator().manual_seed(0)
    T, K = 5, 17
    logits = torch.randn(T, K, generator=g, dtype=torch.float64)
    vf, adv, r = (torch.randn(T, generator=g, dtype=torch.float64) for _ in range(3))
    actions = torch.randint(0, K, [T], generator=g)
    got, parts = a3c.a3c_loss(logits, vf, actions, adv, r)
    assert abs(float(got) - float(oa.loss(logits, vf, actions, adv, r))) < 1e-4
    # masked rows drop out
    live = torch.tensor([True, True, False, True, False])
    got2, _ = a3c.a3c_loss(logits, vf, actions, adv, r, live)
    ref2 = oa.loss(logits[live], vf[live], actions[live], adv[live], r[live])
    assert abs(float(got2) - float(ref2)) < 1e-4


def _policy_and_port(dtype, P, hist, seed):
    from imagesegmentation_amd.polyrl import a3c
    pi = a3c.LSTMPolicy((P, P, 3 + hist), P * P, compute_dtype=dtype, device="cuda:0", seed=seed)
    with torch.no_grad():  # biases away from zero so that their gradients and effects are exercised
        g = torch.Generator(device="cuda:0").manual_seed(seed + 1)
        for v in pi.var_list:
            if v.dim() == 1:
                v.copy_(torch.randn(v.shape, device="cuda:0", generator=g) * 0.1)
    d = lambda t: t.detach().double().cpu().requires_grad_(True)
    tp = dict(convs=[(d(w), d(b)) for w, b in pi.convs], w_x=d(pi.lstm.input_to_state_kernel),
              w_h=d(pi.lstm.hidden_to_state_kernel), bias=d(pi.lstm.bias),
              action=(d(pi.action_kernel), d(pi.action_bias)), value=(d(pi.value_kernel), d(pi.value_bias)))
    return pi, tp


@pytest.mark.gpu
def test_policy_forward_loss_and_gradients_against_cpu_restatement():
    from imagesegmentation_amd.polyrl import a3c
    P, hist, N, T = 16, 2, 3, 4
    pi, tp = _policy_and_port(torch.float32, P, hist, seed=3)
    assert pi.lstm._filters == 32 and pi.state_size.c == torch.Size([P, P, 32])
    g = torch.Generator().manual_seed(5)
    x = torch.rand(N, T, P, P, 3 + hist, generator=g, dtype=torch.float64)
    actions = torch.randint(0, P * P, [N, T], generator=g)
    adv, r = torch.randn(N, T, generator=g, dtype=torch.float64), torch.rand(N, T, generator=g, dtype=torch.float64)
    lengths = torch.tensor([4, 2, 3])
    live = torch.arange(T)[None, :] < lengths[:, None]
    dev = pi.device
    logits, vf, (c, h) = pi.forward(x.float().to(dev), pi.get_initial_features(N))
    z = torch.zeros(N, P, P, 32, dtype=torch.float64)
    rl, rv, (rc, rh) = oa.policy_forward(tp, x, z, z)
    tol = lambda ref: 1e-4 * float(ref.abs().max())
    assert float((logits.double().cpu() - rl).abs().max()) <= tol(rl)
    assert float((vf.double().cpu() - rv).abs().max()) <= tol(rv)
    assert float((c.double().cpu() - rc).abs().max()) <= tol(rc)
    loss, _ = a3c.a3c_loss(logits.reshape(N * T, -1), vf.reshape(-1), actions.reshape(-1).to(dev),
                           adv.reshape(-1).to(dev), r.reshape(-1).to(dev), live.reshape(-1).to(dev))
    ref = sum(oa.loss(rl[n, :int(lengths[n])], rv[n, :int(lengths[n])], actions[n, :int(lengths[n])],
                      adv[n, :int(lengths[n])], r[n, :int(lengths[n])]) for n in range(N))
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref))
    opt = a3c.FlatAdam(pi.var_list, clip_norm=40.0)
    loss.backward()
    ref.backward()
    ref_grads = [t for pair in tp["convs"] for t in pair] + [tp["w_h"], tp["w_x"], tp["bias"], tp["action"][0],
                                                              tp["action"][1], tp["value"][0], tp["value"][1]]
    for (name, v), rt in zip(pi.named_variables(), ref_grads):
        err = float((v.grad.double().cpu() - rt.grad).abs().max())
        assert err <= 2e-4 * float(rt.grad.abs().max()) + 1e-9, (name, err)
    assert opt.flat_grad.abs().sum() > 0  # gradients landed in the flat bucket


@pytest.mark.gpu
@pytest.mark.parametrize("end_token", [False, True])
def test_synchronous_update_runs_and_learns_value(end_token):
    """A few updates on a fixed batch of environments: rollouts obey the step limit and the
    masking, rewards are IoUs in [0, 1], the update changes the parameters and keeps them finite."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv
    ds = data.PolygonDataset.synthetic(12, image_size=28, prediction_size=28, max_timesteps=10, history_length=2,
                                       seed=2, end_token=end_token)
    env = BatchedPolygonEnv(ds, num_envs=8, end_token=end_token, max_steps=a3c.A3C.TIMESTEP_LIMIT, device="cuda:0")
    trainer = a3c.A3C(env, seed=1)
    before = [v.detach().clone() for v in trainer.network.var_list]
    for _ in range(2):
        batch = trainer.process()
        assert batch["si"].shape[:2] == (8, 10) and int(batch["lengths"].max()) <= 10
        assert float(batch["rewards"].min()) >= 0.0 and float(batch["rewards"].max()) <= 1.0
        assert np.isfinite(trainer.last["loss"]) and trainer.last["grad_global_norm"] > 0
    assert trainer.local_steps == 2 and trainer.global_step > 0
    moved = sum(float((a - b).abs().sum()) for a, b in zip(before, trainer.network.var_list))
    assert moved > 0 and all(bool(torch.isfinite(v).all()) for v in trainer.network.var_list)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stepper_acting_equals_stateless_acting(dtype):
    """``act_step`` (recurrent state kept split by filter group on the device) against ``act`` (state passed in and
    out as whole tensors) over four steps: same actions from the same sampling stream, same values, same state."""
    from imagesegmentation_amd.polyrl import a3c
    dev = torch.device("cuda:0")
    N, P = 5, 12
    a = a3c.LSTMPolicy((P, P, 5), P * P, compute_dtype=dtype, device=dev, seed=4)
    b = a3c.LSTMPolicy((P, P, 5), P * P, compute_dtype=dtype, device=dev, seed=4)
    with torch.no_grad():
        for pa, pb in zip(a.var_list, b.var_list):
            pb.copy_(pa)
    gen = torch.Generator(device=dev).manual_seed(9)
    obs = [torch.rand([N, P, P, 5], device=dev, generator=gen) for _ in range(4)]
    c, h = a.get_initial_features(N)
    rec = b.stepper(N)
    for ob in obs:
        act_a, v_a, c, h = a.act(ob, c, h)
        act_b, v_b = b.act_step(ob, rec)
        assert torch.equal(act_a, act_b)
        assert torch.equal(v_a, v_b)
    st = rec.state
    assert torch.equal(st.c, c.to(st.c.dtype)) and torch.equal(st.h, h.to(st.h.dtype))


@pytest.mark.gpu
def test_rollout_replayed_as_a_graph_equals_the_rollout_launched_step_by_step(monkeypatch):
    """A3C.rollout_graph: the acting loop of the 32-filter policy (encoder, cell, heads, sampling with a device-side
    stream position, environment step, terminal IoU, T times) captured as a HIP graph against the same loop launched
    kernel by kernel: observations, actions, values, rewards and lengths of every rollout agree bit for bit over
    several updates (the weights change in between: the graph reads the refreshed weight images), and so do the
    parameters after them."""
    import numpy as np
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv
    dev = torch.device("cuda:0")
    n = 32
    ds = data.PolygonDataset.synthetic(16, image_size=32, prediction_size=32, max_timesteps=10, history_length=2, seed=30)
    rs = np.random.RandomState(40)
    idx = rs.randint(0, len(ds), [n])
    images = ds._images[idx]
    polygons = [np.roll(ds._vertices[i], rs.randint(len(ds._vertices[i])), axis=0) for i in idx]
    runs = {}
    for graph in (False, True):
        env = BatchedPolygonEnv(ds, n, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=dev)
        trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0, learning_rate=1e-3)
        trainer.rollout_graph = graph
        seen = []
        for _ in range(4):
            b = trainer.process(images, polygons)
            seen.append({k: b[k].clone() for k in ("si", "a", "values", "rewards", "lengths")})
        assert (trainer._rollout["graph"] is not None) == graph
        runs[graph] = (seen, trainer.opt.flat_param.clone())
    for a, b in zip(runs[False][0], runs[True][0]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(runs[False][1], runs[True][1])
    # successive rollouts draw different actions (the stream position lives on the device and moves with each step)
    assert not torch.equal(runs[True][0][2]["a"], runs[True][0][3]["a"])


@pytest.mark.gpu
def test_update_that_adopts_the_rollouts_recurrence_equals_the_update_that_runs_its_own():
    """A3C.reuse_forward: the rollout keeps its pass of the 32-filter recurrence (WideRecorder, also inside the
    replayed graph) and the update's backward sweep reads it, against the update that runs the forward steps again.
    The adopted pass holds the very numbers a second pass computes: the update's logits and values are the same bit
    for bit, also after the rollout has become a graph replay.  The adopted outputs are stored time-major and the
    loss takes its rows in that order, so the weight gradients add the same rows in another order: the parameters
    after an update agree to float rounding."""
    import numpy as np
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv
    dev = torch.device("cuda:0")
    n = 32
    ds = data.PolygonDataset.synthetic(16, image_size=32, prediction_size=32, max_timesteps=10, history_length=2, seed=30)
    rs = np.random.RandomState(40)
    idx = rs.randint(0, len(ds), [n])
    images = ds._images[idx]
    polygons = [np.roll(ds._vertices[i], rs.randint(len(ds._vertices[i])), axis=0) for i in idx]
    runs = {}
    for reuse in (True, False):
        env = BatchedPolygonEnv(ds, n, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=dev)
        trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0, learning_rate=1e-3)
        assert trainer.reuse_forward and trainer.reuse_encoder
        trainer.reuse_forward, trainer.reuse_encoder = reuse, False
        start = trainer.opt.flat_param.clone()
        b = trainer.process(images, polygons)
        runs[reuse] = ({k: b[k].clone() for k in ("si", "a", "values", "rewards", "lengths")},
                       trainer.opt.flat_param.clone(), trainer, dict(trainer.last))
    for k in runs[True][0]:
        assert torch.equal(runs[True][0][k], runs[False][0][k]), k
    assert abs(runs[True][3]["loss"] - runs[False][3]["loss"]) <= 1e-6 * abs(runs[False][3]["loss"])
    assert float((runs[True][1] - runs[False][1]).abs().max()) <= 2e-4
    assert float((runs[False][1] - start).abs().max()) > 1e-4
    trainer = runs[True][2]
    pi = trainer.local_network
    for _ in range(3):  # (the second rollout is captured, later ones are replays)
        trainer.process(images, polygons)
    assert trainer._rollout["graph"] is not None
    from imagesegmentation_amd.convLSTM import LSTMStateTuple, SequenceInput, dynamic_rnn
    with trainer.derived:
        batch = trainer.pull_batch(images, polygons)
        si, T = batch["si"], batch["si"].shape[1]
        feats = pi._encode(si.reshape((n * T,) + tuple(si.shape[2:]))).reshape((n, T) + tuple(si.shape[2:4]) + (32,))
        outs = []
        for recorded in (batch["forward"], None):
            seq = SequenceInput(x_seq=feats.transpose(0, 1))
            seq.recorded = recorded
            outs.append(dynamic_rnn(pi.lstm, seq, initial_state=LSTMStateTuple(*pi.get_initial_features(n)),
                                    time_major=False)[0])
        adopted = pi.forward(si, pi.get_initial_features(n), recorded=batch["forward"])
        own = pi.forward(si, pi.get_initial_features(n))
    assert outs[0].transpose(0, 1).is_contiguous() and outs[1].is_contiguous()
    assert torch.equal(outs[0], outs[1])
    # (the heads' library GEMMs split their reduction by the position of a row's tile: equal to bf16 rounding)
    assert adopted[0].shape == own[0].shape and adopted[0].transpose(0, 1).is_contiguous()
    assert float((adopted[0] - own[0]).abs().max()) <= 4e-3 and float((adopted[1] - own[1]).abs().max()) <= 4e-3
    # the encoder's pass adopted as well: its weight gradients sum the same rows step by step instead of environment
    # by environment - the first rollout is the same, the parameters after one update agree to float rounding
    first = {}
    for enc in (True, False):
        env = BatchedPolygonEnv(ds, n, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=dev)
        trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0, learning_rate=1e-3)
        trainer.reuse_encoder = enc
        b = trainer.process(images, polygons)
        first[enc] = ({k: b[k].clone() for k in ("si", "a", "values", "rewards", "lengths")},
                      trainer.opt.flat_param.clone(), dict(trainer.last))
    for k in first[True][0]:
        assert torch.equal(first[True][0][k], first[False][0][k]), k
    assert abs(first[True][2]["loss"] - first[False][2]["loss"]) <= 1e-6 * abs(first[False][2]["loss"])
    assert abs(first[True][2]["grad_global_norm"] - first[False][2]["grad_global_norm"]) <= \
        1e-3 * first[False][2]["grad_global_norm"]
    assert float((first[True][1] - first[False][1]).abs().max()) <= 2e-4
    assert float((first[False][1] - start).abs().max()) > 1e-4


@pytest.mark.gpu
def test_updates_on_episodes_drawn_on_the_device():
    """A3C.process(device_batch=DevicePolygonPool.sample(n)): every update starts fresh episodes without host work;
    the replayed rollout graph keeps reading the environment's tensors in place across the resets (the observations
    of the first step are the pool's images with the first history frames drawn in)."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv, DevicePolygonPool
    dev = torch.device("cuda:0")
    n = 32
    ds = data.PolygonDataset.synthetic(16, image_size=32, prediction_size=32, max_timesteps=10, history_length=2, seed=30)
    env = BatchedPolygonEnv(ds, n, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=dev)
    trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0, learning_rate=1e-3)
    pool = DevicePolygonPool(ds, dev, dtype=torch.float32, vcap=2 + a3c.A3C.TIMESTEP_LIMIT + 1, seed=7)
    firsts = []
    for _ in range(4):
        episodes = pool.sample(n)
        b = trainer.process(device_batch=episodes)
        assert torch.equal(b["si"][:, 0, ..., :3], episodes[0])       # the rollout saw THESE images
        assert int(b["lengths"].min()) >= 1 and torch.isfinite(b["values"]).all()
        firsts.append(b["si"][:, 0].clone())
    assert trainer._rollout["graph"] is not None
    assert not torch.equal(firsts[2], firsts[3])                      # a fresh draw per update
    assert trainer.last["loss"] == trainer.last["loss"]               # (not NaN)


@pytest.mark.gpu
def test_advantages_in_one_launch_against_the_reference_scan():
    """isg_gae (process_rollouts on device tensors in the dtypes a rollout holds) against the reference's per-rollout
    scan restated in numpy, with and without a bootstrap value, lengths 0 .. T."""
    import numpy as np
    from imagesegmentation_amd.polyrl import a3c
    rs = np.random.RandomState(3)
    N, T = 300, 10
    rewards, values, boot = rs.rand(N, T), rs.randn(N, T).astype(np.float32), rs.rand(N)
    lengths = rs.randint(0, T + 1, [N]).astype(np.int32)
    lengths[:3] = (0, 1, T)
    d = "cuda:0"
    for b in (None, boot):
        adv, ret = a3c.process_rollouts(torch.tensor(rewards, device=d), torch.tensor(values, device=d),
                                        torch.tensor(lengths, device=d), 0.99, 0.95,
                                        None if b is None else torch.tensor(b, device=d))
        assert adv.dtype == torch.float64 and ret.dtype == torch.float64
        adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
        for n in range(N):
            l = int(lengths[n])
            assert not adv[n, l:].any() and not ret[n, l:].any()
            if l:
                radv, rret = oa.process_rollout(rewards[n, :l], values[n, :l].astype(np.float64),
                                                0.0 if b is None else b[n], 0.99, 0.95)
                assert np.allclose(adv[n, :l], radv, rtol=0, atol=1e-12) and np.allclose(ret[n, :l], rret, rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_loss_against_the_tensor_form(dtype, monkeypatch):
    """isg_a3c_loss_fwd / _bwd (a3c_loss on device tensors) against the same loss written with tensor operations in
    float64: the three sums and the gradients of the logits and the values, with masked rows and an action column at
    either end."""
    from imagesegmentation_amd.polyrl import a3c
    d = "cuda:0"
    g = torch.Generator(device=d).manual_seed(4)
    R, K = 777, 1024
    logits = (torch.randn([R, K], device=d, generator=g) * 2).to(dtype).requires_grad_(True)
    vf = torch.randn([R], device=d, generator=g).requires_grad_(True)
    actions = torch.randint(0, K, [R], device=d, generator=g)
    actions[0], actions[1] = 0, K - 1
    adv = torch.randn([R], device=d, generator=g, dtype=torch.float64)
    ret = torch.randn([R], device=d, generator=g, dtype=torch.float64)
    live = torch.rand([R], device=d, generator=g) > 0.3
    loss, parts = a3c.a3c_loss(logits, vf, actions, adv, ret, live)
    (loss / 7).backward()
    z64 = logits.detach().double().requires_grad_(True)
    v64 = vf.detach().double().requires_grad_(True)
    monkeypatch.setenv("ISG_NO_FUSED_A3C_LOSS", "1")
    ref, rparts = a3c.a3c_loss(z64, v64, actions, adv, ret, live)
    (ref / 7).backward()
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-5  # (bf16 gradients: one rounding, 2^-8 of the value)
    for k in ("pi_loss", "vf_loss", "entropy"):
        assert abs(float(parts[k]) - float(rparts[k])) <= 1e-4 * abs(float(rparts[k])) + 1e-3, k
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-3
    assert float((logits.grad.double() - z64.grad).abs().max()) <= tol * float(z64.grad.abs().max())
    assert float((vf.grad.double() - v64.grad).abs().max()) <= 1e-6
    assert float(logits.grad[~live].abs().max()) == 0.0 and float(vf.grad[~live].abs().max()) == 0.0


@pytest.mark.gpu
def test_joint_backward_of_the_two_heads_against_the_separate_ones(monkeypatch):
    """LSTMPolicy._heads with grad enabled: the action and value heads as one node whose backward runs one layer of
    A + 1 columns (_JointHeads) against the two separate dense-head nodes (ISG_NO_JOINT_HEADS): the same logits and
    values bit for bit (the forward GEMMs are the same), the same gradients up to the bf16 rounding of GEMMs whose
    reductions split differently, and both against float64."""
    from imagesegmentation_amd.polyrl import a3c
    dev = torch.device("cuda:0")
    P = 16
    pol = a3c.LSTMPolicy((P, P, 5), P * P, compute_dtype=torch.bfloat16, device=dev, seed=3)
    g = torch.Generator(device=dev).manual_seed(4)
    R, K = 96, P * P * 32
    flat0 = (torch.randn([R, K], device=dev, generator=g) * 0.5).bfloat16()
    up_l = torch.randn([R, P * P], device=dev, generator=g)
    up_v = torch.randn([R], device=dev, generator=g)
    heads = [pol.action_kernel, pol.action_bias, pol.value_kernel, pol.value_bias]
    res = {}
    for joint in (True, False):
        if not joint:
            monkeypatch.setenv("ISG_NO_JOINT_HEADS", "1")
        for p in heads:
            p.grad = None
        flat = flat0.clone().requires_grad_(True)
        logits, vf = pol._heads(flat)
        assert logits.dtype == torch.float32 and tuple(logits.shape) == (R, P * P) and tuple(vf.shape) == (R,)
        ((logits * up_l).sum() + (vf * up_v).sum()).backward()
        res[joint] = [logits.detach().clone(), vf.detach().clone(), flat.grad.float()] + [p.grad.clone() for p in heads]
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2:], res[False][2:]):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2.0 ** -6 * float(b.abs().max())
    # float64 on the operands the GEMMs see (bf16 weights and upstream gradients)
    bf = lambda t: t.detach().bfloat16().double()
    d = torch.cat([bf(up_l), bf(up_v)[:, None]], dim=1)
    wj = torch.cat([bf(pol.action_kernel), bf(pol.value_kernel)], dim=1)
    ref_flat = d @ wj.t()
    ref_w = flat0.double().t() @ d
    assert float((res[True][2].double() - ref_flat).abs().max()) <= 2.0 ** -7 * float(ref_flat.abs().max())
    assert float((res[True][3].double() - ref_w[:, :-1]).abs().max()) <= 2.0 ** -7 * float(ref_w.abs().max())
    assert float((res[True][5].double() - ref_w[:, -1:]).abs().max()) <= 2.0 ** -7 * float(ref_w.abs().max())
    assert float((res[True][4].double() - d[:, :-1].sum(0)).abs().max()) <= 1e-4 * float(d.abs().sum(0).max())
    assert float((res[True][6].double() - d[:, -1:].sum(0)).abs().max()) <= 1e-4 * float(d.abs().sum(0).max())
