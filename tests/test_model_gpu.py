"""Whole-model parity: logits, loss, gradients, one optimiser step, metrics and the decode/IoU loop
against the float64 oracle, for the reference's experiment graphs at 28 x 28."""
import numpy as np
import pytest

from oracle import convlstm as oc
from oracle import raster as oracle_raster
from helpers_convlstm import make_batch, make_params

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _load(model, params):
    with torch.no_grad():
        for (w, b), p in zip(model.encoder, params["encoder"]):
            w.copy_(torch.tensor(p["w"]))
            b.copy_(torch.tensor(p["b"]))
        for cell, p in zip(model.rnn.cells, params["rnn"]):
            cell.input_to_state_kernel.copy_(torch.tensor(p["w_x"]))
            cell.hidden_to_state_kernel.copy_(torch.tensor(p["w_h"]))
            cell.bias.copy_(torch.tensor(p["bias"]))
        model.dense_kernel.copy_(torch.tensor(params["dense"]["w"]))
        model.dense_bias.copy_(torch.tensor(params["dense"]["b"]))


def _flat_oracle_grads(grads):
    out = []
    for g in grads["encoder"]:
        out += [g["w"], g["b"]]
    for g in grads["rnn"]:
        out += [g["w_h"], g["w_x"], g["bias"]]
    out += [grads["dense"]["w"], grads["dense"]["b"]]
    return out


def _flat_oracle_params(params):
    out = []
    for p in params["encoder"]:
        out += [p["w"], p["b"]]
    for p in params["rnn"]:
        out += [p["w_h"], p["w_x"], p["bias"]]
    out += [params["dense"]["w"], params["dense"]["b"]]
    return out


def rel_err(got, ref):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


# measured worst over the four nets: 0.077 - 0.111 and 0.9941 - 0.9971 (the deepest tensors of each net)
BF16_GRAD_L2, BF16_GRAD_COS = 1.5e-1, 0.985

# (the reference runs every graph at 28 x 28; the History0 plumbing configuration is named at 32 x 32, where a row of
# the grid is exactly one 32-pixel tile wide instead of a partial one)
CASES = [("History2Large", dict()), ("History0", dict()), ("RNNConv", dict()), ("EndTokenDeep_2", dict()),
         ("History1Large", dict()), ("History0", dict(P=32, B=4))]


@pytest.mark.parametrize("name,kw", CASES)
def test_fp32_logits_loss_grads_and_adam_step(name, kw):
    from imagesegmentation_amd.polyrnn import experiments
    P, B = kw.get("P", 28), kw.get("B", 3)
    rs = np.random.RandomState(len(name))
    model = experiments.make(name, image_size=P, compute_dtype=torch.float32, device="cuda:0")
    hist, T = model.history_length, model.max_timesteps
    params = make_params(rs, P, hist, encoder=model.encoder_spec, layers=model.rnn_layers, units=model.num_classes)
    _load(model, params)
    images, histories, durations, targets = make_batch(rs, B, T, P, hist, end_token=model.end_token)
    durations[1] = min(durations[1], 2)
    loss_ref, logits_ref, grads_ref = oc.model_loss_and_grads(params, images, histories, durations, targets, P)

    model._flat_grad.zero_()
    logits, _ = model.forward(images, durations, histories)
    loss, metrics = model._loss_and_metrics(logits, durations, targets)
    loss.backward()
    # north-star bar: logits within 1e-4 relative in fp32
    assert rel_err(logits, logits_ref) < 1e-4
    assert abs(loss.item() - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    for p, g in zip(model.trainable_variables, _flat_oracle_grads(grads_ref)):
        assert rel_err(p.grad, g) < 2e-4, (tuple(p.shape), rel_err(p.grad, g))
    acc, avg, mx = oc.metrics(logits_ref, targets, durations, P)
    assert abs(metrics["accuracy"].item() - acc) < 1e-12
    assert abs(metrics["avg_error"].item() - avg) < 1e-9 and abs(metrics["max_error"].item() - mx) < 1e-9

    # clip_by_global_norm(1.0) + TensorFlow-form Adam, one step
    clipped, _ = oc.clip_by_global_norm(_flat_oracle_grads(grads_ref), 1.0)
    model._apply_gradients()
    for p, p0, g in zip(model.trainable_variables, _flat_oracle_params(params), clipped):
        want, _, _ = oc.adam_step(p0, g, np.zeros_like(p0), np.zeros_like(p0), 1)
        step = np.abs(want - p0).max()  # = lr for any gradient well above Adam's epsilon
        # elements whose gradient is within a few epsilon (1e-8 / sqrt(1 - beta2)) of zero move by less
        # than lr, in proportion to |g|: there fp32 gradient rounding shows, hence 1 % of the step
        assert np.abs(p.detach().double().cpu().numpy() - want).max() < 1e-2 * step + 1e-9


@pytest.mark.parametrize("name,P", [("History2Large", 28), ("RNNConv", 28), ("EndTokenDeep_2", 28), ("History0", 32)])
def test_bf16_model_tracks_the_oracle(name, P):
    """Documented bf16 tolerance: logits within 5e-2 of the largest logit magnitude; loss within 2e-2."""
    from imagesegmentation_amd.polyrnn import experiments
    B = 4
    rs = np.random.RandomState(99)
    model = experiments.make(name, image_size=P, compute_dtype=torch.bfloat16, device="cuda:0")
    hist, T = model.history_length, model.max_timesteps
    params = make_params(rs, P, hist, encoder=model.encoder_spec, layers=model.rnn_layers, units=model.num_classes)
    _load(model, params)
    images, histories, durations, targets = make_batch(rs, B, T, P, hist, end_token=model.end_token)
    loss_ref, logits_ref, grads_ref = oc.model_loss_and_grads(params, images, histories, durations, targets, P)
    loss = model.train_step(images, durations, histories, targets)
    assert rel_err(model.prediction_logits.float(), logits_ref) < 5e-2
    assert abs(loss.item() - loss_ref) < 2e-2 * abs(loss_ref)
    # Gradients (the step leaves them in place).  bf16 activations, weights and carried state gradients through up
    # to 7 encoder layers, 5 recurrent layers and 10 steps: every variable's gradient keeps its direction (cosine
    # with the float64 gradient above BF16_GRAD_COS) and its size (relative L2 error below BF16_GRAD_L2); the
    # deepest tensors (first encoder kernels of the deep net) are the worst.
    worst_l2, worst_cos = 0.0, 1.0
    for p, g in zip(model.trainable_variables, _flat_oracle_grads(grads_ref)):
        got = p.grad.detach().double().cpu().numpy().ravel()
        ref = np.asarray(g, np.float64).ravel()
        l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
        cos = float(got @ ref) / max(np.linalg.norm(got) * np.linalg.norm(ref), 1e-30)
        worst_l2, worst_cos = max(worst_l2, l2), min(worst_cos, cos)
        assert l2 < BF16_GRAD_L2 and cos > BF16_GRAD_COS, (name, tuple(p.shape), l2, cos)
    print("bf16 gradients of %s: worst relative L2 error %.3e, worst cosine %.5f" % (name, worst_l2, worst_cos))


def test_points_and_dense_histories_agree_and_train_reduces_loss():
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrnn import experiments
    ds = data.PolygonDataset.synthetic(16, 28, 28, 10, 2, seed=1, shape_mask_fn=oracle_raster.create_shape_mask)
    model = experiments.make("History2Large", image_size=28, device="cuda:0", seed=3)
    np.random.seed(0)
    d, images, h, t, _ = ds.get_batch_for_rnn(8)
    np.random.seed(0)
    d2, images2, pts, t2, _ = ds.get_batch_points(8)
    with torch.no_grad():
        a, _ = model.forward(images, d, h)
        b, _ = model.forward(images2, d2, pts)
    # dense frames go through the GEMM, points through the tap scatter: same sum, different order
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    model.learning_rate = 1e-3
    first = model.train_step(images, d, pts, t).item()
    for _ in range(30):
        last = model.train_step(images, d, pts, t).item()
    assert last < first


def test_training_step_under_the_launch_timeline():
    """ops.timeline() (the benchmark's per-launch timing inside a real step): a bf16 training step runs under it,
    leaves the loss it leaves without it, and files every wrapped launch under its label."""
    from imagesegmentation_amd import data, ops
    from imagesegmentation_amd.polyrnn import experiments
    ds = data.PolygonDataset.synthetic(16, 28, 28, 10, 2, seed=1, shape_mask_fn=oracle_raster.create_shape_mask)
    np.random.seed(0)
    d, images, pts, t, _ = ds.get_batch_points(8)
    losses = []
    for timed in (False, True):
        model = experiments.make("History2Large", image_size=28, compute_dtype=torch.bfloat16, device="cuda:0", seed=3)
        model.train_step(images, d, pts, t)
        if timed:
            with ops.timeline() as tl:
                losses.append(model.train_step(images, d, pts, t).item())
            labels = tl.summary()
        else:
            losses.append(model.train_step(images, d, pts, t).item())
    assert losses[0] == losses[1]
    assert any(k.startswith("cell_fwd[") for k in labels) and any(k.startswith("bwd_fused[") for k in labels)
    assert any(k.startswith("relu_bwd_bias[") for k in labels) and all(v > 0 for v in labels.values())


@pytest.mark.parametrize("name", ["History2Large", "EndTokenDeep_2", "History0"])
def test_validate_iou_follows_the_reference_decode_loop(name):
    """The decode loop (one step per vertex, state fed back, np.roll history) against a NumPy
    restatement driven by the oracle's forward pass."""
    from imagesegmentation_amd.polyrnn import experiments
    P, B = 28, 3
    rs = np.random.RandomState(5)
    model = experiments.make(name, image_size=P, device="cuda:0")
    hist, T = model.history_length, model.max_timesteps
    params = make_params(rs, P, hist, encoder=model.encoder_spec, layers=model.rnn_layers, units=model.num_classes,
                         bias_scale=0.5)
    _load(model, params)
    polys = [rs.randint(0, P, [rs.randint(3, 7), 2]) for _ in range(B)]
    images = np.stack([np.stack([oracle_raster.create_shape_mask(p, P)] * 3, axis=2) for p in polys])
    got = model.validate_iou(images, polys, summary_prefix='valid')

    # reference loop (polyrnn/Model.py:218-260 and the per-experiment overrides), on the oracle
    predicted = np.empty([B, T, 2])
    histories = np.zeros([B, T, P, P, hist])
    for b, v in enumerate(polys):
        histories[b, 0] = oracle_raster.create_history(v, len(v) - 1, hist, P)
    state = None
    required = np.ones([B], np.uint32)
    durations = np.zeros([B], np.uint32)
    for i in range(T):
        logits, state = oc.model_forward(params, images, histories, required, init_state=state)
        out = oc.prediction_max(logits, P)
        predicted[:, i, :] = out[:, 0, :]
        durations += required
        for b in range(B):
            if model.end_token:
                if np.all(predicted[b, i, :] == np.array([P, 0])):
                    required[b] = 0
                    predicted[b, i, :] = polys[b][-hist]
                    out[b, 0, :] = 0
            elif np.all(predicted[b, i, :] == polys[b][-hist]):
                required[b] = 0
        if hist > 0:
            mask = np.zeros([B, P, P])
            for b in range(B):
                mask[b, out[b, 0, 1], out[b, 0, 0]] = 1
            histories = np.roll(histories, axis=4, shift=1)
            histories[:, 0, :, :, -1] = mask
    assert np.array_equal(model.last_predicted_vertices, predicted)
    ious = []
    for b in range(B):
        pv = predicted[b][:durations[b]] if name == "EndTokenDeep_2" else predicted[b]
        if name != "History0":
            pv = np.concatenate([polys[b][-hist:], pv], axis=0)
        ious.append(oracle_raster.calculate_iou(pv.astype(int), polys[b], P))
    assert np.array_equal(model.last_ious, np.array(ious))
    assert got == float(np.mean(ious))


@pytest.mark.parametrize("dtype,k", [(torch.float32, 4), (torch.bfloat16, 8), (torch.float32, 2)])
def test_avgpool_kernel(dtype, k):
    from imagesegmentation_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(k)
    x = torch.randn([3, 2, 8 * k, 4 * k, 16], device="cuda:0", generator=g).to(dtype)
    got = ops.avgpool(x, k)
    ref = x.float().reshape(3, 2, 8, k, 4, k, 16).mean(dim=(3, 5))
    assert got.shape == ref.shape and got.dtype == dtype
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert float((got.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


def test_pooled_head_matches_cpu_port_with_mean_pool():
    """head_size < prediction_size: the k x k mean between the recurrence and the dense head (forward
    kernel + gradient folded into the recurrence's backward) against the float64 CPU port, whose
    pool is torch's mean under autograd.  Ragged durations, 2 layers, history as dense frames."""
    from helpers_convlstm import params_to_torch, torch_model_logits
    from imagesegmentation_amd.polyrnn import experiments
    P, hs, B = 32, 8, 3
    rs = np.random.RandomState(42)
    model = experiments.make("History2Large", image_size=P, compute_dtype=torch.float32, device="cuda:0", head_size=hs)
    hist, T = model.history_length, model.max_timesteps
    params = make_params(rs, P, hist, encoder=model.encoder_spec, layers=model.rnn_layers, units=hs * hs)
    params["dense"]["w"] = rs.randn(hs * hs * 16, hs * hs) * 0.05
    _load(model, params)
    images, histories, durations, targets = make_batch(rs, B, T, P, hist)
    targets = targets // (P // hs)
    durations[0] = 1
    tp = params_to_torch(params)
    logits_ref, _ = torch_model_logits(tp, torch.tensor(images), torch.tensor(histories), torch.tensor(durations),
                                       head_pool=P // hs)
    dense = torch.tensor(targets[..., 0].astype(np.int64) * hs + targets[..., 1])
    loss_ref = torch.nn.functional.cross_entropy(logits_ref.reshape(B * T, -1), dense.reshape(-1))
    loss_ref.backward()
    model._flat_grad.zero_()
    logits, _ = model.forward(images, durations, histories)
    loss, _ = model._loss_and_metrics(logits, durations, targets)
    loss.backward()
    assert rel_err(logits, logits_ref.detach().numpy()) < 1e-4
    assert abs(loss.item() - float(loss_ref)) < 1e-5
    ref_grads = []
    for p in tp["encoder"]:
        ref_grads += [p["w"].grad, p["b"].grad]
    for p in tp["rnn"]:
        ref_grads += [p["w_h"].grad, p["w_x"].grad, p["bias"].grad]
    ref_grads += [tp["dense"]["w"].grad, tp["dense"]["b"].grad]
    for p, g in zip(model.trainable_variables, ref_grads):
        assert rel_err(p.grad, g.numpy()) < 2e-4, (tuple(p.shape), rel_err(p.grad, g.numpy()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [16, 32])
def test_fused_bias_relu_and_its_backward(dtype, C):
    from imagesegmentation_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(C)
    z = torch.randn([3, 20, 28, C], device="cuda:0", generator=g).to(dtype)
    bias = torch.randn([C], device="cuda:0", generator=g)
    ref = torch.relu(z.float() + bias).to(dtype)
    y = ops.bias_relu_(z.clone(), bias)
    assert torch.equal(y, ref)
    dy = torch.randn([3, 20, 28, C], device="cuda:0", generator=g).to(dtype)
    dz, dbias = ops.relu_bwd_bias(dy, y)
    ref_dz = torch.where(y > 0, dy, torch.zeros_like(dy))
    assert torch.equal(dz, ref_dz)
    ref_db = ref_dz.double().sum(dim=(0, 1, 2))
    assert float((dbias.double() - ref_db).abs().max()) <= 1e-5 * max(1.0, float(ref_db.abs().max()))


@pytest.mark.parametrize("KS,CO", [(14, 32), (5, 16), (5, 32), (14, 16)])
def test_first_encoder_convolution_kernels_against_float64_conv(KS, CO):
    """isg_conv_in_fwd / _wgrad (14 x 14 -> 32 of the benchmark net, 5 x 5 -> 16 of the reference's 28 x 28 nets;
    3 input channels, SAME, bias, ReLU; bf16) against the oracle's float64 convolution on bf16-rounded operands;
    ragged image size so that partial tiles are covered."""
    from imagesegmentation_amd import ops
    rs = np.random.RandomState(KS + CO)
    B, H, W = 2, 37, 45
    r = lambda a: torch.tensor(a).to(torch.bfloat16).double().numpy()
    x = r(rs.rand(B, H, W, 3))
    w = r(rs.randn(KS, KS, 3, CO) * 0.05)
    bias = rs.randn(CO) * 0.1
    dev = "cuda:0"
    tw = torch.tensor(w, dtype=torch.float32, device=dev)
    tb = torch.tensor(bias, dtype=torch.float32, device=dev)
    assert ops.conv_in_supported(tw, torch.bfloat16, tw.device)
    x4 = ops.conv_in_pad_image(torch.tensor(x, device=dev).to(torch.bfloat16))
    assert x4.shape == (B, H, W, 4) and float(x4[..., 3].abs().max()) == 0.0
    y = ops.conv_in_fwd(x4, ops.conv_in_pack(tw), tb, KS, relu=True)
    assert y.shape == (B, H, W, CO)
    ref = np.maximum(oc.conv2d_same(x, w, bias), 0.0)
    assert rel_err(y, ref) < 1e-2  # bf16 output rounding
    dz = r(rs.randn(B, H, W, CO))
    dw = ops.conv_in_wgrad(x4, torch.tensor(dz, device=dev).to(torch.bfloat16), KS)
    assert dw.shape == (KS, KS, 3, CO)
    _, ref_dw, _ = oc.conv2d_same_backward(x, w, dz)
    assert rel_err(dw, ref_dw) < 1e-4  # float accumulation of exact bf16 products


@pytest.mark.parametrize("C", [32, 16])
@pytest.mark.parametrize("KS", [3, 5])
def test_encoder_square_convolution_kernels_against_float64_conv(KS, C):
    """isg_conv32 / isg_conv16 (forward with bias + ReLU, data gradient) and their weight gradients against the
    oracle's float64 convolution on bf16-rounded operands; ragged image size so that partial tiles are covered."""
    from imagesegmentation_amd import ops
    rs = np.random.RandomState(KS + C)
    B, H, W = 2, 37, 45
    r = lambda a: torch.tensor(a).to(torch.bfloat16).double().numpy()
    x = r(rs.randn(B, H, W, C))
    w = r(rs.randn(KS, KS, C, C) * 0.05)
    bias = rs.randn(C) * 0.1
    dev = "cuda:0"
    bf = lambda a: torch.tensor(a, device=dev).to(torch.bfloat16)
    tw = torch.tensor(w, dtype=torch.float32, device=dev)
    tb = torch.tensor(bias, dtype=torch.float32, device=dev)
    assert ops.conv32_supported(tw, torch.bfloat16, tw.device)
    y = ops.conv32(bf(x), ops.conv32_pack(tw), tb, KS, relu=True)
    assert y.shape == (B, H, W, C)
    assert rel_err(y, np.maximum(oc.conv2d_same(x, w, bias), 0.0)) < 1e-2  # bf16 output rounding
    y_lin = ops.conv32(bf(x), ops.conv32_pack(tw), None, KS, relu=False)
    assert rel_err(y_lin, oc.conv2d_same(x, w)) < 1e-2
    dz = r(rs.randn(B, H, W, C))
    ref_dx, ref_dw, _ = oc.conv2d_same_backward(x, w, dz)
    dx = ops.conv32(bf(dz), ops.conv32_pack(tw, data_gradient=True), None, KS, relu=False)
    assert rel_err(dx, ref_dx) < 1e-2
    dw = ops.conv32_wgrad(bf(x), bf(dz), KS)
    assert dw.shape == (KS, KS, C, C)
    assert rel_err(dw, ref_dw) < 1e-4  # float accumulation of exact bf16 products


@pytest.mark.parametrize("KS", [3, 5])
def test_convolution_into_a_channel_slice_of_wider_pixels(KS):
    """isg_conv32_strided / isg_relu_bwd_bias_strided: the layer's output written as the first 32 channels of
    48-channel pixels (the joint input of a recurrent cell) equals the contiguous output bit for bit, leaves the
    other channels alone, and the ReLU backward reads it there with the same result."""
    from imagesegmentation_amd import ops
    rs = np.random.RandomState(KS)
    T, B, H, W, C = 2, 2, 19, 37, 32
    dev = "cuda:0"
    bf = lambda a: torch.tensor(a, device=dev).to(torch.bfloat16)
    tw = torch.tensor(rs.randn(KS, KS, C, C) * 0.05, dtype=torch.float32, device=dev)
    tb = torch.tensor(rs.randn(C) * 0.1, dtype=torch.float32, device=dev)
    x = bf(rs.randn(T * B, H, W, C))
    want = ops.conv32(x, ops.conv32_pack(tw), tb, KS, relu=True)
    wide = torch.full([T, B, H, W, 48], 7.0, dtype=torch.bfloat16, device=dev)
    slots = wide[..., :C]
    assert ops.pixel_stride(slots) == 48 and ops.pixel_stride(slots[1]) == 48 and ops.pixel_stride(wide[:, 0]) is None
    for t in range(T):
        got = ops.conv32(x[t * B:(t + 1) * B], ops.conv32_pack(tw), tb, KS, relu=True, out=slots[t])
        assert got.data_ptr() == slots[t].data_ptr()
    assert torch.equal(slots.reshape(T * B, H, W, C), want)
    assert bool((wide[..., C:] == 7.0).all())
    dy = bf(rs.randn(T * B, H, W, C))
    dz0, db0 = ops.relu_bwd_bias(dy.clone(), want)
    dz1, db1 = ops.relu_bwd_bias(dy.clone(), slots.view(T * B, H, W, C))
    assert torch.equal(dz0, dz1) and torch.equal(db0, db1)
    # ... and the gradient itself as such a slice (of the joint input's gradient)
    dwide = torch.full([T * B, H, W, 48], 3.0, dtype=torch.bfloat16, device=dev)
    dwide[..., :C] = dy
    dz2, db2 = ops.relu_bwd_bias(dwide[..., :C], slots.view(T * B, H, W, C))
    assert dz2.is_contiguous() and torch.equal(dz0, dz2) and torch.equal(db0, db2)
    assert bool((dwide[..., C:] == 3.0).all()) and torch.equal(dwide[..., :C], dy)


@pytest.mark.parametrize("C", [32, 16])
@pytest.mark.parametrize("KS", [3, 5])
def test_masked_data_gradient_equals_data_gradient_then_relu_backward(KS, C):
    """isg_conv32_masked / isg_conv16_masked (data gradient with the lower layer's ReLU mask and bias gradient in
    the epilogue) against the two passes it replaces, isg_conv32(relu = 0) then isg_relu_bwd_bias: the masked
    gradient bit for bit (same accumulation, same rounding), the bias gradient up to the order of float additions,
    and against its float64 value.  Ragged image size: partial tiles contribute nothing to the sums."""
    from imagesegmentation_amd import ops
    rs = np.random.RandomState(10 * KS + C)
    B, H, W = 3, 37, 45
    dev = "cuda:0"
    bf = lambda a: torch.tensor(a, device=dev).to(torch.bfloat16)
    tw = torch.tensor(rs.randn(KS, KS, C, C) * 0.05, dtype=torch.float32, device=dev)
    wimg = ops.conv32_pack(tw, data_gradient=True)
    dz = bf(rs.randn(B, H, W, C))
    y_below = bf(np.maximum(rs.randn(B, H, W, C), 0.0))         # a ReLU output: about half of it zero
    got, got_db = ops.conv32_masked(dz, wimg, y_below, KS)
    lin = ops.conv32(dz, wimg, None, KS, relu=False)
    want, want_db = ops.relu_bwd_bias(lin, y_below)
    assert torch.equal(got, want)
    # The fused epilogue sums the float values it masks, the separate pass the bf16-rounded tensor it reads: they
    # differ by the accumulated rounding (about 2^-9 sqrt(n) per channel), and the fused sum is the one closer to
    # the float64 value of the unrounded gradient.
    assert float((got_db - want_db).abs().max()) <= 5e-3 * float(want_db.abs().max())
    w64 = tw.to(torch.bfloat16).double().cpu().numpy()   # the weights as the kernel sees them
    lin64, _, _ = oc.conv2d_same_backward(np.zeros([B, H, W, C]), w64, dz.double().cpu().numpy())
    ref_db = (lin64 * (y_below.double().cpu().numpy() > 0)).sum(axis=(0, 1, 2))
    assert np.abs(got_db.double().cpu().numpy() - ref_db).max() <= 1e-4 * np.abs(ref_db).max() + 1e-3
    assert np.abs(got_db.double().cpu().numpy() - ref_db).max() <= np.abs(want_db.double().cpu().numpy() - ref_db).max()


def test_encoder_chain_gradients_equal_the_layer_by_layer_backward():
    """The encoder as one autograd node (ReLU backward of every layer but the top one inside the data-gradient
    kernels) against the same layers as separate nodes: equal outputs, equal gradients."""
    from imagesegmentation_amd.polyrnn import Model as M
    g = torch.Generator(device="cuda:0").manual_seed(4)
    dt, dev = torch.bfloat16, torch.device("cuda:0")
    B, P = 3, 40
    x = (torch.rand([B, P, P, 3], device=dev, generator=g) > 0.5).to(dt)
    spec = [(3, 32, 14), (32, 32, 5), (32, 32, 3)]
    params = []
    for cin, cout, k in spec:
        params.append(((torch.randn([k, k, cin, cout], device=dev, generator=g) * (1.5 / (k * k * cin) ** 0.5)).requires_grad_(True),
                       (torch.randn([cout], device=dev, generator=g) * 0.1).requires_grad_(True)))
    d_out = torch.randn([B, P, P, 32], device=dev, generator=g).to(dt)
    assert M.encoder_chain_supported(x, params, dt)

    def grads(chain):
        for w, b in params:
            w.grad = b.grad = None
        if chain:
            y = M._EncoderChain.apply(x, *[p for wb in params for p in wb])
        else:
            y = x
            for w, b in params:
                y = M.conv2d_same_relu(y, w, b, dt)
        (y.float() * d_out.float()).sum().backward()
        return y.detach().clone(), [(w.grad.clone(), b.grad.clone()) for w, b in params]

    y1, g1 = grads(True)
    y0, g0 = grads(False)
    assert torch.equal(y1, y0)
    for (dw1, db1), (dw0, db0) in zip(g1, g0):
        assert float((dw1 - dw0).abs().max()) <= 1e-5 * float(dw0.abs().max()) + 1e-6   # same masked gradients in
        assert float((db1 - db0).abs().max()) <= 5e-3 * float(db0.abs().max()) + 1e-3   # float vs bf16-rounded sums


@pytest.mark.parametrize("name", ["History2Large", "History0", "RNNConv", "EndTokenDeep_2"])
def test_fp32_model_against_committed_golden_outputs(name):
    """The HIP path (f32) against tests/golden/convlstm_golden.npz - the oracle's frozen logits, loss, metrics
    and gradient signatures for seeded instances of the experiment graphs (no oracle call in this test)."""
    import os
    from helpers_convlstm import GOLDEN_P, golden_case, grad_signature
    from imagesegmentation_amd.polyrnn import experiments
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convlstm_golden.npz"))
    params, images, histories, durations, targets = golden_case(name)
    model = experiments.make(name, image_size=GOLDEN_P, compute_dtype=torch.float32, device="cuda:0")
    model.max_timesteps = histories.shape[1]
    _load(model, params)
    model._flat_grad.zero_()
    logits, _ = model.forward(images, durations, histories)
    loss, metrics = model._loss_and_metrics(logits, durations, targets)
    loss.backward()
    assert rel_err(logits, gold[name + "/logits"]) < 1e-4
    want = gold[name + "/scalars"]
    assert abs(loss.item() - want[0]) < 1e-5 * max(1.0, abs(want[0]))
    assert abs(metrics["accuracy"].item() - want[1]) < 1e-12
    assert abs(metrics["avg_error"].item() - want[2]) < 1e-9 and abs(metrics["max_error"].item() - want[3]) < 1e-9
    # gradient signatures, in the fixture's order: encoder (w, b), rnn (w_x, w_h, bias), dense (w, b)
    got = []
    for w, b in model.encoder:
        got += [w.grad, b.grad]
    for cell in model.rnn.cells:
        got += [cell.input_to_state_kernel.grad, cell.hidden_to_state_kernel.grad, cell.bias.grad]
    got += [model.dense_kernel.grad, model.dense_bias.grad]
    sigs = gold[name + "/grad_signatures"]
    assert len(got) == len(sigs)
    for g, s in zip(got, sigs):
        mine = grad_signature(g.detach().double().cpu().numpy())
        scale = max(s[1], 1e-12)  # the tensor's L2 norm
        assert abs(mine[1] - s[1]) < 2e-4 * scale, (name, tuple(g.shape))
        # sum and random projection are sums of n terms of size ~norm/sqrt(n): compare on the norm's scale
        assert abs(mine[0] - s[0]) < 2e-4 * scale * np.sqrt(g.numel()) and abs(mine[2] - s[2]) < 2e-3 * scale


def test_fused_adam_matches_the_tensorflow_form():
    """isg_adam_tf over a flat buffer whose length is not a multiple of 4, three steps with the global-norm clip
    active and inactive, against the update written out with torch ops in float64."""
    from imagesegmentation_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    n = 4 * 1000 + 3
    p = torch.randn([n], device=dev, generator=g)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    rp, rm, rv = p.double(), m.double(), v.double()
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
    epoch = ops.weights_epoch()
    for t, (gscale, clip) in enumerate([(5.0, 1.0), (1e-3, 1.0), (2.0, None)], start=1):
        grad = torch.randn([n], device=dev, generator=g) * gscale
        kept = grad.clone()
        norm = grad.norm()
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        ops.adam_tf_(p, m, v, grad, lr_t, b1, b2, eps, grad_norm=norm if clip else None, clip_norm=clip)
        assert torch.equal(grad, kept)  # the gradient itself is left alone
        gd = grad.double() * (clip / max(float(norm), clip) if clip else 1.0)
        rm = b1 * rm + (1 - b1) * gd
        rv = b2 * rv + (1 - b2) * gd * gd
        rp = rp - lr_t * rm / (rv.sqrt() + eps)
        assert float((p.double() - rp).abs().max()) <= 1e-6 * float(rp.abs().max())
        assert float((m.double() - rm).abs().max()) <= 1e-6 * float(rm.abs().max()) + 1e-12
        assert float((v.double() - rv).abs().max()) <= 1e-6 * float(rv.abs().max()) + 1e-12
    assert ops.weights_epoch() == epoch + 3


def test_checkpoint_round_trip_names_retention_and_restored_adam(tmp_path):
    """save() / maybe_restore() (reference polyrnn/Model.py:132-148): TensorFlow-style variable names, global_step
    restored, at most two checkpoint files kept, the `checkpoint` marker naming the newest; and the first step
    after a restore is the one a restored TensorFlow graph takes - the Saver holds variables and global_step
    only, so Adam's moments and bias correction start afresh (step size lr, not lr x 3.2)."""
    import os
    from imagesegmentation_amd.polyrnn import experiments
    P, B = 28, 2
    rs = np.random.RandomState(21)
    path = str(tmp_path / "ckpt") + os.sep
    model = experiments.make("History2Large", image_size=P, device="cuda:0", seed=1, filepath=path)
    hist, T = model.history_length, model.max_timesteps
    images, histories, durations, targets = make_batch(rs, B, T, P, hist)
    model.maybe_restore()  # nothing there yet: creates the directory, restores nothing
    assert os.path.isdir(path) and model.global_step == 0
    names = [n for n, _ in model._named_parameters()]
    assert "Estimator/conv1/kernel" in names and "Estimator/prediction/bias" in names
    assert "Estimator/rnn/rnn/multi_rnn_cell/cell_1/ConvLSTMCell/input_to_state_kernel" in names
    for _ in range(3):
        model.train_step(images, durations, histories, targets)
        model.save()
    files = sorted(f for f in os.listdir(path) if f.startswith("model-"))
    assert files == ["model-2.pt", "model-3.pt"]                      # max_to_keep=2
    assert open(os.path.join(path, "checkpoint")).read().strip() == "model-3.pt"
    sd = torch.load(os.path.join(path, "model-3.pt"), weights_only=True)
    assert sorted(sd) == sorted(names + ["global_step"]) and int(sd["global_step"]) == 3
    want = {n: p.detach().clone() for n, p in model._named_parameters()}

    fresh = experiments.make("History2Large", image_size=P, device="cuda:0", seed=2, filepath=path)
    assert not torch.equal(fresh.dense_kernel, model.dense_kernel)
    fresh.train_step(images, durations, histories, targets)           # gives it non-zero moments to forget
    fresh.maybe_restore()
    assert fresh.global_step == 3 and fresh._adam_t == 0
    for n, p in fresh._named_parameters():
        assert torch.equal(p, want[n]), n
    # the recurrence's packed weight images follow the restored values
    with torch.no_grad():
        a, _ = model.forward(images, durations, histories)
        b, _ = fresh.forward(images, durations, histories)
    assert torch.equal(a, b)
    # one step after the restore against the oracle's Adam from zero moments at t = 1
    before = [p.detach().double().cpu().numpy().copy() for p in fresh.trainable_variables]
    fresh._flat_grad.zero_()
    logits, _ = fresh.forward(images, durations, histories)
    loss, _ = fresh._loss_and_metrics(logits, durations, targets)
    loss.backward()
    grads = [p.grad.detach().double().cpu().numpy().copy() for p in fresh.trainable_variables]
    clipped, _ = oc.clip_by_global_norm(grads, 1.0)
    fresh._apply_gradients()
    assert fresh.global_step == 4 and fresh._adam_t == 1
    for p, p0, g in zip(fresh.trainable_variables, before, clipped):
        ref, _, _ = oc.adam_step(p0, g, np.zeros_like(p0), np.zeros_like(p0), 1)
        step = np.abs(ref - p0).max()
        assert np.abs(p.detach().double().cpu().numpy() - ref).max() < 1e-2 * step + 1e-9
        assert step <= 1.0001e-4                                      # lr, not lr x 3.2


def test_decode_loop_stays_on_the_device():
    """No synchronising call (a blocking copy, .item(), nonzero ...) between the first and the last decoded
    vertex: torch's sync debug mode turns any into an error.  Launches per vertex = cells + head + decode step."""
    from imagesegmentation_amd.polyrnn import experiments
    P, B = 28, 5
    rs = np.random.RandomState(8)
    model = experiments.make("EndTokenDeep_2", image_size=P, device="cuda:0")
    polys = [rs.randint(0, P, [rs.randint(3, 8), 2]) for _ in range(B)]
    images = np.stack([np.stack([oracle_raster.create_shape_mask(p, P)] * 3, axis=2) for p in polys])
    images_dev = torch.tensor(images, dtype=torch.float32, device="cuda:0")
    model._decode_on_device(images_dev, polys)  # first call: weight packing, workspace allocation
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        predicted, durations, _, _, _ = model._decode_on_device(images_dev, polys)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert predicted.shape == (B, model.max_timesteps, 2) and predicted.dtype == torch.int32
    assert model.decode_launches_per_step == model.rnn_layers + 2
    pv, dur = model.decode(images, polys)
    assert pv.dtype == np.float64 and np.array_equal(pv, predicted.cpu().numpy())
    assert dur.dtype == np.uint32 and np.array_equal(dur, durations.cpu().numpy())
    assert dur.min() >= 1 and dur.max() <= model.max_timesteps


@pytest.mark.parametrize("end_token", [False, True])
@pytest.mark.parametrize("hist,T", [(2, 10), (1, 5), (0, 10), (2, 3)])
def test_device_batch_builder_matches_the_host_builder(end_token, hist, T):
    """isg_batch_points (one launch per batch from the device-resident dataset) against
    PolygonDataset.get_batch_points on the same random draws: durations, history points, targets, rolled
    polygons and images, bit for bit; explicit start vertices, negative and beyond-n ones included."""
    from imagesegmentation_amd import data
    P = 28
    ds = data.PolygonDataset.synthetic(24, P, P, T, hist, seed=11, end_token=end_token,
                                       shape_mask_fn=oracle_raster.create_shape_mask)
    builder = data.DeviceBatchBuilder(ds, "cuda:0", image_dtype=torch.float32)
    for start_idx in (None, 0, 3):
        np.random.seed(4)
        d, images, pts, tg, verts = ds.get_batch_points(16, start_idx)
        np.random.seed(4)
        got = builder.get_batch_points(16, start_idx)
        assert np.array_equal(got["durations"].cpu().numpy(), d)
        assert np.array_equal(got["points"].cpu().numpy(), pts) and got["points"].shape == (16, T, hist, 2)
        assert np.array_equal(got["targets"].cpu().numpy(), tg)
        assert np.array_equal(got["images"].cpu().numpy(), images.astype(np.float32))
        nv, rolled = got["nverts"].cpu().numpy(), got["vertices"].cpu().numpy()
        for b in range(16):
            assert nv[b] == len(verts[b]) and np.array_equal(rolled[b, :nv[b]], verts[b]) and not rolled[b, nv[b]:].any()
    # np.roll takes any integer shift
    idx = np.arange(8)
    starts = np.array([-1, -7, 0, 5, 100, 2, -30, 9])
    got = builder.build(idx, starts)
    d, pts, tg = data.sequence_points(ds._table, ds._counts, idx, starts, T, hist, P, end_token)
    assert np.array_equal(got["points"].cpu().numpy(), pts) and np.array_equal(got["targets"].cpu().numpy(), tg)
    for b in range(8):
        n = len(ds._vertices[b])
        assert np.array_equal(got["vertices"].cpu().numpy()[b, :n], np.roll(ds._vertices[b], starts[b], axis=0))
    # drawn on the device: valid samples without repeats, starts inside each polygon, reproducible per seed
    a = builder.get_batch_points(24, device_draw=True, seed=5)
    fresh = data.DeviceBatchBuilder(ds, "cuda:0", image_dtype=torch.float32).get_batch_points(24, device_draw=True, seed=5)
    assert torch.equal(a["points"], fresh["points"]) and torch.equal(a["vertices"], fresh["vertices"])
    assert sorted(a["nverts"].cpu().tolist()) == sorted(len(v) for v in ds._vertices)


def test_training_from_the_device_batch_builder_equals_host_batches():
    """A train step fed by DeviceBatchBuilder tensors is the train step fed by the host tuple."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrnn import experiments
    P = 28
    ds = data.PolygonDataset.synthetic(16, P, P, 10, 2, seed=1, shape_mask_fn=oracle_raster.create_shape_mask)
    builder = data.DeviceBatchBuilder(ds, "cuda:0", image_dtype=torch.float32)
    losses = []
    for use_device in (False, True):
        model = experiments.make("History2Large", image_size=P, device="cuda:0", seed=3)
        np.random.seed(0)
        if use_device:
            b = builder.get_batch_points(8)
            loss = model.train_step(b["images"], b["durations"], b["points"], b["targets"])
        else:
            d, images, pts, tg, _ = ds.get_batch_points(8)
            loss = model.train_step(images, d, pts, tg)
        losses.append((loss.item(), model.dense_bias.detach().clone()))
    assert losses[0][0] == losses[1][0] and torch.equal(losses[0][1], losses[1][1])


@pytest.mark.gpu
def test_encoder_weight_images_match_their_documented_layouts():
    """isg_conv32_pack / isg_conv_in_pack (one launch per image) against the element formulas of include/isg.h
    written out with torch: forward and data-gradient images of the square layers (3 x 3 and 5 x 5, 16 and 32
    channels), the first layer's image (14 x 14 -> 32, 5 x 5 -> 16).  Bit-exact: a layout, and one rounding to bf16."""
    from imagesegmentation_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(5)
    for ks in (3, 5):
        for c in (16, 32):
            w = torch.randn([ks, ks, c, c], device=d, generator=g)
            for dgrad in (False, True):
                a = w.flip(0, 1).permute(0, 1, 3, 2) if dgrad else w          # A[ty][tx][k][m]
                a = torch.nn.functional.pad(a, (0, 32 - c))
                want = a.reshape(ks * ks, c // 16, 2, 8, 32).permute(0, 1, 2, 4, 3).reshape(ks * ks * (c // 16), 64, 8)
                got = ops.conv32_pack(w, data_gradient=dgrad)
                assert got.dtype == torch.bfloat16 and torch.equal(got, want.to(torch.bfloat16)), (ks, c, dgrad)
    for ks, cout in ((14, 32), (5, 16), (14, 16), (5, 32)):
        w = torch.randn([ks, ks, 3, cout], device=d, generator=g)
        kxp = (ks + 3) // 4 * 4
        a = torch.nn.functional.pad(w, (0, 32 - cout, 0, 1, 0, kxp - ks))      # [ky][kx][4][32]
        want = a.reshape(ks, kxp // 4, 2, 2, 4, 32).permute(0, 1, 2, 5, 3, 4).reshape(ks * (kxp // 4), 64, 8)
        got = ops.conv_in_pack(w)
        assert torch.equal(got, want.to(torch.bfloat16)), (ks, cout)


def test_bf16_training_tracks_f32_training_over_200_steps():
    """History2Large at 28 x 28 trained for 200 steps on a fixed 64-polygon dataset (the same batches, the same
    initial weights, Adam 1e-3), once in f32 and once in bf16: the bf16 loss curve stays with the f32 one - the
    mean loss of the last 20 steps within 5 % - and both have learnt (below 70 % of the first steps' loss).
    (reference polyrnn/Model.py:96-112: the loss and the update this run repeats.)"""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrnn import experiments
    ds = data.PolygonDataset.synthetic(64, 28, 28, 10, 2, seed=4, shape_mask_fn=oracle_raster.create_shape_mask)
    curves = {}
    for dt in (torch.float32, torch.bfloat16):
        model = experiments.make("History2Large", image_size=28, compute_dtype=dt, device="cuda:0", seed=11,
                                 learning_rate=1e-3)
        np.random.seed(123)
        losses = []
        for _ in range(200):
            d, images, pts, t, _ = ds.get_batch_points(16)
            losses.append(model.train_step(images, d, pts, t))
        curves[dt] = torch.stack(losses).float().cpu().numpy()
    f32, bf16 = curves[torch.float32], curves[torch.bfloat16]
    first, last32, last16 = f32[:5].mean(), f32[-20:].mean(), bf16[-20:].mean()
    print("loss: first steps %.4f, last 20 steps f32 %.4f bf16 %.4f; largest gap over the run %.4f" %
          (first, last32, last16, np.abs(f32 - bf16).max()))
    assert np.isfinite(f32).all() and np.isfinite(bf16).all()
    assert last32 < 0.7 * first and last16 < 0.7 * first
    assert abs(last16 - last32) <= 0.05 * last32
    # step by step the two runs see the same batch: their losses stay within 15 % of each other all the way
    assert np.abs(f32 - bf16).max() <= 0.15 * f32.max()


@pytest.mark.parametrize("name,dtype", [("EndTokenDeep_2", torch.bfloat16), ("History2Large", torch.bfloat16),
                                        ("History2Large", torch.float32)])
def test_graph_replayed_training_step_equals_the_eager_step(name, dtype):
    """Model.step_graphs: forward + loss + backward of a step replayed as a HIP graph (one per input shape and number
    of recurrence steps) against the same steps launched kernel by kernel - the same kernels in the same order, so
    losses, metrics and parameters agree bit for bit over several steps with batches of different longest length
    (different graphs), Adam's step count and the weight-dependent caches included."""
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrnn import experiments
    ds = data.PolygonDataset.synthetic(24, 28, 28, 10, 2, seed=9, shape_mask_fn=oracle_raster.create_shape_mask,
                                       end_token=(name == "EndTokenDeep_2"))
    runs = {}
    for graphs in (False, True):
        model = experiments.make(name, image_size=28, compute_dtype=dtype, device="cuda:0", seed=5, learning_rate=1e-3)
        model.step_graphs = graphs
        np.random.seed(77)
        losses, accs = [], []
        for _ in range(6):
            d, images, pts, t, _ = ds.get_batch_points(8)
            losses.append(model.train_step(images, d, pts, t).clone())
            accs.append(model.last_metrics["accuracy"].clone())
        runs[graphs] = (torch.stack(losses), torch.stack(accs), model._flat_param.clone(), model._adam_t,
                        len(model._step_graph_cache))
    eager, graph = runs[False], runs[True]
    # (an f32 model's encoder runs on the vendor convolution, whose backward cannot be captured: it stays eager)
    assert (graph[4] >= 1) == (dtype == torch.bfloat16) and eager[4] == 0
    if dtype == torch.bfloat16:
        assert torch.equal(eager[0], graph[0]) and torch.equal(eager[1], graph[1])
        assert torch.equal(eager[2], graph[2])
    else:  # (two eager runs: the vendor convolution's weight gradient is not bit-reproducible)
        assert torch.allclose(eager[0], graph[0], rtol=1e-4) and torch.allclose(eager[2], graph[2], atol=1e-4)
    assert eager[3] == graph[3] == 6
    assert bool((eager[0][-1] < eager[0][0]).item())


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype", [("RNNConv", torch.bfloat16), ("EndTokenDeep_2", torch.bfloat16),
                                        ("History2Large", torch.bfloat16), ("History2Large", torch.float32)])
def test_one_refresh_launch_gives_every_packed_image_the_packing_kernels_give(name, dtype, monkeypatch):
    """ops.DerivedWeights: after the first training step every weight image, tap table and compute-type copy the
    step used is a view of two shared buffers that ONE launch (isg_refresh_derived) fills from the flat parameter
    buffer through index maps - derived by running the packing kernels on index values.  Checked: (1) after
    further optimiser steps each refreshed tensor equals, bit for bit, what its own packing kernel writes from the
    same parameters; (2) training with the registry is training without it (ISG_NO_DERIVED=1), bit for bit."""
    from imagesegmentation_amd import data, ops
    from imagesegmentation_amd.polyrnn import experiments
    ds = data.PolygonDataset.synthetic(24, 28, 28, 10, 1 if name == "RNNConv" else 2, seed=9,
                                       shape_mask_fn=oracle_raster.create_shape_mask, end_token=(name == "EndTokenDeep_2"))
    runs = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("ISG_NO_DERIVED", "1")
        model = experiments.make(name, image_size=28, compute_dtype=dtype, device="cuda:0", seed=5,
                                 learning_rate=1e-3)
        np.random.seed(77)
        losses = []
        for _ in range(4):
            d, images, pts, t, _ = ds.get_batch_points(8)
            losses.append(model.train_step(images, d, pts, t).clone())
        runs[off] = (torch.stack(losses), model._flat_param.clone())
        if not off:
            reg = model._derived
            assert isinstance(reg, ops.DerivedWeights) and reg._plan is not None and not reg._recorded
            # one LSTM entry per layer, the head, the first encoder layer, two images per further encoder layer
            # (a float model: the cells' float images only - its encoder runs on the vendor convolution and its
            # head reads the master weights)
            assert len(reg._compiled) == len(model.rnn.cells) + \
                (1 + 1 + 2 * (len(model.encoder) - 1) if dtype == torch.bfloat16 else 0)
            reg.ensure_fresh()
            for key, (obj, names, recompute, _) in reg._compiled.items():
                before = {n: getattr(obj, n).clone() for n in names if getattr(obj, n) is not None}
                for n in before:
                    getattr(obj, n).zero_()
                recompute()  # the entry's own packing kernels, into the same tensors
                for n, ref in before.items():
                    got = getattr(obj, n)
                    assert torch.equal(got.view(torch.int16) if got.dtype == torch.bfloat16 else got,
                                       ref.view(torch.int16) if ref.dtype == torch.bfloat16 else ref), (key, n)
    if dtype == torch.bfloat16:
        assert torch.equal(runs[False][0], runs[True][0]) and torch.equal(runs[False][1], runs[True][1])
    else:  # (the vendor convolution's weight gradient is not bit-reproducible between two runs)
        assert torch.allclose(runs[False][0], runs[True][0], rtol=1e-4) and \
            torch.allclose(runs[False][1], runs[True][1], atol=1e-4)
    assert bool((runs[False][0][-1] < runs[False][0][0]).item())


@pytest.mark.gpu
def test_refresh_after_a_checkpoint_restore_and_in_place_edits():
    """The registry notices parameter changes it did not make: ops.note_weights_changed() (checkpoint restore into
    the flat buffer) and in-place edits of a parameter that torch's version counter sees."""
    from imagesegmentation_amd import data, ops
    from imagesegmentation_amd.polyrnn import experiments
    ds = data.PolygonDataset.synthetic(16, 28, 28, 10, 2, seed=2, shape_mask_fn=oracle_raster.create_shape_mask)
    np.random.seed(1)
    d, images, pts, t, _ = ds.get_batch_points(8)
    model = experiments.make("History2Large", image_size=28, device="cuda:0", seed=5)
    model.train_step(images, d, pts, t)
    model.train_step(images, d, pts, t)

    def valid_loss(m):
        m.validate(images, d, pts, t)
        return m.loss.item()

    base = valid_loss(model)
    keep = model._flat_param.clone()
    with torch.no_grad():
        model._flat_param.mul_(0.5)
    ops.note_weights_changed()
    halved = valid_loss(model)
    assert halved != base
    with torch.no_grad():
        model._flat_param.copy_(keep)
    ops.note_weights_changed()
    assert valid_loss(model) == base
    with torch.no_grad():
        model.rnn.cells[0].input_to_state_kernel.mul_(0.0)  # (a tracked in-place edit, no epoch change)
    assert valid_loss(model) != base


@pytest.mark.gpu
def test_refresh_entry_point_gathers_and_converts():
    """isg_refresh_derived through the binding: gathered bf16 and float sections (negative map entries give zero,
    sections of odd lengths), up to four converted ranges of lengths that are no multiple of 8 and start on odd
    offsets, against torch indexing; bad arguments are refused."""
    import ctypes
    from imagesegmentation_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    n = 100_003
    src = torch.randn([n], device=dev, generator=g)
    m16 = torch.randint(-1, n, [5001], device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    m32 = torch.randint(-1, n, [777], device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    o16 = torch.full([5001], 7.0, dtype=torch.bfloat16, device=dev)
    o32 = torch.full([777], 7.0, dtype=torch.float32, device=dev)
    ranges = [(1, 4097), (5000, 8), (20001, 30000), (99000, 1003)]
    outs = [torch.full([k], 7.0, dtype=torch.bfloat16, device=dev) for _, k in ranges]
    lib = _lib.lib()

    def call(nr, n_src=n):
        a = (ctypes.c_longlong * 4)(*[r[0] for r in ranges])
        b = (ctypes.c_longlong * 4)(*[r[1] for r in ranges])
        c = (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs])
        return lib.isg_refresh_derived(_lib.ptr(src), n_src, _lib.ptr(m16), m16.numel(), _lib.ptr(o16), _lib.ptr(m32),
                                       m32.numel(), _lib.ptr(o32), nr, a, b, c, _lib.current_stream())
    assert call(4) == 0
    gather = lambda m: torch.where(m >= 0, src[m.clamp(min=0).long()], torch.zeros([], device=dev))
    assert torch.equal(o16, gather(m16).to(torch.bfloat16)) and torch.equal(o32, gather(m32))
    for (a, k), o in zip(ranges, outs):
        assert torch.equal(o, src[a:a + k].to(torch.bfloat16))
    assert call(5) != 0                    # more than four ranges
    assert call(4, n_src=99_000) != 0      # a range that ends past the source


@pytest.mark.gpu
@pytest.mark.parametrize("KS,C,cin", [(3, 32, 5), (5, 32, 3), (3, 16, 5)])
def test_padded_input_layer_on_the_square_kernels(KS, C, cin):
    """A layer with fewer input channels than filters on isg_conv32 / isg_conv16 over a zero-padded input
    (isg_conv32_pack_padded) against a float64 convolution of the bf16-rounded operands, forward and weight / bias
    gradients (reference polyrl/a3c/model.py:19-20: 5 -> 32 channels, 3 x 3)."""
    from imagesegmentation_amd.polyrnn.Model import conv2d_same_relu
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(KS * 100 + C + cin)
    x = torch.randn([3, 21, 37, cin], device=dev, generator=g).bfloat16()
    w = (torch.randn([KS, KS, cin, C], device=dev, generator=g) * 0.2).requires_grad_(True)
    b = (torch.randn([C], device=dev, generator=g) * 0.1).requires_grad_(True)
    y = conv2d_same_relu(x, w, b, torch.bfloat16)
    up = torch.randn(y.shape, device=dev, generator=g).bfloat16()
    y.backward(up)
    xd = x.double().permute(0, 3, 1, 2)
    wd = w.detach().bfloat16().double().permute(3, 2, 0, 1).requires_grad_(True)
    bd = b.detach().double().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.conv2d(xd, wd, bd, padding=KS // 2)).permute(0, 2, 3, 1)
    assert (y.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    # the backward pass sees the bf16 output's mask and the bf16 upstream gradient
    (ref * 0).sum().backward()
    wd.grad = None
    bd.grad = None
    pre = torch.nn.functional.conv2d(xd, wd, bd, padding=KS // 2)
    (torch.relu(pre).permute(0, 2, 3, 1) * up.double()).sum().backward()
    gw = wd.grad.permute(2, 3, 1, 0)
    assert (w.grad.double() - gw).abs().max().item() <= 3e-2 * gw.abs().max().item()
    assert (b.grad.double() - bd.grad).abs().max().item() <= 3e-2 * bd.grad.abs().max().item()


def test_dense_head_on_the_padded_class_axis(monkeypatch):
    """_DenseHead over many rows runs its three GEMMs on a copy of the weights padded to a multiple of 8 classes:
    same logits and gradients as on the unpadded copy (bf16 GEMMs whose reduction splits may differ), pad columns
    zero, nothing of the padding in what comes back."""
    from imagesegmentation_amd.polyrnn import Model as M
    g = torch.Generator(device="cuda:0").manual_seed(8)
    R, K, N = 2048, 1568, 785
    flat0 = (torch.randn([R, K], device="cuda:0", generator=g) * 0.5).bfloat16()
    w = torch.nn.Parameter(torch.randn([K, N], device="cuda:0", generator=g) * 0.02)
    b = torch.nn.Parameter(torch.randn([N], device="cuda:0", generator=g) * 0.1)
    up = torch.randn([R, N], device="cuda:0", generator=g).bfloat16()
    res = {}
    for padded in (True, False):
        if not padded:
            monkeypatch.setenv("ISG_NO_PADDED_HEAD", "1")
        cache = {}
        flat = flat0.clone().requires_grad_(True)
        w.grad = b.grad = None
        out = M._DenseHead.apply(flat, w, b, cache)
        assert tuple(out.shape) == (R, N) and out.is_contiguous()
        out.backward(up)
        res[padded] = (out.detach().float(), flat.grad.float(), w.grad.clone(), b.grad.clone())
        if padded:
            assert tuple(cache["w8"].shape) == (K, 792) and not bool(cache["w8"][:, N:].any()) and not bool(cache["b8"][N:].any())
        else:
            assert "w8" not in cache
    ref = flat0.double() @ w.detach().bfloat16().double() + b.detach().bfloat16().double()
    scale = float(ref.abs().max())
    for padded in (True, False):
        assert float((res[padded][0].double() - ref).abs().max()) <= 2.0 ** -7 * scale
    for a, c in zip(res[True][1:], res[False][1:]):
        assert a.shape == c.shape
        assert float((a - c).abs().max()) <= 2.0 ** -6 * float(c.abs().max())
