"""The NumPy ConvLSTM oracle against a float64 torch re-expression and closed-form cases.
(The reference has no golden vectors for this path: TensorFlow 1.x is not importable.)"""
import numpy as np
import torch

from oracle import convlstm as oc
from helpers_convlstm import (make_batch, make_params, params_to_torch, t64, torch_cell, torch_conv_same,
                              torch_model_logits)


def test_conv_same_matches_torch_including_even_kernels():
    rs = np.random.RandomState(0)
    for k, cin, cout in [(3, 5, 7), (5, 3, 4), (14, 3, 2), (1, 2, 3)]:
        x = rs.randn(2, 9, 11, cin)
        w = rs.randn(k, k, cin, cout)
        b = rs.randn(cout)
        got = oc.conv2d_same(x, w, b)
        ref = torch_conv_same(t64(x), t64(w), t64(b)).numpy()
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_cell_forward_backward_match_torch_autograd():
    rs = np.random.RandomState(1)
    B, H, W, C, Fh = 2, 6, 7, 5, 4
    x, c, h = rs.randn(B, H, W, C), rs.randn(B, H, W, Fh), rs.randn(B, H, W, Fh)
    w_x, w_h, bias = rs.randn(3, 3, C, 4 * Fh) * 0.3, rs.randn(3, 3, Fh, 4 * Fh) * 0.3, rs.randn(4 * Fh) * 0.1
    h2, c2, cache = oc.cell_forward(x, c, h, w_x, w_h, bias, return_cache=True)
    tx, tc, th, twx, twh, tb = [t64(a).requires_grad_(True) for a in (x, c, h, w_x, w_h, bias)]
    rh, rc = torch_cell(tx, tc, th, twx, twh, tb)
    assert np.allclose(h2, rh.detach().numpy(), atol=1e-12) and np.allclose(c2, rc.detach().numpy(), atol=1e-12)
    dh, dc = rs.randn(*h2.shape), rs.randn(*c2.shape)
    (rh * t64(dh)).sum().add((rc * t64(dc)).sum()).backward()
    g = oc.cell_backward(cache, w_x, w_h, dh, dc)
    for got, ref in [(g["dx"], tx.grad), (g["dh_prev"], th.grad), (g["dc_prev"], tc.grad), (g["dw_x"], twx.grad),
                     (g["dw_h"], twh.grad), (g["dbias"], tb.grad)]:
        assert np.allclose(got, ref.numpy(), rtol=1e-10, atol=1e-10)


def test_cell_closed_forms():
    B, H, W, C, Fh = 1, 4, 4, 3, 2
    rs = np.random.RandomState(2)
    x, c, h = rs.randn(B, H, W, C), rs.randn(B, H, W, Fh), rs.randn(B, H, W, Fh)
    z = lambda *s: np.zeros(s)
    # all-zero weights and bias: c' = 0.5 c, h' = 0.5 tanh(c')   (forget_bias is NOT applied)
    h2, c2 = oc.cell_forward(x, c, h, z(3, 3, C, 4 * Fh), z(3, 3, Fh, 4 * Fh), z(4 * Fh))
    assert np.allclose(c2, 0.5 * c) and np.allclose(h2, 0.5 * np.tanh(0.5 * c))
    # bias only, zero state and input: spatially constant output
    b = rs.randn(4 * Fh)
    h3, c3 = oc.cell_forward(0 * x, 0 * c, 0 * h, rs.randn(3, 3, C, 4 * Fh), rs.randn(3, 3, Fh, 4 * Fh), b)
    assert np.allclose(h3, h3[:, :1, :1]) and np.allclose(c3, c3[:, :1, :1])
    j, i, f, o = np.split(b, 4)
    assert np.allclose(c3[0, 0, 0], oc.sigmoid(i) * np.tanh(j))


def test_model_logits_loss_grads_match_torch():
    rs = np.random.RandomState(3)
    P, hist, T, B = 8, 2, 4, 3
    for end_token in (False, True):
        units = P * P + (1 if end_token else 0)
        params = make_params(rs, P, hist, encoder=((4, 5), (4, 3)), layers=2, F_=4, units=units)
        images, histories, durations, targets = make_batch(rs, B, T, P, hist, end_token)
        durations[0] = 0  # a row that never runs: zero outputs, logits == dense bias
        loss, logits, grads = oc.model_loss_and_grads(params, images, histories, durations, targets, P)
        tp = params_to_torch(params)
        tl, _ = torch_model_logits(tp, t64(images), t64(histories), torch.tensor(durations))
        assert np.allclose(logits, tl.detach().numpy(), atol=1e-11)
        assert np.allclose(logits[0], params["dense"]["b"][None, :])
        labels = torch.tensor(oc.targets_dense(targets, P))
        tloss = torch.nn.functional.cross_entropy(tl.reshape(B * T, -1), labels.reshape(-1))
        assert abs(loss - tloss.item()) < 1e-11
        tloss.backward()
        for gl, tlp in zip(grads["encoder"] + grads["rnn"] + [grads["dense"]],
                           tp["encoder"] + tp["rnn"] + [tp["dense"]]):
            for k in gl:
                assert np.allclose(gl[k], tlp[k].grad.numpy(), rtol=1e-9, atol=1e-11), k


def test_dynamic_rnn_masking_and_state_carry():
    rs = np.random.RandomState(4)
    T, B, H, C, Fh = 3, 2, 5, 3, 2
    layers = [dict(w_x=rs.randn(3, 3, C, 4 * Fh), w_h=rs.randn(3, 3, Fh, 4 * Fh), bias=rs.randn(4 * Fh)),
              dict(w_x=rs.randn(3, 3, Fh, 4 * Fh), w_h=rs.randn(3, 3, Fh, 4 * Fh), bias=rs.randn(4 * Fh))]
    x = rs.randn(T, B, H, H, C)
    outs, final = oc.rnn_forward(x, [1, 3], layers)
    assert np.all(outs[1:, 0] == 0) and np.any(outs[2, 1] != 0)
    # the state of sample 0 froze after its single step
    o1, f1 = oc.rnn_forward(x[:1], [1, 1], layers)
    for l in range(2):
        assert np.allclose(final[l][0][0], f1[l][0][0]) and np.allclose(final[l][1][0], f1[l][1][0])
    # feeding the final state back as the initial state continues the sequence (decode loop, Model.py:231-236)
    o2, f2 = oc.rnn_forward(x[1:], [0, 2], layers, init_state=f1)
    assert np.allclose(o2[:, 1], outs[1:, 1]) and np.allclose(f2[1][1][1], final[1][1][1])


def test_metrics_and_targets():
    P = 4
    logits = np.zeros([2, 2, P * P])
    logits[0, 0, 1 * P + 2] = 5   # predicts (x=1, y=2)
    logits[0, 1, 0] = 1
    logits[1, 0, 3 * P + 3] = 2
    targets = np.array([[[1, 2], [0, 0]], [[3, 0], [0, 0]]])
    acc, avg, mx = oc.metrics(logits, targets, [2, 1], P)
    assert acc == 2 / 3 and mx == 3.0 and abs(avg - 1.0) < 1e-15
    assert oc.targets_dense(np.array([P, 0]), P) == P * P
    assert np.array_equal(oc.prediction_max(np.array([[0., 7., 7.]]), 2), [[0, 1]])  # first maximum wins
    assert np.allclose(oc.discount_rewards([0, 0, 1.0]), [0.9801, 0.99, 1.0])


def test_oracle_reproduces_its_frozen_outputs():
    """tests/golden/convlstm_golden.npz (made by tests/golden/make_convlstm_fixtures.py) freezes the oracle's
    outputs on seeded instances of four experiment graphs: any drift of the restatement shows here."""
    import os
    from helpers_convlstm import GOLDEN_CASES, GOLDEN_P, flat_grad_list, golden_case, grad_signature
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convlstm_golden.npz"))
    for name in sorted(GOLDEN_CASES):
        params, images, histories, durations, targets = golden_case(name)
        chk = np.array([images.sum(), histories.sum(), durations.sum(), targets.sum(),
                        sum(np.abs(p["w"]).sum() for p in params["encoder"])])
        assert np.allclose(chk, gold[name + "/input_checksum"], rtol=1e-12), "seeded inputs changed: " + name
        loss, logits, grads = oc.model_loss_and_grads(params, images, histories, durations, targets, GOLDEN_P)
        assert np.allclose(logits, gold[name + "/logits"], rtol=1e-9, atol=1e-12)
        assert np.allclose([loss, *oc.metrics(logits, targets, durations, GOLDEN_P)], gold[name + "/scalars"], rtol=1e-9)
        sig = np.stack([grad_signature(g) for g in flat_grad_list(grads)])
        assert np.allclose(sig, gold[name + "/grad_signatures"], rtol=1e-8, atol=1e-12)
