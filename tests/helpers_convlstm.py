"""Shared builders for ConvLSTM tests: random parameters and batches; the float64 torch re-expression
of the reference graph used to cross-check the NumPy oracle lives in oracle/torch_port.py."""
import numpy as np
import torch
import torch.nn.functional as F


def glorot(rs, shape):
    """glorot-uniform over an HWIO conv kernel or an [in, out] matrix (TensorFlow's default)."""
    if len(shape) == 4:
        fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
    else:
        fan_in, fan_out = shape
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, shape)


def make_params(rs, P, hist, encoder=((16, 5), (16, 5), (16, 3)), layers=2, F_=16, units=None, bias_scale=0.1,
                w_scale=1.0):
    units = P * P if units is None else units
    enc, cin = [], 3
    for f, k in encoder:
        enc.append(dict(w=glorot(rs, (k, k, cin, f)) * w_scale, b=rs.uniform(-bias_scale, bias_scale, [f])))
        cin = f
    rnn, c = [], cin + hist
    for _ in range(layers):
        rnn.append(dict(w_x=glorot(rs, (3, 3, c, 4 * F_)) * w_scale, w_h=glorot(rs, (3, 3, F_, 4 * F_)) * w_scale,
                        bias=rs.uniform(-bias_scale, bias_scale, [4 * F_])))
        c = F_
    dense = dict(w=glorot(rs, (P * P * F_, units)), b=rs.uniform(-bias_scale, bias_scale, [units]))
    return dict(encoder=enc, rnn=rnn, dense=dense)


def make_batch(rs, B, T, P, hist, end_token=False):
    """Teacher-forcing batch with the reference's layout (Dataset.py:104-150): random polygons of
    3..8 vertices, duration = n - hist + 1, padded steps all-zero."""
    images = (rs.uniform(size=[B, P, P, 3]) > 0.5).astype(np.float64)
    histories = np.zeros([B, T, P, P, hist])
    targets = np.zeros([B, T, 2], np.int32)
    durations = np.zeros([B], np.int32)
    for b in range(B):
        n = rs.randint(3, 9)
        v = rs.randint(0, P, [n, 2])
        durations[b] = min(n - hist + 1, T)
        for t in range(durations[b]):
            for i in range(hist):
                x, y = v[(t - hist + i + 1) % n]
                histories[b, t, y, x, i] = 1
            targets[b, t] = v[(t + 1) % n]
        if end_token:
            targets[b, durations[b] - 1] = (P, 0)
    return images, histories, durations, targets


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


from oracle.torch_port import cell as torch_cell  # noqa: E402,F401
from oracle.torch_port import conv_same as torch_conv_same  # noqa: E402,F401
from oracle.torch_port import model_logits as torch_model_logits  # noqa: E402,F401


def params_to_torch(params, requires_grad=True):
    def conv(d):
        return {k: t64(v).requires_grad_(requires_grad) for k, v in d.items()}
    return dict(encoder=[conv(p) for p in params["encoder"]], rnn=[conv(p) for p in params["rnn"]],
                dense=conv(params["dense"]))


# ---- frozen oracle cases (tests/golden/convlstm_golden.npz) ---------------------------------------------
# name -> (encoder spec, rnn layers, history length, T, end token); P = 6, B = 2 throughout.  Weights and
# inputs are rebuilt from fixed numpy RandomState seeds (whose streams numpy keeps stable); the file
# holds the oracle's OUTPUTS for them.
GOLDEN_CASES = {
    "History2Large": (((16, 5), (16, 5), (16, 3)), 2, 2, 3, False),
    "History0": (((16, 5), (16, 5), (16, 3)), 2, 0, 3, False),
    "RNNConv": (((32, 14), (32, 5), (32, 3)), 2, 1, 2, False),
    "EndTokenDeep_2": (((16, 5), (16, 5)) + ((16, 3),) * 5, 5, 2, 3, True),
}
GOLDEN_P, GOLDEN_B = 6, 2


def golden_case(name):
    """(params rounded to float32, images, histories, durations, targets, encoder spec, layers, end_token)."""
    enc, layers, hist, T, end_token = GOLDEN_CASES[name]
    rs = np.random.RandomState(100 + sorted(GOLDEN_CASES).index(name))
    units = GOLDEN_P * GOLDEN_P + (1 if end_token else 0)
    params = make_params(rs, GOLDEN_P, hist, encoder=enc, layers=layers, units=units)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)  # what a float32 model can hold exactly
    params = dict(encoder=[dict(w=f32(p["w"]), b=f32(p["b"])) for p in params["encoder"]],
                  rnn=[dict(w_x=f32(p["w_x"]), w_h=f32(p["w_h"]), bias=f32(p["bias"])) for p in params["rnn"]],
                  dense=dict(w=f32(params["dense"]["w"]), b=f32(params["dense"]["b"])))
    images, histories, durations, targets = make_batch(rs, GOLDEN_B, T, GOLDEN_P, hist, end_token=end_token)
    durations[1] = min(durations[1], 2)
    return params, images, histories, durations, targets


def grad_signature(g):
    """(sum, L2 norm, dot with a fixed pseudo-random direction) of one gradient tensor: three numbers that pin
    it as well as the tensor itself would, at a fraction of the size."""
    g = np.asarray(g, np.float64).reshape(-1)
    d = np.random.RandomState(g.size % 9973 + 7).standard_normal(g.size)
    return np.array([g.sum(), np.sqrt((g * g).sum()), (g * d).sum()])


def flat_grad_list(grads):
    """Gradient tensors in a fixed order: encoder (w, b) ..., rnn (w_x, w_h, bias) ..., dense (w, b)."""
    out = []
    for g in grads["encoder"]:
        out += [g["w"], g["b"]]
    for g in grads["rnn"]:
        out += [g["w_x"], g["w_h"], g["bias"]]
    return out + [grads["dense"]["w"], grads["dense"]["b"]]


def backward_factors(gates):
    """What a forward step saves for the fused backward step (include/isg.h, ISG_SAVE_FACTORS), from gate
    activations [.., F, 4] (tanh j, sig i, sig f, sig o): float64 [.., F/2, 4, 2], per channel pair
    A0 A1 B0 B1 f0 f1 o0 o1 with A = i (1 - j^2), B = j i (1 - i)."""
    g, i, f, o = [gates[..., k].double() for k in range(4)]
    fac = torch.stack([i * (1 - g * g), g * i * (1 - i), f, o], dim=-1)   # [.., F, 4]
    lead = fac.shape[:-2]
    return fac.reshape(*lead, fac.shape[-2] // 2, 2, 4).transpose(-1, -2).contiguous()
