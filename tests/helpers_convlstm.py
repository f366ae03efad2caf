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
