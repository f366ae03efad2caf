"""Shared builders for ConvLSTM tests: random parameters and a float64 torch re-expression of the
reference graph (F.conv2d + autograd), used to cross-check the NumPy oracle."""
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


def torch_conv_same(x, w, b=None):
    """x [B,H,W,C] float64 tensor, w HWIO tensor -> [B,H,W,O]; TensorFlow SAME padding."""
    kh, kw = w.shape[0], w.shape[1]
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, kw - 1 - pl, pt, kh - 1 - pt))
    y = F.conv2d(xp, w.permute(3, 2, 0, 1), b)
    return y.permute(0, 2, 3, 1)


def torch_cell(x, c, h, w_x, w_h, bias):
    y = torch_conv_same(h, w_h) + torch_conv_same(x, w_x) + bias
    j, i, f, o = torch.split(y, y.shape[3] // 4, dim=3)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(j)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def torch_model_logits(tp, images, histories, durations):
    """tp: params as float64 tensors (same nesting). Returns logits [B,T,K] and final state."""
    x = images
    for p in tp["encoder"]:
        x = torch.relu(torch_conv_same(x, p["w"], p["b"]))
    B, T = histories.shape[:2]
    Fh = tp["rnn"][0]["w_h"].shape[2]
    P = x.shape[1]
    state = [(torch.zeros(B, P, P, Fh, dtype=torch.float64), torch.zeros(B, P, P, Fh, dtype=torch.float64))
             for _ in tp["rnn"]]
    outs = []
    for t in range(T):
        live = (t < durations).to(torch.float64)[:, None, None, None]
        inp = torch.cat([x, histories[:, t]], dim=3)
        new_state = []
        for l, p in enumerate(tp["rnn"]):
            c_prev, h_prev = state[l]
            h, c = torch_cell(inp, c_prev, h_prev, p["w_x"], p["w_h"], p["bias"])
            new_state.append((live * c + (1 - live) * c_prev, live * h + (1 - live) * h_prev))
            inp = h
        outs.append(live * inp)
        state = new_state
    rnn_out = torch.stack(outs, dim=1)
    logits = rnn_out.reshape(B * T, -1) @ tp["dense"]["w"] + tp["dense"]["b"]
    return logits.reshape(B, T, -1), state


def params_to_torch(params, requires_grad=True):
    def conv(d):
        return {k: t64(v).requires_grad_(requires_grad) for k, v in d.items()}
    return dict(encoder=[conv(p) for p in params["encoder"]], rnn=[conv(p) for p in params["rnn"]],
                dense=conv(params["dense"]))
