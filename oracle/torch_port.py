"""The reference graph re-expressed with stock torch ops (F.conv2d + autograd), any dtype, CPU.
TEST INFRASTRUCTURE ONLY: used by tests as a second opinion on the NumPy oracle and by the
cpu_baseline leg of bench.py as the CPU port timed on the host cores.  The shipped package never
imports it.

Follows reference convLSTM.py:26-45 (cell), polyrnn/History2LargeExperiment.py:13-50 (graph) and
polyrnn/Model.py:96-100 (loss); TensorFlow is not importable here, so this is a port, not the
reference itself.
"""
import torch
import torch.nn.functional as F


def conv_same(x, w, b=None):
    """x [B,H,W,C], w HWIO -> [B,H,W,O]; TensorFlow SAME padding (even kernels pad less before)."""
    kh, kw = w.shape[0], w.shape[1]
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, kw - 1 - pl, pt, kh - 1 - pt))
    return F.conv2d(xp, w.permute(3, 2, 0, 1), b).permute(0, 2, 3, 1)


def cell(x, c, h, w_x, w_h, bias):
    y = conv_same(h, w_h) + conv_same(x, w_x) + bias
    j, i, f, o = torch.split(y, y.shape[3] // 4, dim=3)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(j)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def model_logits(tp, images, histories, durations, head_pool=1):
    """tp: dict(encoder=[dict(w,b)], rnn=[dict(w_x,w_h,bias)], dense=dict(w,b)) of tensors.
    Returns (logits [B,T,K], final state)."""
    x = images
    for p in tp["encoder"]:
        x = torch.relu(conv_same(x, p["w"], p["b"]))
    B, T = histories.shape[:2]
    Fh = tp["rnn"][0]["w_h"].shape[2]
    P = x.shape[1]
    state = [(torch.zeros(B, P, P, Fh, dtype=x.dtype), torch.zeros(B, P, P, Fh, dtype=x.dtype)) for _ in tp["rnn"]]
    outs = []
    for t in range(T):
        live = (t < durations).to(x.dtype)[:, None, None, None]
        inp = torch.cat([x, histories[:, t]], dim=3)
        new_state = []
        for l, p in enumerate(tp["rnn"]):
            c_prev, h_prev = state[l]
            h, c = cell(inp, c_prev, h_prev, p["w_x"], p["w_h"], p["bias"])
            new_state.append((live * c + (1 - live) * c_prev, live * h + (1 - live) * h_prev))
            inp = h
        outs.append(live * inp)
        state = new_state
    rnn_out = torch.stack(outs, dim=1)
    if head_pool > 1:
        hs = P // head_pool
        rnn_out = rnn_out.reshape(B, T, hs, head_pool, hs, head_pool, Fh).mean(dim=(3, 5))
    logits = rnn_out.reshape(B * T, -1) @ tp["dense"]["w"] + tp["dense"]["b"]
    return logits.reshape(B, T, -1), state
