"""CPU oracle for the ConvLSTM vertex-prediction path, float64 NumPy.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package imports this module; tests, smoke() and the CPU-baseline leg of the
benchmark use it as the checker for the HIP kernels.

PARITY UNPINNED BY THE REFERENCE: this path needs TensorFlow 1.x (tf.contrib), which is not
importable next to the reference, and the reference ships no tests, golden vectors or saved outputs
for it.  The functions below restate the reference's graph from its source text and TensorFlow's
documented semantics (SAME padding, NHWC, HWIO kernels, cross-correlation, dynamic_rnn masking,
first-index argmax, sparse softmax cross-entropy).  tests/test_oracle_convlstm.py cross-checks the
restatement against torch.nn.functional.conv2d / autograd in float64 and against closed-form cases.

What it follows
---------------
* cell                     reference convLSTM.py:26-45  (gate order j, i, f, o; forget_bias unused)
* stacked recurrence       reference polyrnn/History2LargeExperiment.py:31-46 (MultiRNNCell +
                           tf.nn.dynamic_rnn(sequence_length=duration, time_major=True))
* encoder / tile / concat  reference polyrnn/History2LargeExperiment.py:15-25 and the per-experiment
                           variants (RNNConvExperiment.py:14-23, EndTokenDeepExperiment_2.py:21-31)
* dense head               reference polyrnn/History2LargeExperiment.py:48-50,
                           polyrnn/EndTokenDeepExperiment_2.py:53-57 (P*P+1 units)
* targets, argmax, metrics, loss   reference polyrnn/Model.py:36-54, 91-100
* REINFORCE loss           reference polyrl/ReinforceOverfitExperiment.py:42-48
"""
import numpy as np


# ---------------------------------------------------------------------------------------------
# convolution (TensorFlow 'SAME', stride 1, NHWC input, HWIO kernel, cross-correlation)
# ---------------------------------------------------------------------------------------------
def _same_pads(k):
    total = k - 1
    return total // 2, total - total // 2


def im2col(x, kh, kw):
    """[B,H,W,C] -> [B,H,W,kh*kw*C] patches, tap-major then channel (matches w.reshape(kh*kw*C, O))."""
    B, H, W, C = x.shape
    pt, pb = _same_pads(kh)
    pl, pr = _same_pads(kw)
    xp = np.zeros([B, H + pt + pb, W + pl + pr, C], dtype=np.float64)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    cols = np.empty([B, H, W, kh * kw, C], dtype=np.float64)
    for dy in range(kh):
        for dx in range(kw):
            cols[:, :, :, dy * kw + dx, :] = xp[:, dy:dy + H, dx:dx + W, :]
    return cols.reshape(B, H, W, kh * kw * C)


def col2im(dcols, kh, kw, C):
    """Adjoint of im2col: [B,H,W,kh*kw*C] -> [B,H,W,C]."""
    B, H, W, _ = dcols.shape
    pt, pb = _same_pads(kh)
    pl, pr = _same_pads(kw)
    dxp = np.zeros([B, H + pt + pb, W + pl + pr, C], dtype=np.float64)
    d = dcols.reshape(B, H, W, kh * kw, C)
    for dy in range(kh):
        for dx in range(kw):
            dxp[:, dy:dy + H, dx:dx + W, :] += d[:, :, :, dy * kw + dx, :]
    return dxp[:, pt:pt + H, pl:pl + W, :]


def conv2d_same(x, w, b=None):
    kh, kw, C, O = w.shape
    y = im2col(np.asarray(x, np.float64), kh, kw) @ np.asarray(w, np.float64).reshape(kh * kw * C, O)
    if b is not None:
        y = y + np.asarray(b, np.float64)
    return y


def conv2d_same_backward(x, w, dy):
    """Returns (dx, dw, db) of y = conv2d_same(x, w, b)."""
    kh, kw, C, O = w.shape
    cols = im2col(np.asarray(x, np.float64), kh, kw)
    wm = np.asarray(w, np.float64).reshape(kh * kw * C, O)
    dcols = dy @ wm.T
    dw = (cols.reshape(-1, kh * kw * C).T @ dy.reshape(-1, O)).reshape(kh, kw, C, O)
    return col2im(dcols, kh, kw, C), dw, dy.reshape(-1, O).sum(axis=0)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


# ---------------------------------------------------------------------------------------------
# the cell (reference convLSTM.py:26-45)
# ---------------------------------------------------------------------------------------------
def cell_forward(x, c_prev, h_prev, w_x, w_h, bias, return_cache=False):
    """x [B,H,W,C]; c_prev, h_prev [B,H,W,F]; w_x [3,3,C,4F]; w_h [3,3,F,4F]; bias [4F].
    Returns (h, c) - and the gate activations when return_cache."""
    y = conv2d_same(h_prev, w_h) + conv2d_same(x, w_x) + np.asarray(bias, np.float64)
    j, i, f, o = np.split(y, 4, axis=3)
    g, si, sf, so = np.tanh(j), sigmoid(i), sigmoid(f), sigmoid(o)
    c = sf * np.asarray(c_prev, np.float64) + si * g
    tc = np.tanh(c)
    h = so * tc
    if return_cache:
        return h, c, dict(x=x, h_prev=h_prev, c_prev=c_prev, g=g, i=si, f=sf, o=so, tc=tc)
    return h, c


def cell_backward(cache, w_x, w_h, dh, dc):
    """Gradients of a cell step given dL/dh and dL/dc of its outputs.
    Returns dict(dx, dh_prev, dc_prev, dw_x, dw_h, dbias, dy)."""
    g, si, sf, so, tc = cache["g"], cache["i"], cache["f"], cache["o"], cache["tc"]
    do = dh * tc
    dct = dc + dh * so * (1.0 - tc * tc)
    dzj = dct * si * (1.0 - g * g)
    dzi = dct * g * si * (1.0 - si)
    dzf = dct * np.asarray(cache["c_prev"], np.float64) * sf * (1.0 - sf)
    dzo = do * so * (1.0 - so)
    dy = np.concatenate([dzj, dzi, dzf, dzo], axis=3)
    dx, dw_x, db = conv2d_same_backward(cache["x"], w_x, dy)
    dh_prev, dw_h, _ = conv2d_same_backward(cache["h_prev"], w_h, dy)
    return dict(dx=dx, dh_prev=dh_prev, dc_prev=dct * sf, dw_x=dw_x, dw_h=dw_h, dbias=db, dy=dy)


# ---------------------------------------------------------------------------------------------
# MultiRNNCell + dynamic_rnn(sequence_length) (reference History2LargeExperiment.py:31-46)
# ---------------------------------------------------------------------------------------------
def rnn_forward(inputs, durations, layers, init_state=None, return_cache=False):
    """inputs [T,B,H,W,C] time-major; durations int [B]; layers = list of dict(w_x, w_h, bias);
    init_state = list of (c, h) per layer or None for zeros.

    dynamic_rnn semantics: for a sample with t >= duration the emitted output is zero and the
    whole state tuple is copied through unchanged.
    Returns (outputs [T,B,H,W,F], final_state list of (c, h))."""
    T, B, H, W, _ = inputs.shape
    F = layers[0]["w_h"].shape[2]
    durations = np.asarray(durations).astype(np.int64)
    state = [(np.zeros([B, H, W, F]), np.zeros([B, H, W, F])) for _ in layers] if init_state is None else \
        [(np.asarray(c, np.float64), np.asarray(h, np.float64)) for c, h in init_state]
    outputs = np.zeros([T, B, H, W, F])
    caches = []
    for t in range(T):
        live = (t < durations).astype(np.float64)[:, None, None, None]
        x = np.asarray(inputs[t], np.float64)
        step_cache = []
        new_state = []
        for l, p in enumerate(layers):
            c_prev, h_prev = state[l]
            h, c, cache = cell_forward(x, c_prev, h_prev, p["w_x"], p["w_h"], p["bias"], return_cache=True)
            step_cache.append(cache)
            new_state.append((live * c + (1 - live) * c_prev, live * h + (1 - live) * h_prev))
            x = h
        outputs[t] = live * x
        state = new_state
        caches.append(step_cache)
    if return_cache:
        return outputs, state, caches
    return outputs, state


def rnn_backward(caches, durations, layers, d_outputs, d_final_state=None):
    """Back-propagation through rnn_forward. d_outputs [T,B,H,W,F]; d_final_state list of (dc, dh)
    or None.  Returns (d_inputs [T,B,H,W,C], grads list of dict(dw_x, dw_h, dbias), d_init_state)."""
    T = len(caches)
    L = len(layers)
    durations = np.asarray(durations).astype(np.int64)
    B, H, W, F = d_outputs.shape[1:]
    dstate = [(np.zeros([B, H, W, F]), np.zeros([B, H, W, F])) for _ in layers] if d_final_state is None else \
        [(np.asarray(dc, np.float64), np.asarray(dh, np.float64)) for dc, dh in d_final_state]
    grads = [dict(dw_x=np.zeros_like(p["w_x"], dtype=np.float64), dw_h=np.zeros_like(p["w_h"], dtype=np.float64),
                  dbias=np.zeros_like(p["bias"], dtype=np.float64)) for p in layers]
    C = layers[0]["w_x"].shape[2]
    d_inputs = np.zeros([T, B, H, W, C])
    for t in reversed(range(T)):
        live = (t < durations).astype(np.float64)[:, None, None, None]
        d_from_above = live * d_outputs[t]  # gradient wrt the top layer's h as an OUTPUT
        new_dstate = [None] * L
        for l in reversed(range(L)):
            dc_s, dh_s = dstate[l]
            dh = live * dh_s + d_from_above
            dc = live * dc_s
            r = cell_backward(caches[t][l], layers[l]["w_x"], layers[l]["w_h"], dh, dc)
            grads[l]["dw_x"] += r["dw_x"]
            grads[l]["dw_h"] += r["dw_h"]
            grads[l]["dbias"] += r["dbias"]
            new_dstate[l] = (r["dc_prev"] + (1 - live) * dc_s, r["dh_prev"] + (1 - live) * dh_s)
            d_from_above = r["dx"]
        d_inputs[t] = d_from_above
        dstate = new_dstate
    return d_inputs, grads, dstate


# ---------------------------------------------------------------------------------------------
# heads, loss, metrics (reference polyrnn/Model.py:36-54, 91-100)
# ---------------------------------------------------------------------------------------------
def targets_dense(targets, prediction_size):
    """x * P + y (Model.py:39); the end token (P, 0) lands on class P*P."""
    targets = np.asarray(targets).astype(np.int64)
    return targets[..., 0] * prediction_size + targets[..., 1]


def softmax_xent(logits, labels):
    """Per-row sparse softmax cross-entropy; logits [..., K], labels [...] int."""
    z = np.asarray(logits, np.float64)
    m = z.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(z - m).sum(axis=-1))
    picked = np.take_along_axis(z, np.asarray(labels).astype(np.int64)[..., None], axis=-1)[..., 0]
    return lse - picked


def softmax_xent_backward(logits, labels, drows):
    """d/dlogits of sum(drows * xent_rows)."""
    z = np.asarray(logits, np.float64)
    p = np.exp(z - z.max(axis=-1, keepdims=True))
    p /= p.sum(axis=-1, keepdims=True)
    onehot = np.zeros_like(p)
    np.put_along_axis(onehot, np.asarray(labels).astype(np.int64)[..., None], 1.0, axis=-1)
    return (p - onehot) * np.asarray(drows, np.float64)[..., None]


def prediction_max(logits, prediction_size):
    """argmax over classes (first maximum wins) -> (x, y) = (a // P, a % P) (Model.py:44, 93-94)."""
    a = np.argmax(logits, axis=-1)
    return np.stack([a // prediction_size, a % prediction_size], axis=-1)


def metrics(logits, targets, durations, prediction_size):
    """accuracy, avg_error, max_error exactly as Model.py:42-54 defines them."""
    B, T = logits.shape[:2]
    durations = np.asarray(durations).astype(np.int64)
    mask = np.arange(T)[None, :] < durations[:, None]
    num = durations.sum()
    dense = targets_dense(targets, prediction_size)
    a = np.argmax(logits, axis=-1)
    accuracy = np.count_nonzero((dense == a) & mask) / num
    err = np.linalg.norm((np.asarray(targets).astype(np.int64) - prediction_max(logits, prediction_size))
                         .astype(np.float64), axis=2) * mask.astype(np.float64)
    return accuracy, err.sum() / float(num), err.max()


# ---------------------------------------------------------------------------------------------
# whole model: encoder -> tile -> concat history -> stacked ConvLSTM -> dense head
# ---------------------------------------------------------------------------------------------
def encoder_forward(images, encoder):
    """encoder = list of dict(w [k,k,Cin,Cout], b [Cout]); conv 'same' + ReLU each."""
    x = np.asarray(images, np.float64)
    acts = []
    for p in encoder:
        pre = conv2d_same(x, p["w"], p["b"])
        acts.append((x, pre))
        x = np.maximum(pre, 0.0)
    return x, acts


def model_forward(params, images, histories, durations, init_state=None, return_cache=False):
    """params = dict(encoder=[...], rnn=[...], dense=dict(w [P*P*F, K], b [K])).
    images [B,P,P,3]; histories [B,T,P,P,hist]; returns logits [B,T,K] (+ final state, cache)."""
    feat, enc_acts = encoder_forward(images, params["encoder"])
    B, T = histories.shape[:2]
    tiled = np.broadcast_to(feat[:, None], (B, T) + feat.shape[1:])
    concat = np.concatenate([tiled, np.asarray(histories, np.float64)], axis=4)
    rnn_in = np.transpose(concat, (1, 0, 2, 3, 4))
    outs, final_state, caches = rnn_forward(rnn_in, durations, params["rnn"], init_state, return_cache=True)
    rnn_out = np.transpose(outs, (1, 0, 2, 3, 4))
    flat = rnn_out.reshape(B * T, -1)
    logits = (flat @ np.asarray(params["dense"]["w"], np.float64) + np.asarray(params["dense"]["b"], np.float64))
    logits = logits.reshape(B, T, -1)
    if return_cache:
        return logits, final_state, dict(enc_acts=enc_acts, caches=caches, flat=flat, feat=feat)
    return logits, final_state


def model_loss(params, images, histories, durations, targets, prediction_size, weights=None):
    """Mean over ALL B*T rows of the cross-entropy (NOT masked by duration, Model.py:98-100).
    With ``weights`` [B,T] it is the REINFORCE loss mean(xent * td_target)
    (ReinforceOverfitExperiment.py:47-48)."""
    logits, _ = model_forward(params, images, histories, durations)
    rows = softmax_xent(logits, targets_dense(targets, prediction_size))
    if weights is not None:
        rows = rows * np.asarray(weights, np.float64)
    return rows.mean(), logits


def model_loss_and_grads(params, images, histories, durations, targets, prediction_size, weights=None):
    """Loss and gradients wrt every parameter (same nesting as ``params``)."""
    logits, _, cache = model_forward(params, images, histories, durations, return_cache=True)
    B, T, K = logits.shape
    labels = targets_dense(targets, prediction_size)
    rows = softmax_xent(logits, labels)
    w = np.ones([B, T]) if weights is None else np.asarray(weights, np.float64)
    loss = (rows * w).mean()
    dlogits = softmax_xent_backward(logits, labels, w / (B * T)).reshape(B * T, K)
    flat = cache["flat"]
    g_dense = dict(w=flat.T @ dlogits, b=dlogits.sum(axis=0))
    dflat = dlogits @ np.asarray(params["dense"]["w"], np.float64).T
    feat = cache["feat"]
    P, F = feat.shape[1], params["rnn"][0]["w_h"].shape[2]
    d_out = np.transpose(dflat.reshape(B, T, P, P, F), (1, 0, 2, 3, 4))
    d_in, g_rnn, _ = rnn_backward(cache["caches"], durations, params["rnn"], d_out)
    Cenc = feat.shape[3]
    dfeat = d_in[..., :Cenc].sum(axis=0)
    g_enc = []
    d = dfeat
    for p, (x, pre) in zip(reversed(params["encoder"]), reversed(cache["enc_acts"])):
        d = d * (pre > 0)
        dx, dw, db = conv2d_same_backward(x, p["w"], d)
        g_enc.append(dict(w=dw, b=db))
        d = dx
    g_enc.reverse()
    return loss, logits, dict(encoder=g_enc, rnn=[dict(w_x=g["dw_x"], w_h=g["dw_h"], bias=g["dbias"]) for g in g_rnn],
                              dense=g_dense)


def discount_rewards(r, gamma=0.99):
    """Reverse cumulative discounted return (reference polyrl/MonteCarloPolicyGradient.py:13-22)."""
    r = np.asarray(r, np.float64)
    out = np.zeros_like(r)
    running = 0.0
    for t in reversed(range(r.size)):
        running = running * gamma + r[t]
        out[t] = running
    return out


def adam_step(param, grad, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """TensorFlow's Adam form (Model.py:110): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps).
    ``step`` counts from 1. Returns (param, m, v)."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    lr_t = lr * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    return param - lr_t * m / (np.sqrt(v) + eps), m, v


def clip_by_global_norm(grads, clip_norm=1.0):
    """tf.clip_by_global_norm: scale = clip / max(global_norm, clip)."""
    gn = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    scale = clip_norm / max(gn, clip_norm)
    return [g * scale for g in grads], gn
