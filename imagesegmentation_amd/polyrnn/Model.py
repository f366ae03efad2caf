"""``Model``: the reference's estimator base (reference polyrnn/Model.py:16-274) on MI355X.

Same constructor and method surface - ``Model(sess, max_timesteps, image_size, prediction_size,
history_length, filepath)``, ``train / validate / validate_iou / save / maybe_restore`` - and the same
graph semantics: CNN encoder -> tile over time -> concat history -> MultiRNNCell(ConvLSTM) under
dynamic_rnn -> dense logits over the P x P (+1) grid; x-major class index; argmax decode; accuracy and
L2 errors masked by duration; loss = mean cross-entropy over ALL B*T rows (padded steps included);
clip_by_global_norm(1.0) + Adam(1e-4) in TensorFlow's form.

There is no session or placeholder: ``sess`` is accepted and ignored, arrays are passed to the
methods directly, and the tensors the reference exposes as graph nodes (``prediction_logits``,
``prediction_max``, ``loss``, ``rnn_final_state`` ...) hold the values of the last evaluation.
Subclasses describe their graph in ``_build_graph`` through ``self.encoder_spec``, ``self.rnn_layers``
and ``self.end_token`` (see polyrnn/experiments.py).

The recurrence runs in the HIP kernels of libisg (convLSTM.dynamic_rnn); the encoder convolutions and
the dense head are plain torch ops (MIOpen / hipBLASLt).
"""
import contextlib
import json
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops, parallel, util
from ..optim import adam_tf_step_, flatten_parameters_
from ..convLSTM import ConvLSTMCell, LSTMStateTuple, MultiRNNCell, RecurrentStepper, SequenceInput, dynamic_rnn
from . import decode


def _glorot(shape, device, gen):
    if len(shape) == 4:
        fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
    else:
        fan_in, fan_out = shape
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, device=device, generator=gen) * 2 - 1) * limit


class _BiasRelu(torch.autograd.Function):
    """max(z + bias, 0) in place over the convolution's own output tensor (NCHW-shaped, channels-last
    memory = NHWC; isg_bias_relu); the backward masks the gradient and sums the bias gradient in one
    pass (isg_relu_bwd_bias).  The tensor handed in must be the convolution's result itself, not a
    view of it: an in-place op on a view makes autograd copy whole activations."""

    @staticmethod
    def forward(ctx, z, bias):
        assert z.is_contiguous(memory_format=torch.channels_last)
        ops.bias_relu_(z.permute(0, 2, 3, 1), bias.detach().float().contiguous())
        ctx.mark_dirty(z)
        ctx.save_for_backward(z)
        ctx.bias_dtype = bias.dtype
        return z

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dz, dbias = ops.relu_bwd_bias(dy.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1))
        return dz.permute(0, 3, 1, 2), dbias.to(ctx.bias_dtype)


class _ConvIn(torch.autograd.Function):
    """First encoder layer (image -> 32 or 16 channels, SAME, bias, ReLU) on the isg_conv_in kernels.  The
    image gets no gradient (it is the network's input)."""

    @staticmethod
    def forward(ctx, x3, w_hwio, bias):
        ks = w_hwio.shape[0]
        x4 = ops.conv_in_pad_image(x3.contiguous())
        y = ops.conv_in_fwd(x4, ops.conv_in_pack(w_hwio), bias.detach().float().contiguous(), ks, relu=True)
        ctx.save_for_backward(x4, y)
        ctx.ks, ctx.dtypes = ks, (w_hwio.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x4, y = ctx.saved_tensors
        dz, dbias = ops.relu_bwd_bias(dy.contiguous(), y)
        dw = ops.conv_in_wgrad(x4, dz, ctx.ks)
        return None, dw.to(ctx.dtypes[0]), dbias.to(ctx.dtypes[1])


class _Conv32(torch.autograd.Function):
    """A 32 -> 32 or 16 -> 16 channel encoder layer (SAME, bias, ReLU) on the isg_conv32 / isg_conv16 kernels: forward with the
    bias and ReLU fused, backward = ReLU mask + bias gradient, data gradient, weight gradient."""

    @staticmethod
    def forward(ctx, x, w_hwio, bias):
        ks = w_hwio.shape[0]
        x = x.contiguous()
        y = ops.conv32(x, ops.conv32_pack(w_hwio), bias.detach().float().contiguous(), ks, relu=True)
        ctx.save_for_backward(x, y, w_hwio)
        ctx.ks, ctx.dtypes = ks, (w_hwio.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w_hwio = ctx.saved_tensors
        dz, dbias = ops.relu_bwd_bias(dy.contiguous(), y)
        dx = ops.conv32(dz, ops.conv32_pack(w_hwio, data_gradient=True), None, ctx.ks, relu=False) \
            if ctx.needs_input_grad[0] else None
        dw = ops.conv32_wgrad(x, dz, ctx.ks)
        return dx, dw.to(ctx.dtypes[0]), dbias.to(ctx.dtypes[1])


class _Conv32Padded(torch.autograd.Function):
    """A layer with fewer input channels than its 32 (16) filters on the isg_conv32 (isg_conv16) kernels: the input is
    padded with zero channels (reference polyrl/a3c/model.py:19-20, whose first layer sees 3 image channels + the
    history frames).  The input gets no gradient (it is an observation)."""

    @staticmethod
    def forward(ctx, x, w_hwio, bias):
        ks, cin, c = w_hwio.shape[0], w_hwio.shape[2], w_hwio.shape[3]
        xp = x.contiguous() if x.shape[-1] == c else ops.pad_channels(x, c)  # (c channels: padded by the caller)
        y = ops.conv32(xp, ops.conv32_pack_padded(w_hwio), bias.detach().float().contiguous(), ks, relu=True)
        ctx.save_for_backward(xp, y)
        ctx.ks, ctx.cin, ctx.dtypes = ks, cin, (w_hwio.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, y = ctx.saved_tensors
        dz, dbias = ops.relu_bwd_bias(dy.contiguous(), y)
        dw = ops.conv32_wgrad(xp, dz, ctx.ks)[:, :, :ctx.cin]
        return None, dw.to(ctx.dtypes[0]), dbias.to(ctx.dtypes[1])


class _EncoderChain(torch.autograd.Function):
    """A whole encoder of hand-written layers - the first on the isg_conv_in kernels (a 3-channel image) or, with its
    input padded with zero channels, on isg_conv32 / isg_conv16 (reference polyrl/a3c/model.py:19-20: image +
    history frames), the others on isg_conv32 / isg_conv16 - as one autograd node (reference
    polyrnn/History2LargeExperiment.py:15-22, polyrnn/RNNConvExperiment.py:14-22).  Seen as a chain, the backward pass needs the separate ReLU-mask and
    bias-gradient pass only for the top layer: every data gradient lands on the ReLU output of the layer below
    and applies that layer's mask and sums its bias gradient in its own epilogue (isg_conv32_masked)."""

    @staticmethod
    def forward(ctx, x3, *params):
        ws, bs = params[0::2], params[1::2]
        ctx.padded_first = not ops.conv_in_supported(ws[0], x3.dtype, x3.device)
        if ctx.padded_first:
            x4 = x3.contiguous() if x3.shape[-1] == ws[0].shape[3] else ops.pad_channels(x3, ws[0].shape[3])
            ys = [ops.conv32(x4, ops.conv32_pack_padded(ws[0]), bs[0].detach().float().contiguous(), ws[0].shape[0],
                             relu=True)]
        else:
            x4 = ops.conv_in_pad_image(x3.contiguous())
            ys = [ops.conv_in_fwd(x4, ops.conv_in_pack(ws[0]), bs[0].detach().float().contiguous(), ws[0].shape[0],
                                  relu=True)]
        for w, b in zip(ws[1:], bs[1:]):
            ys.append(ops.conv32(ys[-1], ops.conv32_pack(w), b.detach().float().contiguous(), w.shape[0], relu=True))
        ctx.save_for_backward(x4, *ys, *ws)
        ctx.biases = [b.detach() for b in bs]  # (only their addresses are used: where their gradients belong)
        ctx.n = len(ws)
        ctx.dtypes = [p.dtype for p in params]
        return ys[-1]

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n
        x4, ys, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:]
        bs = ctx.biases
        grads = [None] * (2 * n)
        # a gradient whose parameter has a sink (ops.grad_sinks) is written there by the kernel itself and autograd
        # gets None for it
        dst = lambda p: ops.grad_destination(p)

        def hand(slot, param, value, to):
            if dst(param) is not None:
                ops.grad_delivered(param)
            else:
                grads[slot] = value.to(to)
        # (a gradient that is a channel slice of wider pixels - the recurrent cell's joint input gradient - is read in
        # place)
        dz, dbias = ops.relu_bwd_bias(dy if ops.pixel_stride(dy) is not None else dy.contiguous(), ys[-1],
                                      dbias_out=dst(bs[n - 1]))
        for k in range(n - 1, 0, -1):
            ks = ws[k].shape[0]
            hand(2 * k, ws[k], ops.conv32_wgrad(ys[k - 1], dz, ks, out=dst(ws[k])), ctx.dtypes[2 * k])
            hand(2 * k + 1, bs[k], dbias, ctx.dtypes[2 * k + 1])
            dz, dbias = ops.conv32_masked(dz, ops.conv32_pack(ws[k], data_gradient=True), ys[k - 1], ks,
                                          dbias_out=dst(bs[k - 1]))
        if ctx.padded_first:  # (the padded kernel's gradient has rows for the zero channels too: no sink for it)
            grads[0] = ops.conv32_wgrad(x4, dz, ws[0].shape[0])[:, :, :ws[0].shape[2]].to(ctx.dtypes[0])
        else:
            hand(0, ws[0], ops.conv_in_wgrad(x4, dz, ws[0].shape[0], out=dst(ws[0])), ctx.dtypes[0])
        hand(1, bs[0], dbias, ctx.dtypes[1])
        return (None,) + tuple(grads)


class _RecordedEncoder(torch.autograd.Function):
    """``_EncoderChain`` (padded first layer) over a forward pass that has already run: ``recorded`` = (padded input,
    the layers' outputs), each [rows, H, W, C] - a rollout's acting steps ran these very launches on these very
    observations with the present weights and kept their outputs (polyrl/a3c: ``LSTMPolicy._encode(keep=)``).  The
    forward hands out the top layer's kept output; the backward is the chain's."""

    @staticmethod
    def forward(ctx, recorded, *params):
        x4, ys = recorded
        ws, bs = params[0::2], params[1::2]
        ctx.padded_first = True
        ctx.save_for_backward(x4, *ys, *ws)
        ctx.biases = [b.detach() for b in bs]
        ctx.n = len(ws)
        ctx.dtypes = [p.dtype for p in params]
        return ys[-1].view_as(ys[-1])

    @staticmethod
    def backward(ctx, dy):
        return _EncoderChain.backward(ctx, dy)


def encoder_chain_supported(x_nhwc, encoder, dtype):
    """Whether ``_EncoderChain`` takes this encoder: an input that needs no gradient, a first layer for the
    isg_conv_in kernels (or, padded, for the square ones) and at least one more layer, all of those for the isg_conv32 / isg_conv16 kernels."""
    if x_nhwc.requires_grad or len(encoder) < 2 or x_nhwc.dtype != dtype or os.environ.get("ISG_NO_ENCODER_CHAIN"):
        return False  # (the environment switch: layer-by-layer nodes, for A/B runs)
    (w0, _), rest = encoder[0], encoder[1:]
    first = ops.conv_in_supported(w0, dtype, x_nhwc.device) or ops.conv32_padded_supported(w0, dtype, x_nhwc.device)
    return first and \
        all(ops.conv32_supported(w, dtype, x_nhwc.device) and w.shape[2] == w0.shape[3] for w, _ in rest)


def conv2d_same_relu(x_nhwc, w_hwio, b, dtype):
    """tf.layers.conv2d(padding='same', activation=relu) on an NHWC tensor with an HWIO kernel
    (reference polyrnn/History2LargeExperiment.py:15-22).  Even kernels pad (k-1)//2 before and the
    rest after, as TensorFlow's SAME does.  The convolution is the vendor library's; bias + ReLU are
    one fused in-place pass when the channel count allows it."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    if not x_nhwc.requires_grad and ops.conv_in_supported(w_hwio, dtype, x_nhwc.device):
        return _ConvIn.apply(x_nhwc, w_hwio, b)
    if x_nhwc.dtype == dtype and ops.conv32_supported(w_hwio, dtype, x_nhwc.device):
        return _Conv32.apply(x_nhwc, w_hwio, b)
    if x_nhwc.dtype == dtype and not x_nhwc.requires_grad and ops.conv32_padded_supported(w_hwio, dtype, x_nhwc.device):
        return _Conv32Padded.apply(x_nhwc, w_hwio, b)
    x = x_nhwc.permute(0, 3, 1, 2)  # NCHW view of channels-last memory
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    if kh % 2 == 0 or kw % 2 == 0:
        x = F.pad(x, (pl, kw - 1 - pl, pt, kh - 1 - pt))
        pad = 0
    else:
        pad = (pt, pl)
    w = w_hwio.to(dtype).permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last)
    cout = w_hwio.shape[3]
    vec = 8 if dtype == torch.bfloat16 else 4
    if x.is_cuda and cout % vec == 0 and 256 % (cout // vec) == 0:
        z = F.conv2d(x, w, None, padding=pad)
        if z.is_contiguous(memory_format=torch.channels_last):
            return _BiasRelu.apply(z, b).permute(0, 2, 3, 1)
        return torch.relu(z + b.to(dtype).view(1, -1, 1, 1)).permute(0, 2, 3, 1).contiguous()
    y = F.relu(F.conv2d(x, w, b.to(dtype), padding=pad))
    return y.permute(0, 2, 3, 1).contiguous()


def _head_weights(dt, w, b, cache):
    """The dense head's weights in the compute type: copies that are converted again when the weights change - by
    the trainer's one refresh launch per step when a derived-weights registry is active (ops.DerivedWeights)."""
    reg = ops.active_derived()
    rkey = ("head", w.data_ptr(), b.data_ptr(), dt)
    if reg is not None:
        e = reg.get(rkey)
        if e is not None:
            return e.w, e.b
    key = (dt, w._version, b._version, w.data_ptr(), ops.weights_epoch())
    if cache.get("key") != key:
        cache["key"], cache["w"], cache["b"] = key, w.detach().to(dt), b.detach().to(dt)
        if reg is not None and dt == torch.bfloat16 and reg.owns(w.detach(), b.detach()):
            import types
            ns = types.SimpleNamespace(w=cache["w"], b=cache["b"])
            reg.record(rkey, ns, ["w", "b"], lambda: (ns.w.copy_(w.detach()), ns.b.copy_(b.detach())), [w, b])
    return cache["w"], cache["b"]


def _head_weights_padded(dt, w, b, cache):
    """``_head_weights`` with the class axis padded with zero columns to a multiple of 8: a bf16 matrix whose rows are
    785 elements long (28 x 28 + 1 classes) starts most rows off a 16-byte boundary, and the library's GEMMs over it
    run a quarter slower than over 792 columns.  A copy of its own beside the unpadded one (which the few-row paths
    of the decode and acting loops keep using)."""
    K, N = w.shape
    Np = (N + 7) // 8 * 8
    reg = ops.active_derived()
    rkey = ("head8", w.data_ptr(), b.data_ptr(), dt)
    if reg is not None:
        e = reg.get(rkey)
        if e is not None:
            return e.w, e.b
    key = (dt, w._version, b._version, w.data_ptr(), ops.weights_epoch())
    if cache.get("key8") != key:
        wp = torch.zeros([K, Np], dtype=dt, device=w.device)
        bp = torch.zeros([Np], dtype=dt, device=w.device)
        wp[:, :N].copy_(w.detach())
        bp[:N].copy_(b.detach())
        cache["key8"], cache["w8"], cache["b8"] = key, wp, bp
        if reg is not None and dt == torch.bfloat16 and reg.owns(w.detach(), b.detach()):
            import types
            ns = types.SimpleNamespace(w=wp, b=bp)
            reg.record(rkey, ns, ["w", "b"], lambda: (ns.w[:, :N].copy_(w.detach()), ns.b[:N].copy_(b.detach())), [w, b])
    return cache["w8"], cache["b8"]


class _DenseHead(torch.autograd.Function):
    """logits = flat @ W + b (reference polyrnn/History2LargeExperiment.py:48-50) with float32 master weights and
    activations in the compute type: the library GEMM runs in the compute type on copies of the weights that are
    cached until the weights change; the weight gradient (a GEMM in the compute type) lands in the parameter's
    gradient sink with ONE converting pass when there is one (ops.grad_sinks) - otherwise autograd would convert it
    to float32 and add it to the gradient in two more passes over its 9.8 M elements."""

    @staticmethod
    def forward(ctx, flat, w, b, cache):
        N = w.shape[1]
        # (many rows against a class axis that is no multiple of 8: the GEMMs run on the padded copy)
        ctx.padded = (N % 8 != 0 and N >= 64 and flat.is_cuda and flat.dtype == torch.bfloat16 and
                      flat.shape[0] >= 2048 and not os.environ.get("ISG_NO_PADDED_HEAD"))
        wc, bc = (_head_weights_padded if ctx.padded else _head_weights)(flat.dtype, w, b, cache)
        ctx.save_for_backward(flat, wc)
        ctx.params = (w.detach(), b.detach())
        out = torch.addmm(bc, flat, wc)
        return out[:, :N].contiguous() if ctx.padded else out

    @staticmethod
    def backward(ctx, dlogits):
        flat, wc = ctx.saved_tensors
        w, b = ctx.params
        dlogits = dlogits.contiguous()
        if ctx.padded:
            N = w.shape[1]
            d = dlogits.new_zeros([dlogits.shape[0], wc.shape[1]])
            d[:, :N] = dlogits
            d_flat = d @ wc.t() if ctx.needs_input_grad[0] else None
            dw = (flat.t() @ d)[:, :N]
            db = decode.column_sums(dlogits)
            out = []
            for prm, g in ((w, dw), (b, db)):
                dst = ops.grad_destination(prm)
                if dst is not None:
                    dst.copy_(g)
                    ops.grad_delivered(prm)
                    out.append(None)
                else:
                    out.append(g.to(prm.dtype))
            return d_flat, out[0], out[1], None
        d_flat = dlogits @ wc.t() if ctx.needs_input_grad[0] else None
        if dlogits.shape[1] < 8 and dlogits.is_cuda:
            # a head of a few columns (the actor-critic value head has one): as [K, R] x [R, 1] the library takes a
            # matrix-vector path that holds the launching thread for 12 ms per call at R = 10,240, K = 32,768; the
            # transposed product is an ordinary GEMM (0.02 ms of host time, 0.14 ms instead of 0.58 on the device)
            dw = (dlogits.t() @ flat).t()
        else:
            dw = flat.t() @ dlogits
        db = decode.column_sums(dlogits) if dlogits.is_cuda else dlogits.sum(dim=0, dtype=torch.float32)
        out = []
        for prm, g in ((w, dw), (b, db)):
            dst = ops.grad_destination(prm)
            if dst is not None:
                dst.copy_(g)
                ops.grad_delivered(prm)
                out.append(None)
            else:
                out.append(g.to(prm.dtype))
        return d_flat, out[0], out[1], None


class _AvgPool(torch.autograd.Function):
    """k x k mean over the grid of [..., P, P, F] (isg_avgpool_fwd); the gradient spreads evenly."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        return ops.avgpool(x, k)

    @staticmethod
    def backward(ctx, d):
        k = ctx.k
        d = d / (k * k)
        return d.repeat_interleave(k, dim=-3).repeat_interleave(k, dim=-2), None


class _DecodeGraph:
    """The decode loop of one (model, batch size) captured as a HIP graph: inputs are copied into the graph's own
    buffers, one launch replays encoder + T x (cells, head, decode step) + the assembly and rasterisation of the
    predicted polygons, and the results are read from its buffers.  The weight images it reads are the model's shared
    buffers (ops.DerivedWeights), refreshed by one launch ahead of a replay when the weights have changed (without
    that registry the packing kernels are part of the graph and read the master weights where they live), so a
    replay after an optimiser step decodes with the new weights.  The graph's buffers are overwritten by the next replay: what is
    handed out of ``run`` are copies, except the recurrent state (a view that lives until then)."""

    def __init__(self, model, images_dev, vt, nvt):
        self.images = images_dev.clone()
        self.vt, self.nvt = vt.clone(), nvt.clone()
        # page-locked staging for the true polygons, kept: pinning a fresh host buffer per decode costs more host
        # time (50 - 80 us each) than the copies themselves
        self._stage = {id(t): torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in (self.vt, self.nvt)}
        self._staged = None  # event: the last copies out of the staging buffers are done

        def body():
            with model._derived:
                predicted, durations, pts0, stepper = model._decode_body(self.images, self.vt, self.nvt)
            return predicted, durations, pts0, stepper, model._iou_body(predicted, durations, pts0, self.vt, self.nvt)
        # one eager pass first: library handles and workspaces come into being outside the capture
        body()
        #: the weight images the loop reads live in the model's shared buffers (ops.DerivedWeights) and are brought
        #: up to date before a replay; this graph is good for as long as they stay where they are
        self.generation = model._derived_generation()
        torch.cuda.synchronize()
        # anything else cached from the weights is rebuilt inside the capture, so that the rebuild is part of the graph
        self._drop_weight_caches(model)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
            self.predicted, self.durations, self.pts0, self.stepper, self.back = body()
        # (what the capture cached belongs to the graph's memory pool and holds nothing until a replay)
        self._drop_weight_caches(model)

    @staticmethod
    def _drop_weight_caches(model):
        for cell in model.rnn.cells:
            cell._packed = None
        model._head_cache.clear()

    @staticmethod
    def _load(dst, src):
        """Host array or tensor -> the graph's buffer, converting on the way, without blocking the host."""
        if not torch.is_tensor(src):
            src = torch.from_numpy(np.ascontiguousarray(src))
            if src.numel() * src.element_size() <= (1 << 22):
                src = src.pin_memory()
        dst.copy_(src, non_blocking=True)

    def _load_staged(self, dst, src):
        stage = self._stage[id(dst)]
        if torch.is_tensor(src) or tuple(src.shape) != tuple(stage.shape):
            return self._load(dst, src)
        stage.numpy()[...] = src  # (converts to the buffer's int32 on the host)
        dst.copy_(stage, non_blocking=True)

    def run(self, images, vt, nvt):
        if self._staged is not None:
            self._staged.synchronize()  # (normally long done: the previous decode's results were read back)
        self._load(self.images, images)
        self._load_staged(self.vt, vt)
        self._load_staged(self.nvt, nvt)
        if self._staged is None:
            self._staged = torch.cuda.Event()
        self._staged.record()
        self.graph.replay()
        return self


class _StepGraph:
    """Forward pass, loss, metrics and backward pass of one training step for one input shape and one number of
    recurrence steps, captured as a HIP graph.  On small grids (the reference's 28 x 28) a step is a few hundred
    launches of a few microseconds each and the host cannot issue them as fast as the GPU retires them; replayed
    as a graph they cost one launch.  Inputs are copied into the graph's buffers; the gradients land in the
    model's flat bucket (the backward passes write there, ops.grad_sinks); the exchange between ranks, the clip
    and Adam follow outside the graph.  The weight images and the head's weight copies are the model's shared
    buffers (ops.DerivedWeights), refreshed by one launch ahead of a replay when the weights have changed; without
    that registry the packing launches are part of the graph and read the master weights where they live."""

    def __init__(self, model, images, durations, points, targets, steps):
        dev = model.device
        self.images = torch.zeros(images.shape, dtype=model.compute_dtype, device=dev)
        self.durations = torch.zeros(durations.shape, dtype=torch.int32, device=dev)
        self.points = torch.zeros(points.shape, dtype=torch.int32, device=dev)
        self.targets = torch.zeros(targets.shape, dtype=torch.int32, device=dev)
        self.steps = steps
        self.load(images, durations, points, targets)
        model._step_body(self.images, self.durations, self.points, self.targets, steps)  # eager: workspaces, handles
        self.generation = model._derived_generation()  # (see _DecodeGraph)
        torch.cuda.synchronize()
        _DecodeGraph._drop_weight_caches(model)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
            self.loss, self.metrics = model._step_body(self.images, self.durations, self.points, self.targets, steps)
        _DecodeGraph._drop_weight_caches(model)

    def load(self, images, durations, points, targets):
        for dst, src in ((self.images, images), (self.durations, durations), (self.points, points),
                         (self.targets, targets)):
            _DecodeGraph._load(dst, src)


class Model:
    #: overridden per experiment (reference: each *Experiment.py's ExperimentModel._build_graph)
    encoder_spec = ((16, 5), (16, 5), (16, 3))
    rnn_layers = 2
    rnn_filters = 16
    end_token = False
    scope_name = 'Estimator'

    def __init__(self, sess, max_timesteps, image_size, prediction_size, history_length, filepath,
                 compute_dtype=torch.float32, device=None, seed=0, head_size=None, learning_rate=1e-4,
                 process_group=None):
        self.sess = sess  # accepted for drop-in compatibility; there is no session
        self.max_timesteps = max_timesteps
        self.image_size = image_size
        self.prediction_size = prediction_size
        self.history_length = history_length
        self._filepath = filepath
        if image_size != prediction_size:
            # the reference concatenates image features and history frames on the channel axis, which
            # needs equal grids (polyrnn/History2LargeExperiment.py:25)
            raise ValueError("image_size and prediction_size must match, as the reference's concat requires")
        self.device = torch.device(device) if device is not None else util._device()
        self.compute_dtype = compute_dtype
        #: side of the grid the dense head predicts on; the reference always uses prediction_size.
        #: A smaller value average-pools the recurrent output first (needed beyond ~64: the
        #: reference's dense(P*P*16 -> P*P) has 4e10 weights at P = 224).
        self.head_size = prediction_size if head_size is None else head_size
        if prediction_size % self.head_size != 0:
            raise ValueError("head_size must divide prediction_size")
        self.learning_rate = learning_rate
        self.process_group = process_group
        self.global_step = 0
        #: Adam's own step count (TensorFlow's beta1_power / beta2_power).  Not the same thing as global_step:
        #: the reference's Saver holds the trainable variables and global_step only (polyrnn/Model.py:132-136), so
        #: after a restore the moments and the bias correction start afresh while global_step continues.
        self._adam_t = 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._named = {}
        self._build_graph()
        self._make_variables()
        self.trainable_variables = [p for _, p in self._named_parameters()]
        # parameters, gradients and both Adam moments each live in one flat buffer: one all-reduce, one norm,
        # one update launch
        self._flat_param = flatten_parameters_(self.trainable_variables)
        #: images and copies derived from the parameters, brought up to date by one launch per optimiser step
        #: (ops.DerivedWeights; ISG_NO_DERIVED=1: every consumer packs for itself, for A/B runs)
        self._derived = ops.DerivedWeights(self._flat_param) \
            if self.device.type == "cuda" and not os.environ.get("ISG_NO_DERIVED") else contextlib.nullcontext()
        self._grad_one = None
        self._adam_m = torch.zeros_like(self._flat_param)
        self._adam_v = torch.zeros_like(self._flat_param)
        n = self._flat_param.numel()
        self._flat_grad = torch.zeros([n], dtype=torch.float32, device=self.device)
        off = 0
        head_at = None
        for p in self.trainable_variables:
            if p is self.dense_kernel:
                head_at = off
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        # the dense head (kernel, then bias) closes the bucket: its share of the all-reduce starts as soon as
        # its gradient exists, under the recurrence's backward pass
        assert head_at is not None and head_at + self.dense_kernel.numel() + self.dense_bias.numel() == n
        self._early_reduce = parallel.EarlyTailReduce(self._flat_grad, [self.dense_kernel, self.dense_bias], head_at,
                                                      process_group)
        self._events = None
        self._head_cache = {}
        self._decode_graph_cache = {}
        self._step_graph_cache = {}
        # where each parameter's gradient belongs: its view of the flat bucket (ops.grad_sinks)
        self._sinks = {p: p.grad for p in self.trainable_variables}
        # values of the last evaluation (graph nodes in the reference)
        self.prediction_logits = None
        self.loss = None
        self.rnn_final_state = None
        self.rnn_initial_state_pl = None
        self.last_metrics = {}

    # -- graph description ---------------------------------------------------------------------
    def _build_graph(self):
        """Subclasses set encoder_spec / rnn_layers / end_token here (or as class attributes)."""

    @property
    def num_classes(self):
        return self.head_size ** 2 + (1 if self.end_token else 0)

    def _make_variables(self):
        dev, gen = self.device, self._gen
        self.encoder = []
        cin = 3
        for n, (filters, k) in enumerate(self.encoder_spec):
            w = torch.nn.Parameter(_glorot((k, k, cin, filters), dev, gen))
            b = torch.nn.Parameter(torch.zeros([filters], device=dev))
            self.encoder.append((w, b))
            cin = filters
        P = self.prediction_size
        cells = []
        for _ in range(self.rnn_layers):
            cells.append(ConvLSTMCell(shape=[P, P], filters=self.rnn_filters, kernel=[3, 3],
                                      initializer=lambda t: t.copy_(_glorot(tuple(t.shape), dev, gen)),
                                      compute_dtype=self.compute_dtype, device=dev))
        self.rnn = MultiRNNCell(cells)
        self.rnn.build(cin + self.history_length, dev)
        self.feature_channels = cin
        hs = self.head_size
        self.dense_kernel = torch.nn.Parameter(_glorot((hs * hs * self.rnn_filters, self.num_classes), dev, gen))
        self.dense_bias = torch.nn.Parameter(torch.zeros([self.num_classes], device=dev))

    def _named_parameters(self):
        """(TensorFlow-style variable name, parameter) pairs, in creation order."""
        s = self.scope_name
        out = []
        for n, (w, b) in enumerate(self.encoder):
            out += [("%s/conv%d/kernel" % (s, n + 1), w), ("%s/conv%d/bias" % (s, n + 1), b)]
        for l, cell in enumerate(self.rnn.cells):
            base = "%s/rnn/rnn/multi_rnn_cell/cell_%d/ConvLSTMCell/" % (s, l)
            out += [(base + "hidden_to_state_kernel", cell.hidden_to_state_kernel),
                    (base + "input_to_state_kernel", cell.input_to_state_kernel), (base + "bias", cell.bias)]
        out += [("%s/prediction/kernel" % s, self.dense_kernel), ("%s/prediction/bias" % s, self.dense_bias)]
        return out

    # -- forward -------------------------------------------------------------------------------
    @property
    def rnn_zero_state(self):
        return lambda batch_size: self.rnn.zero_state(batch_size, self.device)

    def _to_dev(self, a, dtype):
        if torch.is_tensor(a):
            return a.to(device=self.device, dtype=dtype)
        return ops.upload(a, self.device, dtype)

    def _sequence_input(self, features, histories):
        """histories: dense [B,T,P,P,hist] (the reference's feed) or int points [B,T,hist,2]."""
        if self.history_length == 0:
            B = features.shape[0]
            pts = torch.zeros([B, self.max_timesteps, 0, 2], dtype=torch.int32, device=self.device)
            return SequenceInput(static=features, hist_points=pts)
        is_points = (torch.is_tensor(histories) and not histories.is_floating_point() and histories.dim() == 4) or \
                    (not torch.is_tensor(histories) and np.asarray(histories).ndim == 4)
        if is_points:
            return SequenceInput(static=features, hist_points=self._to_dev(histories, torch.int32))
        return SequenceInput(static=features, hist_dense=self._to_dev(histories, self.compute_dtype))

    def encode(self, images):
        x = self._to_dev(images, self.compute_dtype).contiguous()
        if torch.is_grad_enabled() and encoder_chain_supported(x, self.encoder, self.compute_dtype):
            return _EncoderChain.apply(x, *[p for wb in self.encoder for p in wb])
        for w, b in self.encoder:
            x = conv2d_same_relu(x, w, b, self.compute_dtype)
        return x

    def head(self, rnn_out):
        """rnn_out [B,T,P,P,F] -> logits [B,T,K] (reference History2LargeExperiment.py:48-50)."""
        B, T, P = rnn_out.shape[0], rnn_out.shape[1], self.prediction_size
        hs = self.head_size
        if hs != P and rnn_out.shape[2] != hs:  # not pooled by the recurrence itself
            rnn_out = _AvgPool.apply(rnn_out.contiguous(), P // hs)
        flat = rnn_out.reshape(B * T, hs * hs * self.rnn_filters)
        w, b = self.dense_kernel, self.dense_bias
        if torch.is_grad_enabled() and flat.is_cuda:
            logits = _DenseHead.apply(flat, w, b, self._head_cache)
        elif flat.dtype != w.dtype:
            # decode loops call the head once per step with unchanged weights: convert them once
            N = w.shape[1]
            if (N % 8 != 0 and N >= 64 and flat.is_cuda and flat.dtype == torch.bfloat16 and flat.shape[0] >= 2048 and
                    not os.environ.get("ISG_NO_PADDED_HEAD")):
                # (many rows: the copy padded to a multiple of 8 classes, as in ``_DenseHead``)
                wc, bc = _head_weights_padded(flat.dtype, w, b, self._head_cache)
                logits = decode.dense_few_rows(flat, wc, bc[:N]) if not os.environ.get("ISG_NO_SPLIT_HEAD") else None
                if logits is None:
                    logits = torch.addmm(bc, flat, wc)[:, :N]
            else:
                wc, bc = _head_weights(flat.dtype, w, b, self._head_cache)
                logits = decode.dense_few_rows(flat, wc, bc) if not os.environ.get("ISG_NO_SPLIT_HEAD") else None
                if logits is None:
                    logits = torch.addmm(bc, flat, wc)
        else:
            logits = torch.addmm(b.detach(), flat, w.detach()) if not torch.is_grad_enabled() else torch.addmm(b, flat, w)
        return logits.reshape(B, T, self.num_classes)

    def forward(self, images, durations, histories, initial_state=None, steps=None, recorded=None):
        """``recorded``: a pass of the recurrence over these very inputs with the present weights that has already
        run (``convLSTM.SequenceRecorder.record()`` of a rollout); the recurrence's node adopts it instead of
        launching the forward steps again."""
        features = self.encode(images)
        seq = self._sequence_input(features, histories)
        seq.recorded = recorded
        # durations already on the device (a batch from data.DeviceBatchBuilder) are not fetched back: all
        # max_timesteps steps (or ``steps``, a bound the caller knows) run under the per-sample mask
        if steps is None:
            steps = self.max_timesteps if torch.is_tensor(durations) and durations.is_cuda else None
        # the head's average pool rides inside the recurrence node: its gradient is never materialised
        rnn_out, final = dynamic_rnn(self.rnn, seq, sequence_length=durations, initial_state=initial_state,
                                     time_major=False, output_pool=self.prediction_size // self.head_size, steps=steps)
        self.rnn_final_state = final
        self.prediction_logits = self.head(rnn_out)
        return self.prediction_logits, final

    @property
    def prediction_max(self):
        """(x, y) of the argmax class of the last evaluation (reference Model.py:91-94)."""
        return decode.prediction_max(self.prediction_logits, self.head_size)

    def _loss_and_metrics(self, logits, durations, targets):
        targets = self._to_dev(targets, torch.int32)
        dur = self._to_dev(durations, torch.int32)
        B, T, K = logits.shape
        if logits.is_cuda:  # two launches: per-row pass (labels from the (x, y) targets), one summary launch
            loss, m = decode.sequence_loss(logits, targets, dur, self.head_size)
            return loss, dict(loss=loss.detach(), accuracy=m[4], avg_error=m[5], max_error=m[2])
        dense = targets[..., 0] * self.head_size + targets[..., 1]  # x-major (Model.py:39)
        pred = torch.empty([B * T], dtype=torch.int32, device=self.device)
        loss = decode.softmax_xent_mean(logits, dense, argmax_out=pred)
        with torch.no_grad():
            m = decode.sequence_metrics(pred, targets, dur, self.head_size)
            metrics = dict(loss=loss.detach(), accuracy=m[0] / m[3], avg_error=m[1] / m[3], max_error=m[2])
        return loss, metrics

    # -- optimiser (reference Model.py:106-112) --------------------------------------------------
    def _apply_gradients(self, clip=True, all_reduce=True, average=True):
        """All-reduce of the flat gradient bucket across ranks (one RCCL collective), global-norm
        clip of the reduced gradient, TensorFlow-form Adam."""
        flat = self._flat_grad
        if all_reduce:
            self._early_reduce.finish(average=average)
        norm = flat.norm()
        self.global_step += 1
        self._adam_t += 1
        # tf.clip_by_global_norm(grads, 1.0) (scale = clip / max(norm, clip)) is applied inside the update
        adam_tf_step_(self._flat_param, self._adam_m, self._adam_v, flat, self._adam_t, self.learning_rate,
                      clip_norm=1.0 if clip else None, norm=norm)
        return norm

    def _on_gradient_delivered(self, param):
        # a backward pass wrote this parameter's gradient into the flat bucket itself (no AccumulateGrad node ran,
        # so no post-accumulate hook fired): the early share of the all-reduce counts it here
        if param is self.dense_kernel or param is self.dense_bias:
            self._early_reduce._on_grad(param)

    #: True: training steps whose history arrives as vertex points are captured as HIP graphs, one per input shape
    #: and number of recurrence steps (``_StepGraph``), and replayed.  For launch-bound sizes (28 x 28 grids); at
    #: 224 x 224 the step is bound by its kernels and eager launches lose nothing.  Off by default: the eager step
    #: overlaps the head's share of the gradient exchange with the backward pass, the graph step exchanges after it.
    step_graphs = False

    def _step_body(self, images, durations, histories, targets, steps=None, arm=False):
        """Zero fill, forward, loss, metrics, backward: everything of a step up to the gradient exchange."""
        # Every parameter is used once per step, so the hand-written backward passes ASSIGN their gradients into
        # the flat bucket (ops.grad_sinks) - the dense head's 9.8 M elements included; the zero fill is for what
        # autograd still accumulates itself (layers on library kernels), i.e. everything but the head's share.
        self._flat_grad[:self._early_reduce.offset].zero_()
        with self._derived, ops.grad_sinks(self._sinks, self._on_gradient_delivered) as sinks:
            logits, _ = self.forward(images, durations, histories, steps=steps)
            loss, metrics = self._loss_and_metrics(logits, durations, targets)
            if arm:
                self._early_reduce.arm()
            if self._grad_one is None:
                self._grad_one = torch.ones([], dtype=loss.dtype, device=loss.device)
            loss.backward(gradient=self._grad_one)  # (a cached one: no fill launch per step)
        for p in (self.dense_kernel, self.dense_bias):
            if p.data_ptr() not in sinks.delivered:  # (cannot happen with the head above; kept as a guard)
                raise RuntimeError("the dense head's gradient was not delivered to the flat bucket")
        return loss.detach(), metrics

    def _derived_generation(self):
        """Compiles what the registry of derived weights recorded and says where its tensors live now (a count of
        the moves; None without a registry).  A captured graph remembers the answer."""
        if not isinstance(self._derived, ops.DerivedWeights):
            return None
        self._derived.compile()
        return self._derived.generation

    def _graph_usable(self, graph):
        """Before a replay: the graph still reads the right buffers, and they hold the current weights."""
        if not isinstance(self._derived, ops.DerivedWeights):
            return True
        if graph.generation != self._derived.generation:
            return False
        self._derived.ensure_fresh()
        return True

    def _encoder_is_native(self):
        """Whether every encoder layer runs on this library's kernels (bf16; first layer 3 -> 16 / 32 channels, the
        others 16 -> 16 / 32 -> 32): no vendor convolution anywhere in the step."""
        (w0, _), rest = self.encoder[0], self.encoder[1:]
        return self.compute_dtype == torch.bfloat16 and len(self.encoder) >= 2 and \
            ops.conv_in_supported(w0, self.compute_dtype, self.device) and \
            all(ops.conv32_supported(w, self.compute_dtype, self.device) and w.shape[2] == w0.shape[3] for w, _ in rest)

    def _graphed_step(self, images, durations, points, targets):
        """The step body as a graph replay, or None when this call has no graph form."""
        if not (torch.is_tensor(points) or isinstance(points, np.ndarray)):
            return None
        if not self._encoder_is_native():
            return None  # (the vendor convolution's backward does not survive a stream capture)
        pts_int = (not points.is_floating_point()) if torch.is_tensor(points) else points.dtype.kind in "iu"
        if not pts_int or len(points.shape) != 4:
            return None  # dense history frames: the eager path
        if torch.is_tensor(durations):
            steps = self.max_timesteps  # lengths that live on the device stay there: every step runs masked
        else:
            d = np.asarray(durations)
            steps = int(min(max(int(d.max()) if d.size else 0, 0), self.max_timesteps))
        key = (tuple(images.shape), tuple(points.shape), tuple(np.shape(targets)), steps)
        graph = self._step_graph_cache.get(key)
        if graph is not None and not self._graph_usable(graph):
            graph = None
            del self._step_graph_cache[key]
        if graph is None:
            if len(self._step_graph_cache) >= 2 * self.max_timesteps + 2:
                self._step_graph_cache.pop(next(iter(self._step_graph_cache)))
            try:
                graph = _StepGraph(self, images, durations, points, targets, steps)
            except RuntimeError as err:
                import warnings
                warnings.warn("training step not captured as a graph (%s): launching it kernel by kernel" % err)
                self.step_graphs = False
                return None
            self._step_graph_cache[key] = graph
        graph.load(images, durations, points, targets)
        graph.graph.replay()
        return graph.loss, graph.metrics

    def train_step(self, images, durations, histories, targets):
        """Forward, backward, clip, Adam. Returns the (device) loss tensor."""
        done = self._graphed_step(images, durations, histories, targets) if self.step_graphs else None
        if done is None:
            done = self._step_body(images, durations, histories, targets, arm=True)
        loss, metrics = done
        self._apply_gradients()
        if isinstance(self._derived, ops.DerivedWeights):
            self._derived.compile()  # (what this step packed for the first time joins the one refresh launch)
        self.loss = loss
        self.last_metrics = metrics
        return self.loss

    # -- reference surface -----------------------------------------------------------------------
    def train(self, images, durations, histories, targets, additional_feed_args={}):
        self.train_step(images, durations, histories, targets)
        self._add_summaries('train')

    def validate(self, images, durations, histories, targets, additional_feed_args={}):
        with torch.no_grad(), self._derived:
            logits, _ = self.forward(images, durations, histories)
            loss, metrics = self._loss_and_metrics(logits, durations, targets)
        self.loss = loss
        self.last_metrics = metrics
        self._add_summaries('valid')

    def _add_summaries(self, prefix, extra=None):
        if self._filepath is None:
            return
        if self._events is None:
            os.makedirs(self._filepath, exist_ok=True)
            self._events = open(os.path.join(self._filepath, 'events.jsonl'), 'a')
        rec = {"step": self.global_step, "time": time.time()}
        for k, v in self.last_metrics.items():
            rec["%s/%s" % (prefix, k)] = float(v)
        if extra:
            rec.update(extra)
        self._events.write(json.dumps(rec) + "\n")
        self._events.flush()

    # checkpoints: torch state dict under TensorFlow-style names (reference Model.py:132-148)
    def state_dict(self):
        sd = {name: p.detach().cpu() for name, p in self._named_parameters()}
        sd["global_step"] = torch.tensor(self.global_step)
        return sd

    def load_state_dict(self, sd):
        """Variables and global_step, as the reference's Saver restores them; the optimiser's slots are not part
        of its checkpoints, so the moments and Adam's step count restart from zero exactly as a restored
        TensorFlow graph's do."""
        with torch.no_grad():
            for name, p in self._named_parameters():
                p.copy_(sd[name].to(self.device))
            self._adam_m.zero_()
            self._adam_v.zero_()
        self._adam_t = 0
        self.global_step = int(sd["global_step"])
        ops.note_weights_changed()

    def save(self):
        """tf.train.Saver(max_to_keep=2).save(sess, filepath + 'model', global_step) (reference
        polyrnn/Model.py:132-148): one file per call named by the step, a `checkpoint` marker naming the
        newest, older files beyond two removed."""
        os.makedirs(self._filepath, exist_ok=True)
        path = os.path.join(self._filepath, 'model-%d.pt' % self.global_step)
        torch.save(self.state_dict(), path)
        with open(os.path.join(self._filepath, 'checkpoint'), 'w') as f:
            f.write(os.path.basename(path))
        keep = sorted((p for p in os.listdir(self._filepath) if p.startswith('model-') and p.endswith('.pt')),
                      key=lambda s: int(s[6:-3]))
        for old in keep[:-2]:  # max_to_keep=2
            os.remove(os.path.join(self._filepath, old))

    def maybe_restore(self):
        if not os.path.exists(self._filepath):
            os.makedirs(self._filepath)
            print('No model found at {}'.format(self._filepath))
        marker = os.path.join(self._filepath, 'checkpoint')
        if os.path.exists(marker):
            with open(marker) as f:
                name = f.read().strip()
            self.load_state_dict(torch.load(os.path.join(self._filepath, name), weights_only=True))
            print('Restored model from {} at step {}'.format(self._filepath, self.global_step))

    # -- autoregressive decode + IoU (reference Model.py:208-260) ---------------------------------
    #: which predicted vertices enter the IoU polygon: the base class rasterises all max_timesteps
    #: predictions (Model.py:253-255); the deep end-token experiment slices by duration.
    iou_slice_by_duration = False
    #: History0Experiment builds the predicted polygon from the predictions alone (its :158)
    iou_prepend_history = True

    #: Decode loops of one batch size are captured once as a HIP graph and replayed (``_DecodeGraph``); False (or
    #: ISG_NO_DECODE_GRAPH=1) launches every kernel of every vertex from the host instead.
    decode_graphs = True

    def _decode_body(self, images_dev, vt, nvt):
        """Everything between the uploaded inputs and the decoded polygons, launches only (no host round trip):
        encoder, first history points and closing vertices from the true polygons, then per vertex one fused cell
        launch per layer, the head GEMM and ONE isg_decode_step launch (argmax, stop rule / end-token rule,
        predicted polygon, history roll) - the reference's loop, polyrnn/Model.py:218-249.  Returns (predicted
        [B,T,2], durations [B], first history points [B,hist,2], stepper)."""
        B = vt.shape[0]
        T, hist, dev = self.max_timesteps, self.history_length, self.device
        with torch.no_grad():
            features = self.encode(images_dev)
            # frame i of the first step shows true_vertices[len - hist + i] (create_history with
            # end_idx = len(v) - 1, Model.py:224-226); the closing vertex is true_vertices[-hist] (Model.py:240)
            close, pts0, pts, required, durations, predicted = decode.decode_begin(vt, nvt, hist, T)
            stepper = RecurrentStepper(self.rnn, features, hist)
            for i in range(T):
                out = stepper.step(pts, durations=required, t=0)
                decode.decode_step(self.head(out)[:, 0], self.head_size, self.end_token, close, required, durations,
                                   predicted, i, pts)
        return predicted, durations, pts0, stepper

    def _decode_graph_for(self, B, image_shape, vmax):
        """The captured loop for this batch size, or None when graphs are off / the shape has none."""
        if not self.decode_graphs or os.environ.get("ISG_NO_DECODE_GRAPH") or vmax != util.VMAX_DEFAULT:
            return None
        key = (B, tuple(image_shape))
        graph = self._decode_graph_cache.get(key)
        if graph is not None and not self._graph_usable(graph):
            graph = None
            del self._decode_graph_cache[key]
        if graph is None:
            if len(self._decode_graph_cache) >= 4:  # (each holds its buffers: keep a handful of batch sizes)
                self._decode_graph_cache.pop(next(iter(self._decode_graph_cache)))
            dev = self.device
            images = torch.zeros(image_shape, dtype=self.compute_dtype, device=dev)
            vt = torch.zeros([B, vmax, 2], dtype=torch.int32, device=dev)
            vt[:, 1, 0], vt[:, 2, 1] = 1, 1  # (a valid triangle per sample for the capture's own eager pass)
            nvt = torch.full([B], 3, dtype=torch.int32, device=dev)
            try:
                graph = _DecodeGraph(self, images, vt, nvt)
            except RuntimeError as err:  # a library op inside the loop that cannot be captured
                import warnings
                warnings.warn("decode loop not captured as a graph (%s): launching it step by step" % err)
                self.decode_graphs = False
                return None
            self._decode_graph_cache[key] = graph
        return graph

    def _decode_run(self, images, true_vertices):
        """Decode + IoU counts on the device.  Returns (predicted, durations, pts0, vt, nvt, back) - ``back`` int32
        [B, 2 T + 3] = predicted vertices, durations, intersection, union per sample."""
        B = len(true_vertices)
        dev = self.device
        vt_host, nvt_host = util.pack_polygons([np.asarray(v) for v in true_vertices])
        #: device launches per decoded vertex: the cells, the head GEMM, the decode step
        self.decode_launches_per_step = len(self.rnn.cells) + 2
        shape = tuple(images.shape)
        graph = self._decode_graph_for(B, shape, vt_host.shape[1])
        if graph is not None:
            g = graph.run(images, vt_host, nvt_host)
            #: host launches of the last decode loop (graph replay: the whole loop is one)
            self.decode_host_launches = 1
            self.rnn_final_state = g.stepper.state
            return g.predicted, g.durations, g.pts0, g.vt, g.nvt, g.back
        vt = ops.upload(vt_host, dev, torch.int32)
        nvt = ops.upload(nvt_host, dev, torch.int32)
        images_dev = self._to_dev(images, self.compute_dtype).contiguous()
        with self._derived:
            predicted, durations, pts0, stepper = self._decode_body(images_dev, vt, nvt)
        self.decode_host_launches = self.max_timesteps * self.decode_launches_per_step
        self.rnn_final_state = stepper.state
        return predicted, durations, pts0, vt, nvt, self._iou_body(predicted, durations, pts0, vt, nvt)

    def _decode_on_device(self, images, true_vertices):
        """The reference's per-vertex loop (polyrnn/Model.py:218-249) on the device.  The whole loop - T x (L cell
        launches + head GEMM + decode step), the encoder and the weight packing in front of it - is ONE graph
        launch after the first decode of a batch size (``_DecodeGraph``); without graphs it is L + 2 launches per
        vertex.  No host round trip inside either way.  Returns int32 device tensors (predicted [B,T,2],
        durations [B], first history points [B,hist,2], true polygons [B,vmax,2], their vertex counts [B]); with
        graphs they are the graph's buffers, valid until the next decode of that batch size."""
        return self._decode_run(images, true_vertices)[:5]

    def _iou_body(self, predicted, durations, pts0, vt, nvt):
        """Predicted polygons assembled and rasterised against the true ones on the device (reference
        polyrnn/Model.py:252-260): int32 [B, 2 T + 3] = predicted vertices, durations, intersection, union."""
        B, T = predicted.shape[0], self.max_timesteps
        P, hist = self.prediction_size, self.history_length
        # (the polygon starts with the last `hist` true vertices where the experiment says so)
        lead = hist if self.iou_prepend_history else 0
        vmax = max(T + lead, vt.shape[1], util.VMAX_DEFAULT)
        vp, count = decode.decode_polygons(predicted, durations, pts0, P // self.head_size, lead > 0,
                                           self.iou_slice_by_duration, vmax)
        if vt.shape[1] != vmax:
            vt = torch.nn.functional.pad(vt, (0, 0, 0, vmax - vt.shape[1])).contiguous()
        inter, union = util.raster_iou(vp, count, vt, nvt, P)
        return torch.cat([predicted.reshape(B, -1), durations[:, None], inter[:, None], union[:, None]], dim=1)

    def decode(self, images, true_vertices):
        """Autoregressive decode: one recurrence step per vertex with the previous state fed back, argmax
        vertex, history shifted by np.roll(+1).  Returns (predicted_vertices [B,T,2] float64,
        prediction_durations [B]) on the host - one device-to-host copy, after the loop."""
        back = self._decode_run(images, true_vertices)[5].cpu().numpy()
        B = back.shape[0]
        return back[:, :-3].reshape(B, -1, 2).astype(np.float64), back[:, -3].astype(np.uint32)

    def validate_iou(self, images, true_vertices, summary_prefix='valid', additional_feed_args={}):
        """Decode, then mask IoU of every predicted polygon against its true one (reference Model.py:208-260):
        polygons are assembled and rasterised on the device; predictions, durations and the IoU counts come
        back in one copy."""
        back = self._decode_run(images, true_vertices)[5].cpu().numpy()
        B, T = back.shape[0], self.max_timesteps
        ious = back[:, -2].astype(np.float64) / back[:, -1].astype(np.float64)
        self.last_ious = ious
        self.last_predicted_vertices = back[:, :-3].reshape(B, T, 2).astype(np.float64)
        self.last_prediction_durations = back[:, -3].astype(np.uint32)
        tag = summary_prefix + '/iou' if summary_prefix else 'iou'
        self._add_summaries(summary_prefix or 'valid', extra={tag: float(np.mean(ious))})
        return float(np.mean(ious))
