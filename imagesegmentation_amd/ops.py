"""Thin tensor-level wrappers over the ConvLSTM entry points of libisg (include/isg.h).

Every function enqueues HIP kernels on torch's current stream and returns torch tensors; none has a
CPU path.  Activations are NHWC, float32 or bfloat16; the cell state and parameter gradients are
float32.
"""
import os

import torch

from . import _lib

F_HIDDEN = 16  # the kernels are specialised for the reference's 16 hidden channels


# ISG_BLOCKING_UPLOADS=1 restores plain pageable copies (a debugging aid: it throttles the host to the GPU's pace)
_BLOCKING_UPLOADS = bool(os.environ.get("ISG_BLOCKING_UPLOADS"))


def upload(array, device, dtype=None):
    """Host array -> device tensor without stalling the host: small arrays (sequence lengths, targets, vertex
    points) go through pinned memory as an asynchronous copy, so the launches queued behind them keep flowing
    while the GPU is still busy with earlier work.  A pageable copy would block until the stream drains."""
    import numpy as np
    arr = np.asarray(array)
    # a conversion the host can do itself happens before the copy (no converting launch on the device)
    host_types = {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32, torch.float64: np.float64,
                  torch.uint8: np.uint8}
    if dtype in host_types and arr.dtype != host_types[dtype]:
        arr = arr.astype(host_types[dtype])
    arr = np.ascontiguousarray(arr)
    device = torch.device(device)
    try:
        t = torch.from_numpy(arr)
    except TypeError:  # dtypes torch has no name for (uint16, uint32, ...)
        t = torch.as_tensor(arr.astype(np.int64 if arr.dtype.kind in "ui" else np.float64))
    if device.type == "cuda" and 0 < t.numel() * t.element_size() <= (1 << 20) and not _BLOCKING_UPLOADS:
        t = t.pin_memory().to(device, non_blocking=True)
    else:
        t = t.to(device)
    return t if dtype is None else t.to(dtype)


_weights_epoch = 0


def weights_epoch():
    """Counts optimiser steps taken by ``adam_tf_``: the kernel updates parameters behind torch's back (no
    version counter moves), so anything cached from parameter values keys on this as well."""
    return _weights_epoch


def note_weights_changed():
    """For code that writes parameters through views torch does not track (checkpoint restore into the flat
    buffer): invalidates everything cached from parameter values."""
    global _weights_epoch
    _weights_epoch += 1


def adam_tf_(param, m, v, grad, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_norm=None, clip_norm=None):
    """One TensorFlow-form Adam step over flat float32 buffers in one launch (isg_adam_tf); with ``clip_norm``
    the gradient is scaled by clip_norm / max(grad_norm, clip_norm) on the fly (grad_norm: device scalar)."""
    global _weights_epoch
    for t in (param, m, v, grad):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == param.numel()
    _need_cuda(param, "param")
    rc = _lib.lib().isg_adam_tf(_lib.ptr(param), _lib.ptr(m), _lib.ptr(v), _lib.ptr(grad), param.numel(), float(lr_t),
                                float(beta1), float(beta2), float(eps), _lib.ptr(grad_norm),
                                float(clip_norm) if clip_norm is not None else 0.0, _lib.current_stream())
    _lib.check(rc, "isg_adam_tf")
    _weights_epoch += 1


# ---- derived weights -----------------------------------------------------------------------------------
# Everything the kernels read that is a re-arrangement (and narrowing) of parameter values: the packed operand images
# of the convolutions, tap tables, compute-type copies of dense weights.  A trainer whose parameters live in one flat
# buffer (polyrnn Model) can own a DerivedWeights registry: the packing functions below record what they produced
# while it is active, ``compile`` turns the recordings into index maps into the flat buffer (by packing a buffer of
# index values with the very same kernels), and from then on ONE launch (isg_refresh_derived) brings all of it up to
# date after an optimiser step, where each image used to be one or two launches of its own per step.
_derived = None


def active_derived():
    return _derived


class DerivedWeights:
    def __init__(self, flat_param):
        assert flat_param.dtype == torch.float32 and flat_param.is_contiguous()
        self.flat = flat_param
        self._recorded = {}   # key -> (object holding the tensors, attribute names, recompute(), parameters read)
        self._compiled = {}   # key -> the same object once its tensors are views of the shared buffers
        self._dirty = False   # recordings the maps do not cover yet
        self._stamp = None
        self._plan = None
        self._depth = 0
        self.generation = 0   # counts compiles that moved the tensors

    def __enter__(self):
        global _derived
        self._depth += 1
        self._outer = _derived if self._depth == 1 else self._outer
        _derived = self
        return self

    def __exit__(self, *exc):
        global _derived
        self._depth -= 1
        if self._depth == 0:
            _derived = self._outer
        return False

    def owns(self, *tensors):
        """Whether these are float32 views of the flat buffer (what the maps can describe)."""
        lo = self.flat.data_ptr()
        hi = lo + self.flat.numel() * 4
        return all(t.dtype == torch.float32 and t.is_contiguous() and lo <= t.data_ptr() and
                   t.data_ptr() + t.numel() * 4 <= hi for t in tensors)

    def _now(self):
        return (weights_epoch(), tuple(p._version for e in self._compiled.values() for p in e[3]))

    def get(self, key):
        """The recorded object under ``key`` with its tensors up to date, or None (not compiled yet: pack as usual).
        Under a stream capture the tensors are handed out as they are - the captured graph then holds no packing
        launches, and whoever replays it calls ``ensure_fresh`` first and drops the graph when ``generation`` has
        moved (a later ``compile`` moves every tensor to new shared buffers)."""
        e = self._compiled.get(key)
        if e is None:
            return None
        if not torch.cuda.is_current_stream_capturing():
            self.ensure_fresh()
        return e[0]

    def record(self, key, obj, names, recompute, params):
        if key not in self._compiled and not torch.cuda.is_current_stream_capturing():
            self._recorded[key] = (obj, list(names), recompute, list(params))
            self._dirty = True

    def ensure_fresh(self):
        """Refreshes every compiled tensor if a parameter changed since the last refresh (one launch).  Code that
        replays a captured graph calls this before the replay: the graph holds no packing launches."""
        if self._plan is not None and self._stamp != self._now():
            self.refresh()

    def compile(self):
        """Turns what was recorded since the last call into maps (no-op when there is nothing new)."""
        if not self._dirty:
            return
        self._dirty = False
        entries = dict(self._compiled)
        entries.update(self._recorded)
        self._recorded = {}
        flat = self.flat
        n = flat.numel()
        keep = flat.clone()
        index = torch.arange(n, device=flat.device, dtype=torch.int64)
        slots = [(key, name) for key, e in entries.items() for name in e[1] if getattr(e[0], name) is not None]
        acc = {slot: None for slot in slots}
        zero = {}
        try:
            for shift in (0, 8, 16, 24):
                if shift and (n - 1) >> shift == 0:
                    break
                # the byte plus one, a value every compute type holds exactly; zero stays what padding looks like
                flat.copy_((((index >> shift) & 255) + 1).to(torch.float32))
                for e in entries.values():
                    e[2]()
                for key, name in slots:
                    v = getattr(entries[key][0], name).detach().reshape(-1).to(torch.float32).round().to(torch.int64)
                    z = v == 0
                    if (key, name) in zero and not torch.equal(z, zero[(key, name)]):
                        raise RuntimeError("derived weights: %r is not a pure re-arrangement of parameters" % (key,))
                    zero[(key, name)] = z
                    part = (v - 1).clamp_(min=0) << shift
                    acc[(key, name)] = part if acc[(key, name)] is None else acc[(key, name)] + part
        finally:
            flat.copy_(keep)
        maps16, maps32, ranges, views = [], [], [], []
        off = {torch.bfloat16: 0, torch.float32: 0}
        for key, name in slots:
            t = getattr(entries[key][0], name)
            m = torch.where(zero[(key, name)], torch.full_like(acc[(key, name)], -1), acc[(key, name)])
            if int(m.max()) >= n:
                raise RuntimeError("derived weights: %r reads outside the parameter buffer" % (key,))
            k = m.numel()
            run = k >= 4096 and t.dtype == torch.bfloat16 and len(ranges) < 4 and \
                bool((m == m[0] + torch.arange(k, device=m.device)).all()) and int(m[0]) >= 0
            if run:
                ranges.append((int(m[0]), k, t))
                continue
            if t.dtype not in off:
                raise RuntimeError("derived weights: %r holds %s" % (key, t.dtype))
            pad = (-k) % 8  # every tensor starts on a 16-byte boundary of its shared buffer
            m = torch.cat([m, torch.full([pad], -1, dtype=m.dtype, device=m.device)]) if pad else m
            (maps16 if t.dtype == torch.bfloat16 else maps32).append(m)
            views.append((key, name, t.dtype, off[t.dtype], k, list(t.shape)))
            off[t.dtype] += k + pad
        dev = flat.device
        cat = lambda ms: torch.cat(ms).to(torch.int32) if ms else torch.zeros([0], dtype=torch.int32, device=dev)
        plan = dict(map16=cat(maps16), map32=cat(maps32), ranges=ranges,
                    out16=torch.zeros([off[torch.bfloat16]], dtype=torch.bfloat16, device=dev),
                    out32=torch.zeros([off[torch.float32]], dtype=torch.float32, device=dev))
        for key, name, dt, o, k, shape in views:
            buf = plan["out16"] if dt == torch.bfloat16 else plan["out32"]
            setattr(entries[key][0], name, buf[o:o + k].view(shape))
        self._compiled, self._plan = entries, plan
        self.generation += 1
        self.refresh()

    def refresh(self):
        import ctypes
        pl = self._plan
        nr = len(pl["ranges"])
        src = (ctypes.c_longlong * 4)(*[r[0] for r in pl["ranges"]])
        cnt = (ctypes.c_longlong * 4)(*[r[1] for r in pl["ranges"]])
        out = (ctypes.c_void_p * 4)(*[r[2].data_ptr() for r in pl["ranges"]])
        rc = _lib.lib().isg_refresh_derived(_lib.ptr(self.flat), self.flat.numel(), _lib.ptr(pl["map16"]),
                                            pl["map16"].numel(), _lib.ptr(pl["out16"]), _lib.ptr(pl["map32"]),
                                            pl["map32"].numel(), _lib.ptr(pl["out32"]), nr, src, cnt, out,
                                            _lib.current_stream())
        _lib.check(rc, "isg_refresh_derived")
        self._stamp = self._now()


# ---- gradient sinks ------------------------------------------------------------------------------------
# A trainer that owns one flat gradient buffer (polyrnn Model, optim.FlatAdam users) can name, per parameter, the
# view of that buffer the gradient belongs in.  The hand-written backward passes then write there themselves and
# hand autograd ``None`` for the parameter: no gradient tensor of its own, no AccumulateGrad add per parameter
# (14 small launches per step of the benchmark net).  A sink receives ASSIGNMENTS: the trainer guarantees that the
# parameter is used once per backward pass.  Outside a ``grad_sinks`` block nothing changes.
_grad_sinks = None


class grad_sinks:
    """``with grad_sinks({param: destination view, ...}, on_delivered=callback):`` - ``on_delivered(param_key)``
    is called after a backward pass has written a parameter's gradient into its view."""

    def __init__(self, mapping, on_delivered=None):
        self.map = {p.data_ptr(): (dst, p) for p, dst in mapping.items()}
        self.on_delivered = on_delivered
        self.delivered = set()

    def __enter__(self):
        global _grad_sinks
        self._outer, _grad_sinks = _grad_sinks, self
        return self

    def __exit__(self, *exc):
        global _grad_sinks
        _grad_sinks = self._outer


def grad_destination(param):
    """The view a parameter's gradient is to be written into (float32, contiguous), or None."""
    if _grad_sinks is None or param is None:
        return None
    hit = _grad_sinks.map.get(param.data_ptr())
    return hit[0] if hit is not None else None


def grad_delivered(param):
    sinks = _grad_sinks
    if sinks is None:
        return
    hit = sinks.map.get(param.data_ptr())
    if hit is not None:
        sinks.delivered.add(param.data_ptr())
        if sinks.on_delivered is not None:
            sinks.on_delivered(hit[1])


def dtype_code(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    raise TypeError("libisg stores activations as float32 or bfloat16, not %s" % dtype)


def channel_multiple(dtype):
    return 16 if dtype == torch.bfloat16 else 4


def padded_channels(c, dtype):
    m = channel_multiple(dtype)
    return (c + m - 1) // m * m


# ---- optional in-step timing -------------------------------------------------------------------------
# ``with timeline() as tl:`` makes every wrapped op record a pair of events on the current stream around its
# launch(es); ``tl.summary()`` gives mean milliseconds per "op[layerN]" label.  Used by the benchmark to read
# kernel durations inside a real training step; costs nothing when off.  The launches at the two ends of a
# sequence are other template instantiations that move fewer bytes (out of the zero state: ",zs"; no incoming
# state gradient: ",edge"): they are filed under labels of their own, "op[layerN,zs]".
_timeline = None
_layer = 0
_variant = ""


def set_layer(l, variant=""):
    global _layer, _variant
    _layer, _variant = l, variant


class timeline:
    def __enter__(self):
        global _timeline
        self.records = []
        _timeline = self
        return self

    def __exit__(self, *exc):
        global _timeline
        _timeline = None

    def summary(self):
        torch.cuda.synchronize()
        acc = {}
        for label, a, b in self.records:
            acc.setdefault(label, []).append(a.elapsed_time(b))
        return {k: sum(v) / len(v) for k, v in acc.items()}


def _timed(name):
    def deco(fn):
        def wrapper(*args, **kwargs):
            tl = _timeline
            if tl is None:
                return fn(*args, **kwargs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(*args, **kwargs)
            b.record()
            tl.records.append(("%s[layer%d%s]" % (name, _layer, _variant), a, b))
            return out
        wrapper.__doc__, wrapper.__name__ = fn.__doc__, fn.__name__
        return wrapper
    return deco


def _timed_as(label_of):
    """As ``_timed`` with the label made from the call's arguments (the encoder's launches are told apart by their
    shapes, not by a recurrence layer)."""
    def deco(fn):
        def wrapper(*args, **kwargs):
            tl = _timeline
            if tl is None:
                return fn(*args, **kwargs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(*args, **kwargs)
            b.record()
            tl.records.append((label_of(*args, **kwargs), a, b))
            return out
        wrapper.__doc__, wrapper.__name__ = fn.__doc__, fn.__name__
        return wrapper
    return deco


def _need_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: libisg has no CPU path" % name)


class PackedWeights:
    """MFMA operand images of one cell's weights (forward + data-gradient), see
    isg_convlstm_pack_weights."""

    def __init__(self, w_x, w_h, bias, cs, dtype, hist=0):
        _need_cuda(w_x, "w_x")
        self.hist = int(hist)            # trailing one-hot input channels applied as a tap scatter
        self.c = int(w_x.shape[2]) - self.hist  # channels that go through the GEMM
        self.cs = int(cs)
        self.f = int(w_h.shape[2])
        self.dtype = dtype
        lib = _lib.lib()
        n_fwd = lib.isg_convlstm_packed_elems(self.cs, self.f, 0)
        n_dg = lib.isg_convlstm_packed_elems(self.cs, self.f, 1)
        if n_fwd < 0 or n_dg < 0:
            raise _lib.IsgError("unsupported ConvLSTM shape: C=%d (stride %d), F=%d" % (self.c, self.cs, self.f))
        self.fwd = torch.empty([n_fwd], dtype=dtype, device=w_x.device)
        self.dgrad = torch.empty([n_dg], dtype=dtype, device=w_x.device)
        self.bias = bias.detach().to(torch.float32).contiguous()
        self.hist_table = torch.empty([9, self.hist, self.f, 4], dtype=torch.float32, device=w_x.device) \
            if self.hist else None
        self._w_x, self._w_h = w_x, w_h
        self.repack()

    def repack(self):
        """(Re)builds the images from the current parameter values."""
        w_x32 = self._w_x.detach().to(torch.float32).contiguous()
        w_h32 = self._w_h.detach().to(torch.float32).contiguous()
        rc = _lib.lib().isg_convlstm_pack_weights(_lib.ptr(w_x32), _lib.ptr(w_h32), self.c, self.cs, self.hist, self.f,
                                                  dtype_code(self.dtype), _lib.ptr(self.fwd), _lib.ptr(self.dgrad),
                                                  _lib.ptr(self.hist_table), _lib.current_stream())
        _lib.check(rc, "isg_convlstm_pack_weights")


def _points_arg(hist_points, hist):
    """(pointer, ints between samples) of an int32 [B, hist, 2] view whose last two axes are dense."""
    if hist == 0:
        return None, 0
    if hist_points is None:
        raise ValueError("these packed weights expect %d history point(s) per sample" % hist)
    assert hist_points.dtype == torch.int32 and hist_points.shape[1:] == (hist, 2)
    assert hist_points.stride(2) == 1 and hist_points.stride(1) == 2
    return hist_points.data_ptr(), hist_points.stride(0)


def pooled_fwd_supported(packed, B, H, W, pool):
    """Whether cell_fwd can deliver its emitted output already averaged over pool x pool windows
    (ISG_NO_FUSED_POOL=1 answers no: the separate pooling pass, for A/B runs)."""
    if os.environ.get("ISG_NO_FUSED_POOL"):
        return False
    return bool(_lib.lib().isg_convlstm_pooled_fwd_supported(packed.cs, packed.f, dtype_code(packed.dtype), int(B),
                                                             int(H), int(W), int(pool), packed.hist))


SAVE_GATES, SAVE_FACTORS = 0, 1  # what a forward step keeps for the backward pass (ISG_SAVE_* of include/isg.h)
STATE_BF16 = 0x10                # or-ed into the save layout: c_prev / c_out hold bf16 (ISG_STATE_BF16)
CPREV_BF16 = 0x200               # or-ed into bwd_fused's mode: c_prev holds bf16 (ISG_CPREV_BF16)


def state_bf16_supported(packed, B, H, W):
    """Whether the forward step can keep the cell state in bf16 for this cell shape (the bf16 ping-pong kernels;
    ISG_STATE_F32=1 answers no: float cell states everywhere, for A/B runs)."""
    if os.environ.get("ISG_STATE_F32"):
        return False
    return bool(_lib.lib().isg_convlstm_state_bf16_supported(packed.cs, packed.f, dtype_code(packed.dtype), int(B),
                                                             int(H), int(W)))


def saved_shape(B, H, W, F, layout):
    """Shape of the buffer a forward step saves for backward: the gate activations [B,H,W,F,4] (read by
    ``pointwise_bwd`` together with both cell states) or the backward factors [B,H,W,F/2,4,2] (bf16, same size;
    ``bwd_fused`` reads them with c_prev and the step's h; layout in include/isg.h)."""
    return [B, H, W, F, 4] if layout == SAVE_GATES else [B, H, W, F // 2, 4, 2]


@_timed("cell_fwd")
def cell_fwd(x, h_prev, c_prev, packed, durations=None, t=0, out=None, out_batch_stride=0, save_gates=True,
             h_out=None, c_out=None, gates=None, hist_points=None, pool_out=None, pool_batch_stride=0, pool=1,
             save_layout=SAVE_GATES, h_pixel_stride=None, out_pixel_stride=None):
    """One forward step. x [B,H,W,Cs], h_prev [B,H,W,F] (activation dtype), c_prev float32 - or bfloat16, which
    makes the step read and write the cell state as bfloat16 (where ``state_bf16_supported``; ``c_out`` then
    decides when c_prev is None).
    h_prev = c_prev = None: the zero state.
    hist_points: int32 [B,hist,2] when the packed weights carry history channels.
    ``gates`` (or ``save_gates``): the buffer kept for backward, in ``save_layout`` (see ``saved_shape``).
    ``pool_out`` ([B, H/pool, W/pool, F] view with batch stride ``pool_batch_stride``; only where
    ``pooled_fwd_supported``) receives the window means of the emitted output instead of ``out``.
    ``h_pixel_stride`` / ``out_pixel_stride`` (elements): ``h_out`` / ``out`` are 16-channel slices of wider arrays
    (their ``data_ptr()`` is the first pixel's first channel); the slices must not overlap what the step reads.
    Returns (h, c, gates)."""
    _need_cuda(x, "x")
    B, H, W, Cs = x.shape
    F = packed.f
    assert Cs == packed.cs and x.dtype == packed.dtype
    assert (h_prev is None) == (c_prev is None)  # both None: the zero state, nothing is read for it
    cdt = c_prev.dtype if c_prev is not None else (c_out.dtype if c_out is not None else torch.float32)
    assert h_prev is None or h_prev.dtype == packed.dtype
    assert cdt == torch.float32 or (cdt == torch.bfloat16 and x.dtype == torch.bfloat16)
    h = torch.empty([B, H, W, F], dtype=x.dtype, device=x.device) if h_out is None else h_out
    c = torch.empty([B, H, W, F], dtype=cdt, device=x.device) if c_out is None else c_out
    assert c.dtype == cdt, "the cell state before and after a step is stored in one type"
    if cdt == torch.bfloat16:
        save_layout = int(save_layout) | STATE_BF16
    if gates is None and save_gates:
        gates = torch.empty(saved_shape(B, H, W, F, save_layout & ~STATE_BF16), dtype=x.dtype, device=x.device)
    assert gates is None or (gates.is_contiguous() and gates.numel() == B * H * W * F * 4)
    pts, pts_stride = _points_arg(hist_points, packed.hist)
    if pool_out is not None:
        assert out is None and pool_out.dtype == x.dtype
        rc = _lib.lib().isg_convlstm_cell_fwd_pooled(
            _lib.ptr(x), _lib.ptr(h_prev), _lib.ptr(c_prev), _lib.ptr(packed.fwd), _lib.ptr(packed.bias), B, H, W, Cs,
            F, dtype_code(x.dtype), _lib.ptr(durations), int(t), pts, packed.hist, pts_stride,
            _lib.ptr(packed.hist_table), _lib.ptr(h), _lib.ptr(c), _lib.ptr(gates), int(save_layout), int(pool),
            pool_out.data_ptr(),
            int(pool_batch_stride), _lib.current_stream())
        _lib.check(rc, "isg_convlstm_cell_fwd_pooled")
        return h, c, gates
    if h_pixel_stride is not None or out_pixel_stride is not None:
        assert h_out is not None or h_pixel_stride is None
        rc = _lib.lib().isg_convlstm_cell_fwd_strided(
            _lib.ptr(x), _lib.ptr(h_prev), _lib.ptr(c_prev), _lib.ptr(packed.fwd), _lib.ptr(packed.bias), B, H, W, Cs,
            F, dtype_code(x.dtype), _lib.ptr(durations), int(t), pts, packed.hist, pts_stride,
            _lib.ptr(packed.hist_table), h.data_ptr(), int(h_pixel_stride or F), _lib.ptr(c),
            out.data_ptr() if out is not None else None, int(out_batch_stride), int(out_pixel_stride or F),
            _lib.ptr(gates), int(save_layout), _lib.current_stream())
        _lib.check(rc, "isg_convlstm_cell_fwd_strided")
        return h, c, gates
    rc = _lib.lib().isg_convlstm_cell_fwd(_lib.ptr(x), _lib.ptr(h_prev), _lib.ptr(c_prev), _lib.ptr(packed.fwd),
                                          _lib.ptr(packed.bias), B, H, W, Cs, F, dtype_code(x.dtype),
                                          _lib.ptr(durations), int(t), pts, packed.hist, pts_stride,
                                          _lib.ptr(packed.hist_table), _lib.ptr(h), _lib.ptr(c),
                                          out.data_ptr() if out is not None else None, int(out_batch_stride),
                                          _lib.ptr(gates), int(save_layout), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_cell_fwd")
    return h, c, gates


@_timed("pointwise_bwd")
def pointwise_bwd(gates, c_prev, c_cur, dh_above, dh_above_batch_stride, dh_state, dc_state, durations=None, t=0,
                  dy=None, dc_prev=None, pool=1):
    """Gate backward. Returns (dy [B,H,W,F,4], dc_prev [B,H,W,F]).  ``pool`` = k > 1: dh_above is
    the gradient of the k x k average-pooled output, [H/k, W/k, F] per sample.  The carried cell-state
    gradient (dc_state in, dc_prev out) is float32 or bfloat16, both the same; float32 when neither is given."""
    B, H, W, F, _ = gates.shape
    dy = torch.empty_like(gates) if dy is None else dy
    dc_dtype = dc_prev.dtype if dc_prev is not None else (dc_state.dtype if dc_state is not None else torch.float32)
    dc_prev = torch.empty([B, H, W, F], dtype=dc_dtype, device=gates.device) if dc_prev is None else dc_prev
    assert dc_prev.dtype == dc_dtype and (dc_state is None or dc_state.dtype == dc_dtype)
    rc = _lib.lib().isg_convlstm_pointwise_bwd(
        _lib.ptr(gates), _lib.ptr(c_prev), _lib.ptr(c_cur), dh_above.data_ptr() if dh_above is not None else None,
        int(dh_above_batch_stride), int(pool), _lib.ptr(dh_state), _lib.ptr(dc_state), dtype_code(dc_dtype),
        B, H, W, F, dtype_code(gates.dtype),
        _lib.ptr(durations), int(t), _lib.ptr(dy), _lib.ptr(dc_prev), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_pointwise_bwd")
    return dy, dc_prev


@_timed("dgrad")
def dgrad(dy, packed, dh_prev, dx=None, dx_channels=0, dx_accumulate=False, durations=None, t=0):
    """Data gradient. dh_prev [B,H,W,F] is written (left untouched for masked samples); dx optional."""
    B, H, W, F, _ = dy.shape
    rc = _lib.lib().isg_convlstm_dgrad(
        _lib.ptr(dy), _lib.ptr(packed.dgrad), B, H, W, packed.cs, F, dtype_code(dy.dtype), _lib.ptr(durations),
        int(t), _lib.ptr(dx), int(dx_channels), int(dx.shape[-1]) if dx is not None else 0, int(bool(dx_accumulate)),
        _lib.ptr(dh_prev), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_dgrad")
    return dx, dh_prev


_workspaces = {}

WGRAD_ASSIGN, WGRAD_ADD, WGRAD_DEFER_FIRST, WGRAD_DEFER = 0, 1, 2, 3


def wgrad_workspace(cs, f, device, cached=True):
    """Scratch for the per-workgroup partial sums; ``cached=False`` gives a private one (needed
    when sums are deferred over several launches)."""
    key = (cs, f, str(device))
    ws = _workspaces.get(key) if cached else None
    if ws is None:
        nbytes = _lib.lib().isg_convlstm_wgrad_workspace_bytes(cs, f)
        if nbytes < 0:
            raise _lib.IsgError("weight gradient supports at most %d input channels (got %d)" % (64 - f, cs))
        ws = torch.empty([nbytes // 4], dtype=torch.float32, device=device)
        if cached:
            _workspaces[key] = ws
    return ws


@_timed("wgrad")
def wgrad(x, h_prev, dy, c_real, dw_x, dw_h, dbias, accumulate, durations=None, t=0, hist_points=None, hist=0,
          mode=None, workspace=None):
    """Parameter gradients of one step into float32 dw_x [3,3,C+hist,4F], dw_h [3,3,F,4F], dbias [4F];
    c_real = channels of x that went through the GEMM.  ``mode`` (one of the WGRAD_* codes)
    overrides ``accumulate``; the deferred modes need a private ``workspace`` and a closing
    ``wgrad_finish``."""
    B, H, W, Cs = x.shape
    F = h_prev.shape[-1]
    if mode is None:
        mode = WGRAD_ADD if accumulate else WGRAD_ASSIGN
    if workspace is None:
        if mode in (WGRAD_DEFER_FIRST, WGRAD_DEFER):
            raise ValueError("deferred weight-gradient sums need their own workspace")
        workspace = wgrad_workspace(Cs, F, x.device)
    pts, pts_stride = _points_arg(hist_points, hist)
    rc = _lib.lib().isg_convlstm_wgrad(_lib.ptr(x), _lib.ptr(h_prev), _lib.ptr(dy), B, H, W, int(c_real), Cs, F,
                                       dtype_code(x.dtype), _lib.ptr(durations), int(t), pts, int(hist), pts_stride,
                                       _lib.ptr(workspace), int(mode), _lib.ptr(dw_x), _lib.ptr(dw_h),
                                       _lib.ptr(dbias), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_wgrad")


def wgrad_finish(workspace, shape, c_real, hist, dw_x, dw_h, dbias, accumulate=False):
    """Fold the partial sums left by deferred ``wgrad`` calls over inputs of ``shape`` [B,H,W,Cs]."""
    B, H, W, Cs = shape
    rc = _lib.lib().isg_convlstm_wgrad_finish(_lib.ptr(workspace), B, H, W, int(c_real), Cs, int(hist),
                                              dw_h.shape[2], int(bool(accumulate)), _lib.ptr(dw_x), _lib.ptr(dw_h),
                                              _lib.ptr(dbias), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_wgrad_finish")


def bwd_fused_supported(packed, B, H, W, dx_channels=None):
    """Whether the one-launch backward step exists for this cell shape (ISG_NO_FUSED_BWD=1 answers no: the
    three-kernel backward, for A/B runs).  ``dx_channels``: how many channels of the input gradient the caller
    will ask the step for - it stores them in runs of 8, so a count that is no multiple of 8 answers no."""
    if os.environ.get("ISG_NO_FUSED_BWD"):
        return False
    if dx_channels is not None and (dx_channels < 8 or dx_channels % 8 != 0):
        return False
    return bool(_lib.lib().isg_convlstm_bwd_fused_supported(packed.cs, packed.f, dtype_code(packed.dtype), int(B), int(H),
                                                            int(W), packed.hist))


_hist_scratch = {}
HIST_DEFER = 0x100


def bwd_fused_hist_begin(B, steps, packed, device):
    """Zeroed scratch for the history taps' dY of ``steps`` fused backward steps (one slot per step), folded into
    ``dw_x`` once by ``bwd_fused_hist_finish`` instead of after every step."""
    n = _lib.lib().isg_convlstm_bwd_fused_hist_bytes(B * steps, packed.hist)
    scratch = torch.empty([n // 2], dtype=torch.bfloat16, device=device)
    rc = _lib.lib().isg_convlstm_bwd_fused_hist_begin(_lib.ptr(scratch), B * steps, packed.hist, _lib.current_stream())
    _lib.check(rc, "isg_convlstm_bwd_fused_hist_begin")
    return scratch


def bwd_fused_hist_finish(scratch, B, steps, packed, dw_x, accumulate=False):
    rc = _lib.lib().isg_convlstm_bwd_fused_hist_finish(_lib.ptr(scratch), B * steps, packed.hist, packed.c,
                                                       int(bool(accumulate)), _lib.ptr(dw_x), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_bwd_fused_hist_finish")



@_timed("bwd_fused")
def bwd_fused(factors, c_prev, h_cur, dh_above, dh_above_batch_stride, dh_state, dc_state, x, h_prev, packed, dc_prev,
              dh_prev, workspace, mode, dw_x, dx=None, dx_channels=0, dx_accumulate=False, durations=None, t=0,
              hist_points=None, pool=1, hist_slot=None, dh_above_pixel_stride=None):
    """One backward step of the cell in one launch (isg_convlstm_bwd_fused): gate backward from the ``factors``
    the forward step saved (``cell_fwd(save_layout=SAVE_FACTORS)``), the cell state before the step (``c_prev``)
    and the step's new h (``h_cur``); data gradient (dx optional, dh_prev) and
    weight gradient (deferred partial sums in ``workspace``; the history taps' rows of ``dw_x`` directly).
    ``hist_slot`` = (scratch, step): the history taps' dY goes to slot ``step`` of a sequence's scratch
    (``bwd_fused_hist_begin``) and ``dw_x`` is left to ``bwd_fused_hist_finish``.
    dc_state / dh_state are read, dc_prev / dh_prev written: different buffers.  c_prev = h_prev = None: the step
    out of the zero state; dc_prev = dh_prev = None: the gradient of the state before the step is not wanted."""
    B, H, W, F = h_cur.shape
    gates = factors
    assert factors.is_contiguous() and factors.numel() == B * H * W * F * 4 and factors.dtype == torch.bfloat16
    h_stride = pixel_stride(h_cur)
    dh_stride = pixel_stride(dh_state) if dh_state is not None else F
    da_stride = F if dh_above_pixel_stride is None else int(dh_above_pixel_stride)
    if h_stride != F or dh_stride != F or da_stride != F:
        # h, its carried gradient and / or the output's gradient as 16-channel slices of wider pixels
        # (isg_convlstm_bwd_fused_strided)
        assert h_stride is not None and dh_stride is not None and packed.hist == 0 and pool == 1
        if c_prev is not None and c_prev.dtype == torch.bfloat16:
            mode = int(mode) | CPREV_BF16
        rc = _lib.lib().isg_convlstm_bwd_fused_strided(
            _lib.ptr(gates), _lib.ptr(c_prev), h_cur.data_ptr(), int(h_stride),
            dh_above.data_ptr() if dh_above is not None else None, int(dh_above_batch_stride), da_stride, int(pool),
            dh_state.data_ptr() if dh_state is not None else None, int(dh_stride), _lib.ptr(dc_state), _lib.ptr(x),
            _lib.ptr(h_prev), _lib.ptr(packed.dgrad), B, H, W, packed.c, packed.cs, F, dtype_code(gates.dtype),
            _lib.ptr(durations), int(t), _lib.ptr(dc_prev), _lib.ptr(dh_prev), _lib.ptr(dx), int(dx_channels),
            int(dx.shape[-1]) if dx is not None else 0, int(bool(dx_accumulate)), _lib.ptr(workspace), int(mode),
            _lib.current_stream())
        _lib.check(rc, "isg_convlstm_bwd_fused_strided")
        return
    assert (c_prev is None or c_prev.dtype in (torch.float32, torch.bfloat16)) and h_cur.dtype == torch.bfloat16
    if c_prev is not None and c_prev.dtype == torch.bfloat16:
        mode = int(mode) | CPREV_BF16
    pts, pts_stride = _points_arg(hist_points, packed.hist)
    scratch = None
    scratch_ptr = None
    if packed.hist and hist_slot is not None:
        scratch_ptr = hist_slot[0].data_ptr() + int(hist_slot[1]) * B * packed.hist * 9 * 64 * 2
        mode = int(mode) | HIST_DEFER
    elif packed.hist:
        key = (B, packed.hist, str(gates.device))
        scratch = _hist_scratch.get(key)
        if scratch is None:
            n = _lib.lib().isg_convlstm_bwd_fused_hist_bytes(B, packed.hist)
            scratch = _hist_scratch[key] = torch.empty([n // 2], dtype=torch.bfloat16, device=gates.device)
    rc = _lib.lib().isg_convlstm_bwd_fused(
        _lib.ptr(gates), _lib.ptr(c_prev), _lib.ptr(h_cur), dh_above.data_ptr() if dh_above is not None else None,
        int(dh_above_batch_stride), int(pool), _lib.ptr(dh_state), _lib.ptr(dc_state), _lib.ptr(x), _lib.ptr(h_prev),
        _lib.ptr(packed.dgrad), B, H, W, packed.c, packed.cs, F, dtype_code(gates.dtype), _lib.ptr(durations), int(t),
        pts, packed.hist, pts_stride, scratch_ptr if scratch_ptr is not None else _lib.ptr(scratch), _lib.ptr(dc_prev),
        _lib.ptr(dh_prev), _lib.ptr(dx),
        int(dx_channels), int(dx.shape[-1]) if dx is not None else 0, int(bool(dx_accumulate)), _lib.ptr(workspace),
        int(mode), _lib.ptr(dw_x), _lib.current_stream())
    _lib.check(rc, "isg_convlstm_bwd_fused")


def avgpool(x, k):
    """Mean over k x k windows of a contiguous NHWC tensor [..., H, W, F] -> [..., H/k, W/k, F]."""
    _need_cuda(x, "x")
    assert x.is_contiguous()
    H, W, F = x.shape[-3:]
    n = x.numel() // (H * W * F)
    out = torch.empty(tuple(x.shape[:-3]) + (H // k, W // k, F), dtype=x.dtype, device=x.device)
    rc = _lib.lib().isg_avgpool_fwd(_lib.ptr(x), n, H, W, F, int(k), dtype_code(x.dtype), _lib.ptr(out),
                                    _lib.current_stream())
    _lib.check(rc, "isg_avgpool_fwd")
    return out


def bias_relu_(y, bias):
    """In place y = max(y + bias, 0) on a contiguous NHWC tensor; bias float32 [C]."""
    _need_cuda(y, "y")
    assert y.is_contiguous() and bias.dtype == torch.float32 and bias.is_contiguous()
    C = y.shape[-1]
    rc = _lib.lib().isg_bias_relu(_lib.ptr(y), _lib.ptr(bias), y.numel() // C, C, dtype_code(y.dtype),
                                  _lib.current_stream())
    _lib.check(rc, "isg_bias_relu")
    return y


_bias_ws = {}


def _pixel_levels(views):
    """The pixel dimensions (all but the last) of same-shaped views, merged where every view walks them with one
    stride: [(extent, [element stride per view])], innermost first; None when a view's channels are not adjacent."""
    if any(v.dim() < 2 or v.stride(-1) != 1 for v in views):
        return None
    levels = []
    for d in reversed(range(views[0].dim() - 1)):
        n = views[0].shape[d]
        if n == 1:
            continue
        if levels and all(v.stride(d) == st * levels[-1][0] for v, st in zip(views, levels[-1][1])):
            levels[-1][0] *= n
        else:
            levels.append([n, [v.stride(d) for v in views]])
    return levels


def copy_slice(dst, src):
    """dst[...] = src[...] for two channel-slice views of NHWC arrays (same shape and dtype): one launch of 16-byte
    moves where the views' pixel dimensions merge into at most three strided levels (a slice of a wider array; a
    batch-major array seen time-major) - a torch copy of such views moves 2-byte elements through a generic strided
    kernel, several times slower.  Any other pair goes through ``dst.copy_``.  Returns dst."""
    if dst.is_cuda and src.is_cuda and dst.dtype == src.dtype and dst.shape == src.shape and dst.numel() > 0:
        levels = _pixel_levels([dst, src])
        es = dst.element_size()
        nbytes = dst.shape[-1] * es
        if levels is not None and len(levels) <= 3 and nbytes % 16 == 0 and dst.data_ptr() % 16 == 0 and \
                src.data_ptr() % 16 == 0:
            levels = levels + [[1, [0, 0]]] * (3 - len(levels))  # innermost first; absent outer levels: one block
            if levels[0][1] == [0, 0]:
                levels[0] = [1, [dst.shape[-1], dst.shape[-1]]]
            ds = [levels[k][1][0] * es for k in (2, 1, 0)]
            ss = [levels[k][1][1] * es for k in (2, 1, 0)]
            if all(v % 16 == 0 for v in ds + ss) and ds[2] >= nbytes and ss[2] >= nbytes:
                import ctypes
                arr = ctypes.c_longlong * 3
                rc = _lib.lib().isg_copy_channel_slice(src.data_ptr(), arr(*ss), dst.data_ptr(), arr(*ds), nbytes,
                                                       levels[2][0], levels[1][0], levels[0][0], _lib.current_stream())
                _lib.check(rc, "isg_copy_channel_slice")
                return dst
    dst.copy_(src)
    return dst


def pad_channels(x, cout, out=None):
    """x float32 or bfloat16 [.., cin] -> bfloat16 [.., cout] with zero channels cin..: conversion and padding in one
    pass (an observation for a first encoder layer on the square kernels, ``conv32_pack_padded``)."""
    assert x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] <= cout and cout % 8 == 0
    x = x.contiguous()
    y = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=torch.bfloat16, device=x.device) if out is None else out
    assert y.dtype == torch.bfloat16 and y.is_contiguous() and tuple(y.shape) == tuple(x.shape[:-1]) + (cout,)
    rc = _lib.lib().isg_pad_channels(_lib.ptr(x), dtype_code(x.dtype), int(x.shape[-1]), _lib.ptr(y), int(cout),
                                     x.numel() // x.shape[-1], _lib.current_stream())
    _lib.check(rc, "isg_pad_channels")
    return y


def pixel_stride(t):
    """Elements from one pixel of ``t`` [..., C] to the next when its pixels are evenly spaced in memory in index order
    (a contiguous tensor: C; a channel slice of a contiguous tensor with wider pixels: that width), else None."""
    if t.dim() < 2 or t.stride(-1) != 1:
        return None
    step = None
    for size, stride in zip(reversed(t.shape[:-1]), reversed(t.stride()[:-1])):
        if size == 1:
            continue
        if step is None:
            base = step = stride
        elif stride != step:
            return None
        step = step * size
    return t.shape[-1] if step is None else base


@_timed_as(lambda dy, y, dbias_out=None: "relu_bwd_bias[C=%d]" % y.shape[-1])
def relu_bwd_bias(dy, y, dbias_out=None):
    """(dz, dbias): dz = dy masked by y > 0 (written over dy when dy is contiguous and owns its memory),
    dbias float32 [C] = sum of dz over pixels (into ``dbias_out`` when given).  ``y`` may be a channel slice of
    wider pixels (``pixel_stride``)."""
    C = y.shape[-1]
    ystride, dystride = pixel_stride(y), pixel_stride(dy)
    assert ystride is not None and ystride >= C and dystride is not None and dystride >= C and dy.shape == y.shape
    key = (C, str(y.device))
    ws = _bias_ws.get(key)
    if ws is None:
        ws = torch.empty([_lib.lib().isg_relu_bwd_bias_workspace_bytes(C) // 4], dtype=torch.float32, device=y.device)
        _bias_ws[key] = ws
    dz = torch.empty(dy.shape, dtype=dy.dtype, device=dy.device)
    dbias = torch.empty([C], dtype=torch.float32, device=y.device) if dbias_out is None else dbias_out
    assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == C
    if ystride != C or dystride != C:
        rc = _lib.lib().isg_relu_bwd_bias_strided(dy.data_ptr(), dystride, y.data_ptr(), ystride, y.numel() // C, C,
                                                  dtype_code(y.dtype), _lib.ptr(dz), _lib.ptr(dbias), _lib.ptr(ws),
                                                  _lib.current_stream())
        _lib.check(rc, "isg_relu_bwd_bias_strided")
        return dz, dbias
    rc = _lib.lib().isg_relu_bwd_bias(_lib.ptr(dy), _lib.ptr(y), y.numel() // C, C, dtype_code(y.dtype), _lib.ptr(dz),
                                      _lib.ptr(dbias), _lib.ptr(ws), _lib.current_stream())
    _lib.check(rc, "isg_relu_bwd_bias")
    return dz, dbias


# ---- the encoder's first convolution (14 x 14 taps, 3 -> 32 channels, bf16) ----------------------
def conv_in_supported(w_hwio, dtype, device):
    """Whether the hand-written first-layer kernels take this layer: 3 input channels, square 14 x 14 or 5 x 5
    taps, 32 or 16 output channels, bf16."""
    kh, kw, cin, cout = w_hwio.shape
    return (kh == kw and cin == 3 and dtype == torch.bfloat16 and device.type == "cuda" and
            _lib.lib().isg_conv_in_packed_elems(int(kh), int(cout)) > 0)


def conv_in_pack(w_hwio, out=None):
    """HWIO float weights [KS,KS,3,cout] -> the kernel's bf16 A-fragment image [KS * KXP/4, 64, 8] (output-channel
    rows padded to the 32 of an MFMA tile)."""
    reg = _derived if out is None else None
    key = ("conv_in", w_hwio.data_ptr(), tuple(w_hwio.shape))
    if reg is not None:
        e = reg.get(key)
        if e is not None:
            return e.out
    ks, cout = w_hwio.shape[0], w_hwio.shape[3]
    kxp = (ks + 3) // 4 * 4
    w = w_hwio.detach().float().contiguous()
    if out is None:
        out = torch.empty([ks * (kxp // 4), 64, 8], dtype=torch.bfloat16, device=w.device)
    rc = _lib.lib().isg_conv_in_pack(_lib.ptr(w), int(ks), int(cout), _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_conv_in_pack")
    if reg is not None and reg.owns(w_hwio):
        import types
        ns = types.SimpleNamespace(out=out)
        reg.record(key, ns, ["out"], lambda: conv_in_pack(w_hwio, out=ns.out), [w_hwio])
    return out


def conv_in_pad_image(x3):
    """bf16 [B,H,W,3] -> [B,H,W,4] with a zero fourth channel."""
    assert x3.dtype == torch.bfloat16 and x3.is_contiguous() and x3.shape[-1] == 3
    x4 = torch.empty(tuple(x3.shape[:-1]) + (4,), dtype=x3.dtype, device=x3.device)
    rc = _lib.lib().isg_conv_in_pad_image(_lib.ptr(x3), x3.numel() // 3, _lib.ptr(x4), _lib.current_stream())
    _lib.check(rc, "isg_conv_in_pad_image")
    return x4


@_timed_as(lambda x4, wimg, bias, ks, relu=True: "conv_in_fwd[%dx%d]" % (ks, ks))
def conv_in_fwd(x4, wimg, bias, ks, relu=True):
    B, H, W, _ = x4.shape
    cout = bias.shape[0]
    y = torch.empty([B, H, W, cout], dtype=torch.bfloat16, device=x4.device)
    rc = _lib.lib().isg_conv_in_fwd(_lib.ptr(x4), _lib.ptr(wimg), _lib.ptr(bias), B, H, W, int(ks), int(cout),
                                    int(bool(relu)), _lib.ptr(y), _lib.current_stream())
    _lib.check(rc, "isg_conv_in_fwd")
    return y


_conv_in_ws = {}


@_timed_as(lambda x4, dz, ks, out=None: "conv_in_wgrad[%dx%d]" % (ks, ks))
def conv_in_wgrad(x4, dz, ks, out=None):
    """float32 [KS,KS,3,cout] weight gradient from the padded image and the masked output gradient."""
    B, H, W, _ = x4.shape
    cout = dz.shape[-1]
    assert dz.is_contiguous() and dz.dtype == torch.bfloat16 and dz.shape == (B, H, W, cout)
    key = (int(ks), int(cout), str(x4.device))
    ws = _conv_in_ws.get(key)
    if ws is None:
        ws = torch.empty([_lib.lib().isg_conv_in_wgrad_workspace_bytes(int(ks), int(cout)) // 4], dtype=torch.float32,
                         device=x4.device)
        _conv_in_ws[key] = ws
    dw = torch.empty([ks, ks, 3, cout], dtype=torch.float32, device=x4.device) if out is None else out
    assert dw.dtype == torch.float32 and dw.is_contiguous() and tuple(dw.shape) == (ks, ks, 3, cout)
    rc = _lib.lib().isg_conv_in_wgrad(_lib.ptr(x4), _lib.ptr(dz), B, H, W, int(ks), int(cout), _lib.ptr(ws), _lib.ptr(dw),
                                      _lib.current_stream())
    _lib.check(rc, "isg_conv_in_wgrad")
    return dw


# ---- the encoder's 32 -> 32 channel convolutions (3 x 3 / 5 x 5 taps, bf16) ---------------------------
def conv32_supported(w_hwio, dtype, device):
    """Whether the hand-written C -> C encoder kernels take this layer: square 3 x 3 or 5 x 5, C = 32 or 16, bf16."""
    kh, kw, cin, cout = w_hwio.shape
    return (kh == kw and cin == cout and cin in (16, 32) and dtype == torch.bfloat16 and device.type == "cuda" and
            _lib.lib().isg_conv32_packed_elems(int(kh)) > 0)


def conv32_pack(w_hwio, data_gradient=False, out=None):
    """HWIO float weights [KS,KS,C,C] (C = 32 or 16) -> the kernel's bf16 A-fragment image [KS*KS*C/16, 64, 8]
    (output-channel rows padded to the 32 of an MFMA tile); with ``data_gradient`` the image of the transposed
    convolution (flipped taps, channels swapped)."""
    reg = _derived if out is None else None
    key = ("conv32", w_hwio.data_ptr(), tuple(w_hwio.shape), bool(data_gradient))
    if reg is not None:
        e = reg.get(key)
        if e is not None:
            return e.out
    w = w_hwio.detach().float().contiguous()
    ks, c = w.shape[0], w.shape[2]
    if out is None:
        out = torch.empty([ks * ks * (c // 16), 64, 8], dtype=torch.bfloat16, device=w.device)
    rc = _lib.lib().isg_conv32_pack(_lib.ptr(w), int(ks), int(c), int(bool(data_gradient)), _lib.ptr(out),
                                    _lib.current_stream())
    _lib.check(rc, "isg_conv32_pack")
    if reg is not None and reg.owns(w_hwio):
        import types
        ns = types.SimpleNamespace(out=out)
        reg.record(key, ns, ["out"], lambda: conv32_pack(w_hwio, data_gradient, out=ns.out), [w_hwio])
    return out


def conv32_pack_padded(w_hwio, out=None):
    """HWIO float weights [KS,KS,cin,C] with cin < C (C = 32 or 16) -> the forward image of ``conv32`` for an input
    padded to C channels with zeros."""
    reg = _derived if out is None else None
    key = ("conv32_padded", w_hwio.data_ptr(), tuple(w_hwio.shape))
    if reg is not None:
        e = reg.get(key)
        if e is not None:
            return e.out
    w = w_hwio.detach().float().contiguous()
    ks, cin, c = w.shape[0], w.shape[2], w.shape[3]
    if out is None:
        out = torch.empty([ks * ks * (c // 16), 64, 8], dtype=torch.bfloat16, device=w.device)
    rc = _lib.lib().isg_conv32_pack_padded(_lib.ptr(w), int(ks), int(c), int(cin), _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_conv32_pack_padded")
    if reg is not None and reg.owns(w_hwio):
        import types
        ns = types.SimpleNamespace(out=out)
        reg.record(key, ns, ["out"], lambda: conv32_pack_padded(w_hwio, out=ns.out), [w_hwio])
    return out


def conv32_padded_supported(w_hwio, dtype, device):
    """Whether ``conv32`` takes this layer on an input padded with zero channels: square 3 x 3 or 5 x 5, 32 or 16
    output channels, fewer input channels than that, bf16."""
    kh, kw, cin, cout = w_hwio.shape
    return (kh == kw and cin < cout and cout in (16, 32) and dtype == torch.bfloat16 and device.type == "cuda" and
            _lib.lib().isg_conv32_packed_elems(int(kh)) > 0)


@_timed_as(lambda x, wimg, bias, ks, relu, out=None: "conv32[%dx%d C=%d]" % (ks, ks, x.shape[-1]))
def conv32(x, wimg, bias, ks, relu, out=None):
    B, H, W, c = x.shape
    assert c in (16, 32) and x.dtype == torch.bfloat16 and x.is_contiguous()
    y = torch.empty_like(x) if out is None else out
    assert y.shape == x.shape and y.dtype == x.dtype and y.data_ptr() != x.data_ptr()
    if not y.is_contiguous():  # a 32-channel slice of wider pixels (isg_conv32_strided)
        ystride = pixel_stride(y)
        assert c == 32 and ystride is not None and ystride >= c, "out: evenly spaced pixels of 32 channels"
        rc = _lib.lib().isg_conv32_strided(_lib.ptr(x), _lib.ptr(wimg), _lib.ptr(bias), B, H, W, int(ks), int(bool(relu)),
                                           y.data_ptr(), ystride, _lib.current_stream())
        _lib.check(rc, "isg_conv32_strided")
        return y
    fn = _lib.lib().isg_conv32 if c == 32 else _lib.lib().isg_conv16
    rc = fn(_lib.ptr(x), _lib.ptr(wimg), _lib.ptr(bias), B, H, W, int(ks), int(bool(relu)), _lib.ptr(y),
            _lib.current_stream())
    _lib.check(rc, "isg_conv%d" % c)
    return y


_conv32_mask_ws = {}


@_timed_as(lambda dz, wimg_dgrad, y_below, ks, dbias_out=None: "conv32_masked[%dx%d C=%d]" % (ks, ks, dz.shape[-1]))
def conv32_masked(dz, wimg_dgrad, y_below, ks, dbias_out=None):
    """(dz_below, dbias_below): the data gradient of a conv32 / conv16 layer masked by the ReLU of the layer below
    (``y_below`` = that layer's output) and the masked values summed per channel (float32 [C]) - the layer below's
    ReLU backward and bias gradient without a pass of their own (isg_conv32_masked)."""
    B, H, W, c = dz.shape
    assert c in (16, 32) and dz.dtype == torch.bfloat16 and dz.is_contiguous()
    assert y_below.shape == dz.shape and y_below.dtype == dz.dtype and y_below.is_contiguous()
    key = str(dz.device)
    ws = _conv32_mask_ws.get(key)
    if ws is None:
        ws = _conv32_mask_ws[key] = torch.empty([_lib.lib().isg_conv32_masked_workspace_bytes() // 4], dtype=torch.float32,
                                                device=dz.device)
    out = torch.empty_like(dz)
    dbias = torch.empty([c], dtype=torch.float32, device=dz.device) if dbias_out is None else dbias_out
    assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == c
    fn = _lib.lib().isg_conv32_masked if c == 32 else _lib.lib().isg_conv16_masked
    rc = fn(_lib.ptr(dz), _lib.ptr(wimg_dgrad), _lib.ptr(y_below), B, H, W, int(ks), _lib.ptr(out), _lib.ptr(dbias),
            _lib.ptr(ws), _lib.current_stream())
    _lib.check(rc, "isg_conv%d_masked" % c)
    return out, dbias


_conv32_ws = {}


@_timed_as(lambda x, dz, ks, out=None: "conv32_wgrad[%dx%d C=%d]" % (ks, ks, x.shape[-1]))
def conv32_wgrad(x, dz, ks, out=None):
    B, H, W, c = x.shape
    assert x.is_contiguous() and dz.is_contiguous() and dz.dtype == torch.bfloat16 and dz.shape == x.shape and c in (16, 32)
    key = (int(ks), str(x.device))
    ws = _conv32_ws.get(key)
    if ws is None:
        ws = torch.empty([_lib.lib().isg_conv32_wgrad_workspace_bytes(int(ks)) // 4], dtype=torch.float32,
                         device=x.device)
        _conv32_ws[key] = ws
    dw = torch.empty([ks, ks, c, c], dtype=torch.float32, device=x.device) if out is None else out
    assert dw.dtype == torch.float32 and dw.is_contiguous() and tuple(dw.shape) == (ks, ks, c, c)
    fn = _lib.lib().isg_conv32_wgrad if c == 32 else _lib.lib().isg_conv16_wgrad
    rc = fn(_lib.ptr(x), _lib.ptr(dz), B, H, W, int(ks), _lib.ptr(ws), _lib.ptr(dw), _lib.current_stream())
    _lib.check(rc, "isg_conv32_wgrad")
    return dw


# ---- teacher-forcing batch builder -------------------------------------------------------------------------
def batch_points(table, counts, indices, starts, max_timesteps, hist, prediction_size, end_token):
    """isg_batch_points: (durations [B], points [B,T,hist,2], targets [B,T,2], rolled vertices [B,vcap,2],
    vertex counts [B]), all int32 device tensors, for samples ``indices`` rolled to ``starts`` of the padded vertex
    ``table`` int32 [N,vcap,2] / ``counts`` int32 [N]."""
    _need_cuda(table, "table")
    for t in (table, counts, indices, starts):
        assert t.dtype == torch.int32 and t.is_contiguous()
    B, T, vcap = indices.shape[0], int(max_timesteps), table.shape[1]
    dev = table.device
    durations = torch.empty([B], dtype=torch.int32, device=dev)
    points = torch.empty([B, T, int(hist), 2], dtype=torch.int32, device=dev)
    targets = torch.empty([B, T, 2], dtype=torch.int32, device=dev)
    rolled = torch.empty([B, vcap, 2], dtype=torch.int32, device=dev)
    nverts = torch.empty([B], dtype=torch.int32, device=dev)
    rc = _lib.lib().isg_batch_points(_lib.ptr(table), _lib.ptr(counts), vcap, _lib.ptr(indices), _lib.ptr(starts), B, T,
                                     int(hist), int(prediction_size), int(bool(end_token)), _lib.ptr(durations),
                                     _lib.ptr(points), _lib.ptr(targets), _lib.ptr(rolled), _lib.ptr(nverts),
                                     _lib.current_stream())
    _lib.check(rc, "isg_batch_points")
    return durations, points, targets, rolled, nverts
