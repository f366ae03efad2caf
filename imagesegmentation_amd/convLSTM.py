"""ConvLSTM cell and its stacked recurrence, with the reference's names and call shapes.

* ``ConvLSTMCell(shape, filters, kernel, initializer=None, forget_bias=1.0)`` with ``state_size``,
  ``output_size`` and ``cell(x, (c, h)) -> (output, LSTMStateTuple(c, h))``     reference convLSTM.py:5-45
* ``MultiRNNCell([cells])`` + ``dynamic_rnn(cell, inputs, sequence_length, initial_state,
  time_major=True) -> (outputs, final_state)``   the TensorFlow pieces the reference composes at
  polyrnn/History2LargeExperiment.py:31-46

Tensors are NHWC torch CUDA tensors.  Variables keep the reference's names and layouts
(``input_to_state_kernel`` [kh, kw, C, 4F], ``hidden_to_state_kernel`` [kh, kw, F, 4F], ``bias`` [4F],
gate blocks j|i|f|o) and are float32; ``compute_dtype`` selects float32 (f32 MFMA) or bfloat16 (bf16
MFMA, float32 accumulation and cell state) for the activations.  ``forget_bias`` is accepted and, as
in the reference, never used.
"""
import os
import collections
import math

import torch

from . import ops

LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


def glorot_uniform_(tensor, generator=None):
    """TensorFlow's default variable initializer for get_variable (glorot uniform) on an HWIO kernel.  ``generator``:
    the stream the numbers come from (a trainer whose ranks must hold the same weights passes its seeded one;
    None = torch's global stream)."""
    kh, kw, cin, cout = tensor.shape
    limit = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    with torch.no_grad():
        return tensor.uniform_(-limit, limit, generator=generator)


def _pad_channels(x, cs):
    c = x.shape[-1]
    if c == cs:
        return x.contiguous()
    out = torch.zeros(x.shape[:-1] + (cs,), dtype=x.dtype, device=x.device)
    out[..., :c] = x
    return out


class ConvLSTMCell(torch.nn.Module):
    def __init__(self, shape, filters, kernel, initializer=None, forget_bias=1.0, compute_dtype=torch.float32,
                 device=None):
        super().__init__()
        if list(kernel) != [3, 3]:
            raise NotImplementedError("the HIP cell implements the reference's 3x3 kernel only")
        if filters <= 0 or filters % ops.F_HIDDEN != 0:
            raise NotImplementedError("the HIP cell takes filters in multiples of %d (the reference's nets use 16, "
                                      "its a3c policy 32)" % ops.F_HIDDEN)
        self._kernel = list(kernel)
        self._filters = int(filters)
        self._initializer = initializer
        self._forget_bias = forget_bias  # stored, unused (reference convLSTM.py:15 vs :42)
        self._shape = list(shape)
        self.compute_dtype = compute_dtype
        self._device = device
        gates = 4 * self._filters
        self.hidden_to_state_kernel = torch.nn.Parameter(
            self._init(torch.empty(self._kernel + [self._filters, gates], device=device)))
        self.bias = torch.nn.Parameter(torch.zeros([gates], device=device))
        self.input_to_state_kernel = None  # created on first call, when the input channels are known
        self._packed = None
        self._packed_key = None

    def _init(self, t):
        if self._initializer is not None:
            with torch.no_grad():
                self._initializer(t)
            return t
        return glorot_uniform_(t)

    @property
    def state_size(self):
        size = torch.Size(self._shape + [self._filters])
        return LSTMStateTuple(size, size)

    @property
    def output_size(self):
        return torch.Size(self._shape + [self._filters])

    def build(self, channels, device=None):
        """Creates ``input_to_state_kernel`` for ``channels`` inputs (TensorFlow does this inside
        the first call through get_variable)."""
        if self.input_to_state_kernel is None:
            device = device if device is not None else self.hidden_to_state_kernel.device
            w = torch.empty(self._kernel + [int(channels), 4 * self._filters], device=device)
            self.input_to_state_kernel = torch.nn.Parameter(self._init(w))
        elif self.input_to_state_kernel.shape[2] != channels:
            raise ValueError("cell was built for %d input channels, got %d" %
                             (self.input_to_state_kernel.shape[2], channels))
        return self

    def zero_state(self, batch_size, device=None):
        device = device if device is not None else self.hidden_to_state_kernel.device
        shape = [batch_size] + self._shape + [self._filters]
        return LSTMStateTuple(torch.zeros(shape, dtype=torch.float32, device=device),
                              torch.zeros(shape, dtype=self.compute_dtype, device=device))

    def packed(self, hist=0):
        """Operand images of the current weights; rebuilt whenever a parameter changed in place.
        ``hist`` trailing input channels are one-hot history frames handled as a tap scatter."""
        params = (self.input_to_state_kernel, self.hidden_to_state_kernel, self.bias)
        reg = ops.active_derived()
        if reg is not None:  # a trainer that refreshes every derived tensor in one launch per step
            rkey = ("lstm",) + tuple(p.data_ptr() for p in params) + (self.compute_dtype, hist)
            packed = reg.get(rkey)
            if packed is not None:
                return packed
        key = tuple((p.data_ptr(), p._version) for p in params) + (self.compute_dtype, hist, ops.weights_epoch())
        if self._packed is None or self._packed_key != key:
            cs = ops.padded_channels(self.input_to_state_kernel.shape[2] - hist, self.compute_dtype)
            self._packed = ops.PackedWeights(self.input_to_state_kernel, self.hidden_to_state_kernel, self.bias, cs,
                                             self.compute_dtype, hist=hist)
            self._packed_key = key
            if reg is not None and reg.owns(*params):
                pw = self._packed
                reg.record(rkey, pw, ["fwd", "dgrad", "hist_table"], pw.repack, params)
        return self._packed

    def forward(self, x, h, scope=None):
        """One step: x [B, H, W, C]; h = (c, h) state tuple.  Returns (output, LSTMStateTuple(c, h))."""
        previous_memory, previous_output = h
        self.build(x.shape[-1], x.device)
        if self._filters != ops.F_HIDDEN:
            return self._forward_wide(x, previous_memory, previous_output)
        new_h, new_c = _CellFunction.apply(x, previous_memory, previous_output, self.input_to_state_kernel,
                                           self.hidden_to_state_kernel, self.bias, self)
        return new_h, LSTMStateTuple(new_c, new_h)


    def _forward_wide(self, x, c_prev, h_prev):
        """filters = 16 k (k > 1) as k launches of the 16-filter kernels: output channels
        [16 s, 16 s + 16) depend on all of x and all of h_prev but on no other output channel, so
        group s is a 16-filter cell whose input is x joined with h_prev[16:] (their weights joined
        the same way), whose recurrent input is h_prev[:16], and whose gate columns are the
        group's columns of each j|i|f|o block."""
        Fh, F = ops.F_HIDDEN, self._filters
        if x.shape[-1] + F - Fh > 64 - Fh:
            raise NotImplementedError("a %d-filter cell takes at most %d input channels" % (F, 64 - F))
        dt = self.compute_dtype
        x_wide = torch.cat([x.to(dt), h_prev[..., Fh:].to(dt)], dim=-1)
        h_head = h_prev[..., :Fh]
        hs, cs = [], []
        for s in range(F // Fh):
            cols = torch.cat([torch.arange(g * F + Fh * s, g * F + Fh * s + Fh) for g in range(4)]).to(x.device)
            w_h_cols = self.hidden_to_state_kernel.index_select(3, cols)
            w_x = torch.cat([self.input_to_state_kernel.index_select(3, cols), w_h_cols[:, :, Fh:]], dim=2)
            h_s, c_s = _CellFunction.apply(x_wide, c_prev[..., Fh * s:Fh * s + Fh], h_head, w_x,
                                           w_h_cols[:, :, :Fh].contiguous(), self.bias.index_select(0, cols), None,
                                           dt)
            hs.append(h_s)
            cs.append(c_s)
        new_h, new_c = torch.cat(hs, dim=-1), torch.cat(cs, dim=-1)
        return new_h, LSTMStateTuple(new_c, new_h)


class _CellFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c_prev, h_prev, w_x, w_h, bias, cell, dtype=None):
        if cell is not None:
            dt = cell.compute_dtype
            packed = cell.packed()
        else:  # one group of a wide cell: weights are slices, packed per call
            dt = dtype
            packed = ops.PackedWeights(w_x.detach(), w_h.detach(), bias.detach(),
                                       ops.padded_channels(w_x.shape[2], dt), dt)
        xs = _pad_channels(x.to(dt), packed.cs)
        hp = h_prev.to(dt).contiguous()
        cp = c_prev.to(torch.float32).contiguous()
        h, c, gates = ops.cell_fwd(xs, hp, cp, packed)
        ctx.save_for_backward(xs, hp, cp, c, gates)
        ctx.packed = packed
        ctx.in_dtypes = (x.dtype, c_prev.dtype, h_prev.dtype)
        ctx.c_real = x.shape[-1]
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        xs, hp, cp, c, gates = ctx.saved_tensors
        packed = ctx.packed
        dt = packed.dtype
        B, H, W, _ = xs.shape
        F = packed.f
        dh = dh.to(dt).contiguous() if dh is not None else None
        dc = dc.to(torch.float32).contiguous() if dc is not None else None
        dy, dc_prev = ops.pointwise_bwd(gates, cp, c, dh, H * W * F, None, dc)
        dw_x = torch.zeros([3, 3, ctx.c_real, 4 * F], dtype=torch.float32, device=xs.device)
        dw_h = torch.zeros([3, 3, F, 4 * F], dtype=torch.float32, device=xs.device)
        dbias = torch.zeros([4 * F], dtype=torch.float32, device=xs.device)
        ops.wgrad(xs, hp, dy, ctx.c_real, dw_x, dw_h, dbias, accumulate=False)
        dx = torch.empty_like(xs)
        dh_prev = torch.empty_like(hp)
        ops.dgrad(dy, packed, dh_prev, dx=dx, dx_channels=packed.cs)
        xd, cd, hd = ctx.in_dtypes
        return dx[..., :ctx.c_real].to(xd), dc_prev.to(cd), dh_prev.to(hd), dw_x, dw_h, dbias, None, None


class MultiRNNCell(torch.nn.Module):
    """Stack of cells: the output of cell l is the input of cell l+1 (tf.contrib.rnn.MultiRNNCell)."""

    def __init__(self, cells):
        super().__init__()
        self.cells = torch.nn.ModuleList(cells)

    @property
    def state_size(self):
        return tuple(c.state_size for c in self.cells)

    @property
    def output_size(self):
        return self.cells[-1].output_size

    def zero_state(self, batch_size, device=None):
        return tuple(c.zero_state(batch_size, device) for c in self.cells)

    def build(self, channels, device=None):
        for cell in self.cells:
            cell.build(channels, device)
            channels = cell._filters
        return self

    def forward(self, x, state):
        new_state = []
        for cell, s in zip(self.cells, state):
            x, s2 = cell(x, s)
            new_state.append(s2)
        return x, tuple(new_state)


class SequenceInput:
    """Where the stacked recurrence takes its layer-0 input from.

    Either ``x_seq`` [T, B, H, W, C] (generic, time-major), or the reference's tiled form
    (polyrnn/History2LargeExperiment.py:24-25): ``static`` [B, H, W, Cf] features repeated at every
    step, concatenated with that step's history frames, given densely (``hist_dense``
    [B, T, H, W, hist]) or as vertex points (``hist_points`` int32 [B, T, hist, 2], x < 0 = empty).
    The tiled tensor is never materialised: one [B, H, W, Cs] buffer is reused and only its history
    channels are rewritten per step.
    """

    def __init__(self, x_seq=None, static=None, hist_dense=None, hist_points=None):
        self.x_seq, self.static, self.hist_dense, self.hist_points = x_seq, static, hist_dense, hist_points
        if x_seq is not None:
            self.T, self.B, self.H, self.W, self.C = x_seq.shape
            self.hist = 0
            self.device = x_seq.device
        else:
            self.B, self.H, self.W, cf = static.shape
            self.device = static.device
            if hist_dense is not None:
                self.T, self.hist = hist_dense.shape[1], hist_dense.shape[4]
            else:
                self.T, self.hist = hist_points.shape[1], hist_points.shape[2]
            self.C = cf + self.hist
        self._buf = None
        self._buf_t = None
        #: a pass over this very input that has already run (``SequenceRecorder.record()``): the autograd node of
        #: ``dynamic_rnn`` adopts it instead of running the forward launches again
        self.recorded = None

    @property
    def scatter_hist(self):
        """History given as points: the cell takes the frames as a 9-tap weight scatter and the
        GEMM input is the static features alone."""
        return self.x_seq is None and self.hist_points is not None and self.hist > 0

    def layer0_packed(self, cell):
        return cell.packed(hist=self.hist if self.scatter_hist else 0)

    def prepare(self, dtype, cs):
        self.dtype, self.cs = dtype, cs
        if self.x_seq is not None:
            return
        cf = self.static.shape[-1]
        if self.hist_points is not None:
            self.hist_points = self.hist_points.to(torch.int32).contiguous()
        if cf == cs and self.static.dtype == dtype and self.static.is_contiguous() and (self.scatter_hist or
                                                                                         self.hist == 0):
            self._buf = self.static.detach()
            return
        self._buf = torch.zeros([self.B, self.H, self.W, cs], dtype=dtype, device=self.static.device)
        self._buf[..., :cf] = self.static.to(dtype)

    def points(self, t):
        """int32 [B, hist, 2] view of step t's frame points (scatter mode), else None."""
        return self.hist_points[:, t] if self.scatter_hist else None

    def get(self, t):
        """[B, H, W, Cs] input of step t, valid until the next call."""
        if self.x_seq is not None:
            return _pad_channels(self.x_seq[t].to(self.dtype), self.cs)
        if self._buf_t == t or self.hist == 0 or self.scatter_hist:
            return self._buf
        cf = self.static.shape[-1]
        self._buf[..., cf:cf + self.hist] = self.hist_dense[:, t].to(self.dtype)
        self._buf_t = t
        return self._buf


def _state_dtype(packed, B, H, W, dx_channels=None):
    """Storage type of a layer's cell state.  A bf16 layer that runs on the fused kernels (the shapes of every net
    of the reference) keeps it as bf16: the arithmetic of a step stays float and the new state is rounded once,
    when it is stored - 64 fewer bytes per pixel in the forward step, 32 in the backward step.  The rule does not
    depend on whether a pass keeps anything for backward, so that training, validation and step-by-step decoding
    of one model compute the same numbers.  Float layers, and bf16 layers on the three-kernel backward, keep float
    states."""
    if ops.bwd_fused_supported(packed, B, H, W, dx_channels) and ops.state_bf16_supported(packed, B, H, W):
        return torch.bfloat16
    return torch.float32


_ZERO_STATES = {}


def _zero_state(shape, dtype, device):
    key = (tuple(shape), dtype, str(device))
    z = _ZERO_STATES.get(key)
    if z is None:
        z = _ZERO_STATES[key] = torch.zeros(shape, dtype=dtype, device=device)
    return z


class _SequencePass:
    """The buffers of one pass over a sequence and the launches of one of its steps - shared by the whole-sequence
    forward (``_run_sequence_forward``) and by a loop that advances step by step while it keeps everything a backward
    sweep needs (``SequenceRecorder``)."""

    def __init__(self, cells, seq, t_run, init_state, save, pool=1):
        self.cells, self.seq, self.t_run, self.save, self.pool = cells, seq, t_run, save, pool
        L, T = len(cells), seq.T
        dt = cells[0].compute_dtype
        B, H, W, F = seq.B, seq.H, seq.W, cells[0]._filters
        dev = seq.device
        self.packed = packed = [seq.layer0_packed(cells[0])] + [c.packed() for c in cells[1:]]
        seq.prepare(dt, packed[0].cs)
        # Per layer a list of state tensors: entry 0 is the initial state, entry k the state after step k - 1 when
        # everything is kept for backward, else entries 1 and 2 alternate.  None = zero state: ONE zero tensor per
        # shape, kept for the life of the process and only ever read - neither a fill nor a copy per sequence.
        n_buf = T if save else 2
        self.Hs, self.Cs = [], []
        # a layer that starts from the zero state: its first step is told so and reads no state at all
        # (ISG_NO_ZERO_STATE=1: the shared zero tensors are read like any state, for A/B runs)
        self.from_zero = [init_state[l][0] is None and init_state[l][1] is None and
                          not os.environ.get("ISG_NO_ZERO_STATE") for l in range(L)]
        # What a layer keeps for backward (next to its states): the backward factors where the fused backward step
        # exists for its shape, else the gate activations for the three-kernel backward.
        # (Layer 0's input gradient has the width of the static features when the input is the tiled form: the fused
        # step writes it in runs of 8 channels, other widths take the three-kernel backward.)
        dx0 = seq.static.shape[-1] if seq.x_seq is None else None
        self.layouts = [ops.SAVE_FACTORS if ops.bwd_fused_supported(packed[l], B, H, W, dx0 if l == 0 else None)
                        else ops.SAVE_GATES for l in range(L)]
        cdts = [_state_dtype(packed[l], B, H, W, dx0 if l == 0 else None) for l in range(L)]
        for l in range(L):
            c0, h0 = init_state[l]
            c_init = _zero_state([B, H, W, F], cdts[l], dev) if c0 is None else c0.detach().to(cdts[l]).contiguous()
            h_init = _zero_state([B, H, W, F], dt, dev) if h0 is None else h0.detach().to(dt).contiguous()
            self.Hs.append([h_init] + list(torch.empty([n_buf, B, H, W, F], dtype=dt, device=dev).unbind(0)))
            self.Cs.append([c_init] + list(torch.empty([n_buf, B, H, W, F], dtype=cdts[l], device=dev).unbind(0)))
        self.Gs = [torch.empty([T] + ops.saved_shape(B, H, W, F, self.layouts[l]), dtype=dt, device=dev)
                   for l in range(L)] if save else None
        # every step that runs writes its whole slice (zeros for finished samples); only the tail needs clearing
        self.fused_pool = pool > 1 and ops.pooled_fwd_supported(packed[L - 1], B, H, W, pool)
        self.oh, self.ow = (H // pool, W // pool) if self.fused_pool else (H, W)
        self.out = torch.empty([B, T, self.oh, self.ow, F], dtype=dt, device=dev)
        if t_run < T:
            self.out[:, t_run:].zero_()
        self.cur = 0

    def step(self, t, durations_dev, t_mask=None):
        """The launches of step t.  ``t_mask``: the step index the per-sample lengths are compared with (default t)."""
        cells, seq, save = self.cells, self.seq, self.save
        L, T = len(cells), seq.T
        H, W, F = seq.H, seq.W, cells[0]._filters
        x = seq.get(t)
        cur = self.cur
        nxt = cur + 1 if save else (2 if cur == 1 else 1)
        for l in range(L):
            top = l == L - 1
            first = t == 0 and self.from_zero[l]
            ops.set_layer(l, ",zs" if first else "")
            ops.cell_fwd(x, None if first else self.Hs[l][cur], None if first else self.Cs[l][cur], self.packed[l],
                         durations_dev, t if t_mask is None else t_mask,
                         out=self.out[:, t] if top and not self.fused_pool else None, out_batch_stride=T * H * W * F,
                         save_gates=False, h_out=self.Hs[l][nxt], c_out=self.Cs[l][nxt],
                         gates=self.Gs[l][t] if save else None, save_layout=self.layouts[l],
                         hist_points=seq.points(t) if l == 0 else None,
                         pool_out=self.out[:, t] if top and self.fused_pool else None,
                         pool_batch_stride=T * self.oh * self.ow * F, pool=self.pool)
            x = self.Hs[l][nxt]
        self.cur = nxt
        ops.set_layer(0)

    def result(self, init_state):
        L = len(self.cells)
        out = self.out
        if self.pool > 1 and not self.fused_pool:
            out = ops.avgpool(out, self.pool)
        cur, Hs, Cs = self.cur, self.Hs, self.Cs
        if cur == 0:  # no step ran: hand out copies, never the shared zero tensors or the caller's own state
            # (of the state as it was given, where one was: not of its copy in the kernels' storage type)
            final = [(Cs[l][0].clone() if init_state[l][0] is None else init_state[l][0].detach().clone(),
                      Hs[l][0].clone() if init_state[l][1] is None else init_state[l][1].detach().clone())
                     for l in range(L)]
        else:
            final = [(Cs[l][cur], Hs[l][cur]) for l in range(L)]
        return out, final, (Hs, Cs, self.Gs, self.packed, self.layouts)


def _run_sequence_forward(cells, seq, durations_dev, t_run, init_state, save, pool=1):
    """Shared by the autograd function and the no-grad decode path.  ``pool`` = k > 1: the returned outputs
    are the k x k window means, [B, T, H/k, W/k, F] - written by the top layer's kernel itself where it can,
    by a pooling pass over the full-resolution outputs otherwise."""
    run = _SequencePass(cells, seq, t_run, init_state, save, pool)
    for t in range(t_run):
        run.step(t, durations_dev)
    return run.result(init_state)


class SequenceRecorder:
    """A rollout's recurrence advanced one step at a time from the zero state - each step's history points are only
    known once the previous step's action has been taken - that keeps what the backward sweep of ``dynamic_rnn``
    reads (states, backward factors, outputs of every step).  ``SequenceInput.recorded = recorder.record()`` on the
    sequence input of the update's ``dynamic_rnn`` then makes that node adopt the pass instead of running it again:
    the same launches on the same inputs would write the same numbers (reference
    polyrl/ReinforceOverfitExperiment.py:147-176 runs the network a second time for the update).  Every sample runs
    every step here (a finished episode's later steps are ignored by the update's per-sample lengths: its backward
    sweep skips them and their outputs meet zero weights in the loss)."""

    def __init__(self, multi, static, hist, steps):
        cells = list(multi.cells) if isinstance(multi, MultiRNNCell) else [multi]
        B = static.shape[0]
        (multi if isinstance(multi, MultiRNNCell) else MultiRNNCell(cells)).build(static.shape[-1] + hist, static.device)
        self.hist_points = torch.full([B, steps, hist, 2], -1, dtype=torch.int32, device=static.device)
        self.seq = SequenceInput(static=static, hist_points=self.hist_points)
        self.init_state = [(None, None)] * len(cells)
        self.run = _SequencePass(cells, self.seq, steps, self.init_state, True)
        self.t = 0

    def step(self, hist_points=None):
        """Advances every layer once with this step's history points (int32 [B, hist, 2]).  Returns the top layer's
        output of the step, [B, 1, H, W, F] (a view of the kept outputs)."""
        t = self.t
        if self.seq.hist > 0:
            self.seq.hist_points[:, t] = hist_points
        self.run.step(t, None)
        self.t = t + 1
        return self.run.out[:, t:t + 1]

    def record(self):
        """What ``SequenceInput.recorded`` takes; the recorder must have run all of its steps."""
        if self.t != self.seq.T:
            raise RuntimeError("the recorder ran %d of its %d steps" % (self.t, self.seq.T))
        return self.run.result(self.init_state)


class RecurrentStepper:
    """The stacked cells advanced one step at a time, without autograd, for decode loops and rollouts
    (reference polyrnn/Model.py:231-249, polyrl/ReinforceOverfitExperiment.py:136-145 feed the state back
    through placeholders).  The state lives in two ping-pong buffers per layer: a step is one fused
    cell launch per layer and no copy.  Layer 0 takes ``static`` features [B,H,W,Cf] plus the step's
    history points (int32 [B,hist,2], x < 0 = empty frame)."""

    def __init__(self, multi, static, hist, initial_state=None):
        cells = list(multi.cells) if isinstance(multi, MultiRNNCell) else [multi]
        self.cells = cells
        B, H, W, cf = static.shape
        dev = static.device
        dt = cells[0].compute_dtype
        F = cells[0]._filters
        (multi if isinstance(multi, MultiRNNCell) else MultiRNNCell(cells)).build(cf + hist, dev)
        self.hist = int(hist)
        pts = torch.full([B, 1, self.hist, 2], -1, dtype=torch.int32, device=dev)
        self._seq = SequenceInput(static=static, hist_points=pts)
        self.packed = [self._seq.layer0_packed(cells[0])] + [c.packed() for c in cells[1:]]
        self._seq.prepare(dt, self.packed[0].cs)
        # (the cell state in the type the sequence node stores it in: a decode loop repeats a training pass's numbers)
        cdts = [_state_dtype(self.packed[l], B, H, W, cf if l == 0 else None) for l in range(len(cells))]
        # one zero fill per storage type instead of one per buffer (a decode loop at a small batch is a chain of
        # launches of a few microseconds each: every launch counts)
        n, L = B * H * W * F, len(cells)
        n16 = (2 * L + 1) * n if dt == torch.bfloat16 else 0
        flat16 = torch.zeros([n16 + 2 * n * sum(1 for c in cdts if c == torch.bfloat16)], dtype=torch.bfloat16, device=dev)
        flat32 = torch.zeros([(0 if dt == torch.bfloat16 else (2 * L + 1) * n) +
                              2 * n * sum(1 for c in cdts if c == torch.float32)], dtype=torch.float32, device=dev)
        used = {torch.bfloat16: 0, torch.float32: 0}
        pools = {torch.bfloat16: flat16, torch.float32: flat32}

        def take(count, dtype, shape):
            o = used[dtype]
            used[dtype] = o + count
            return pools[dtype][o:o + count].view(shape)
        self.H = [take(2 * n, dt, [2, B, H, W, F]) for _ in cells]
        self.C = [take(2 * n, cdts[l], [2, B, H, W, F]) for l in range(L)]
        self.out = take(n, dt, [B, 1, H, W, F])
        self.cur = 0
        self._stride = H * W * F
        if initial_state is not None:
            for l, (c, h) in enumerate(initial_state):
                self.C[l][0].copy_(c)
                self.H[l][0].copy_(h)

    @property
    def state(self):
        """((c, h), ...) views of the current state; overwritten two steps later."""
        return tuple(LSTMStateTuple(self.C[l][self.cur], self.H[l][self.cur]) for l in range(len(self.cells)))

    def step(self, hist_points=None, durations=None, t=0):
        """Advances every layer once.  Returns the top layer's emitted output [B,1,H,W,F] (zero for
        samples with t >= durations; a buffer reused by the next step)."""
        cur, nxt = self.cur, 1 - self.cur
        x = self._seq.get(0)
        L = len(self.cells)
        pts = hist_points.to(torch.int32).contiguous() if self.hist else None
        for l in range(L):
            ops.cell_fwd(x, self.H[l][cur], self.C[l][cur], self.packed[l], durations, t,
                         out=self.out if l == L - 1 else None, out_batch_stride=self._stride, save_gates=False,
                         h_out=self.H[l][nxt], c_out=self.C[l][nxt], hist_points=pts if l == 0 else None)
            x = self.H[l][nxt]
        self.cur = nxt
        return self.out


class WideStepper:
    """One cell of 16 k filters advanced one step at a time, without autograd (acting in a rollout): the state stays
    split by 16-filter group and nothing is joined by a copy - the joint input [x | h[16:]] lives in two buffers that
    alternate from step to step, every further group writes its h straight into the NEXT step's buffer (the cell
    kernel's strided h output) and every group writes its 16 channels of the joint output (its strided masked
    output).  A step is one copy of x and k fused cell launches, where a ``dynamic_rnn`` call per step would split
    and rejoin the whole state every time."""

    def __init__(self, cell, batch, device, initial_state=None):
        self.cell = cell
        Fh, F, dt = ops.F_HIDDEN, cell._filters, cell.compute_dtype
        H, W = cell._shape
        self.k = F // Fh
        self.C = int(cell.input_to_state_kernel.shape[2])
        self.groups = _wide_groups(cell, self.C)
        cs = self.groups[0][1].cs
        self.xw = torch.zeros([2, batch, H, W, cs], dtype=dt, device=device)   # [x | h of groups 1..k-1 | padding]
        self.H0 = torch.zeros([2, batch, H, W, Fh], dtype=dt, device=device)   # group 0's h: the recurrent input
        self.Cst = [torch.zeros([2, batch, H, W, Fh], dtype=torch.float32, device=device) for _ in range(self.k)]
        self.out = torch.empty([batch, H, W, F], dtype=dt, device=device)
        self.cur = 0
        if initial_state is not None:
            c0, h0 = initial_state
            self.H0[0].copy_(h0[..., :Fh])
            self.xw[0][..., self.C:self.C + F - Fh] = h0[..., Fh:]
            for g in range(self.k):
                self.Cst[g][0].copy_(c0[..., Fh * g:Fh * g + Fh])

    def reset(self):
        """Back to the zero state, in place (the buffers keep their addresses: a captured graph may hold them)."""
        for t in [self.xw, self.H0] + self.Cst:
            t.zero_()
        self.cur = 0

    @property
    def state(self):
        """(c, h) of the whole cell, joined (a copy)."""
        cur, C, F, Fh = self.cur, self.C, self.cell._filters, ops.F_HIDDEN
        return LSTMStateTuple(torch.cat([c[cur] for c in self.Cst], dim=-1),
                              torch.cat([self.H0[cur], self.xw[cur][..., C:C + F - Fh]], dim=-1))

    def step(self, x):
        """x [B, H, W, C] -> the cell's output h [B, H, W, F] of this step: a buffer of the stepper that the next
        step overwrites."""
        Fh, C, F = ops.F_HIDDEN, self.C, self.cell._filters
        cur, nxt = self.cur, 1 - self.cur
        groups = _wide_groups(self.cell, C)  # (re-packed when the weights have changed)
        xw, cs = self.xw[cur], self.xw.shape[-1]
        _, H, W, _ = xw.shape
        ops.copy_slice(xw[..., :C], x if x.dtype == xw.dtype else x.to(xw.dtype))
        for g in range(self.k):
            h_out = self.H0[nxt] if g == 0 else self.xw[nxt][..., C + Fh * (g - 1):C + Fh * g]
            ops.cell_fwd(xw, self.H0[cur], self.Cst[g][cur], groups[g][1], None, 0,
                         out=self.out[..., Fh * g:Fh * g + Fh], out_batch_stride=H * W * F, out_pixel_stride=F,
                         save_gates=False, h_out=h_out, h_pixel_stride=Fh if g == 0 else cs, c_out=self.Cst[g][nxt])
        self.cur = nxt
        return self.out


class WideRecorder:
    """``WideStepper`` for a rollout whose update follows under the same weights: the cell is advanced one step at a
    time from the zero state and everything the backward sweep of ``dynamic_rnn`` reads is kept (joint inputs, states
    and backward factors of every step and group, the joint outputs), in buffers that stay where they are - a
    captured graph may replay the loop.  ``SequenceInput.recorded = recorder.record()`` on the input of the update's
    ``dynamic_rnn`` makes that node adopt the pass instead of launching the same steps on the same inputs again."""

    def __init__(self, cell, batch, device, steps):
        self.cell, self.T = cell, int(steps)
        Fh, F, dt = ops.F_HIDDEN, cell._filters, cell.compute_dtype
        H, W = cell._shape
        T, B = self.T, batch
        self.k = k = F // Fh
        self.C = C = int(cell.input_to_state_kernel.shape[2])
        self.groups = _wide_groups(cell, C)
        cs = self.groups[0][1].cs
        self.layout = ops.SAVE_FACTORS if ops.bwd_fused_supported(self.groups[0][1], B, H, W) else ops.SAVE_GATES
        # (one slot more than steps: a further group's h is written where the NEXT step reads it, inside that step's
        # joint input - also after the last step - and the backward sweep reads it there; the h slots of step 0 are
        # the zero state and are never written)
        self.xw = torch.zeros([T + 1, B, H, W, cs], dtype=dt, device=device)
        self.Hs = [[torch.zeros([B, H, W, Fh], dtype=dt, device=device)] +
                   list(torch.empty([T, B, H, W, Fh], dtype=dt, device=device).unbind(0))] + \
                  [[self.xw[t, ..., C + Fh * (s - 1):C + Fh * s] for t in range(T + 1)] for s in range(1, k)]
        self.Cs = [[torch.zeros([B, H, W, Fh], dtype=torch.float32, device=device)] +
                   list(torch.empty([T, B, H, W, Fh], dtype=torch.float32, device=device).unbind(0)) for _ in range(k)]
        self.Gs = [torch.empty([T] + ops.saved_shape(B, H, W, Fh, self.layout), dtype=dt, device=device)
                   for _ in range(k)]
        # the joint outputs, stored time-major (a step's output is then one contiguous block: what ``step`` returns)
        # and handed to the update as the batch-major view of the same memory
        self.out_tm = torch.empty([T, B, H, W, F], dtype=dt, device=device)
        self.out = self.out_tm.transpose(0, 1)
        self.t = 0

    def reset(self):
        """Back to the first step of the zero state (nothing to clear: every step rewrites its slots, and the slots of
        the zero state are never written)."""
        self.t = 0

    def input_slots(self):
        """[T, B, H, W, C]: where the steps' inputs live, the first C channels of the joint input's wider pixels.  A
        producer that can write pixels that far apart (``ops.conv32(out=)``) fills ``input_slots()[t]`` itself and hands
        that very view to ``step``, which then copies nothing."""
        return self.xw[:self.T, ..., :self.C]

    def step(self, x):
        """x [B, H, W, C] -> the cell's output h [B, H, W, F] of this step (its block of the kept outputs)."""
        Fh, C, F, t, T = ops.F_HIDDEN, self.C, self.cell._filters, self.t, self.T
        if t >= T:
            raise RuntimeError("the recorder holds %d steps" % T)
        groups = _wide_groups(self.cell, C)
        xw = self.xw[t]
        _, H, W, _ = xw.shape
        slot = xw[..., :C]
        if not (x.data_ptr() == slot.data_ptr() and x.stride() == slot.stride() and x.dtype == xw.dtype):
            ops.copy_slice(slot, x if x.dtype == xw.dtype else x.to(xw.dtype))  # (else: written in place, ``input_slots``)
        for s in range(self.k):
            ops.cell_fwd(xw, self.Hs[0][t], self.Cs[s][t], groups[s][1], None, t,
                         out=self.out_tm[t, ..., Fh * s:Fh * s + Fh], out_batch_stride=H * W * F, out_pixel_stride=F,
                         save_gates=False, h_out=self.Hs[s][t + 1], h_pixel_stride=Fh if s == 0 else xw.shape[-1],
                         c_out=self.Cs[s][t + 1], gates=self.Gs[s][t], save_layout=self.layout)
        self.t = t + 1
        return self.out_tm[t]

    def record(self):
        """What ``SequenceInput.recorded`` takes once all steps have run."""
        if self.t != self.T:
            raise RuntimeError("the recorder ran %d of its %d steps" % (self.t, self.T))
        return (self.xw[:self.T], self.Hs, self.Cs, self.Gs, _wide_groups(self.cell, self.C), self.layout, self.out)


class _SequenceFunction(torch.autograd.Function):
    """MultiRNNCell + dynamic_rnn as ONE autograd node: T*L fused cell launches forward, and the
    hand-written reverse sweep (pointwise, weight-gradient and data-gradient kernels) backward."""

    @staticmethod
    def forward(ctx, multi, seq, durations_dev, t_run, pool, *tensors):
        # tensors = [layer-0 differentiable input] + [c0, h0] * L + [w_x, w_h, bias] * L
        cells = list(multi.cells)
        L = len(cells)
        init_state = [(tensors[1 + 2 * l], tensors[2 + 2 * l]) for l in range(L)]
        recorded = getattr(seq, "recorded", None)
        if recorded is not None:
            # a pass over the same inputs with the same weights that already ran step by step (SequenceRecorder)
            if pool != 1 or any(c is not None or h is not None for c, h in init_state) or t_run != seq.T:
                raise NotImplementedError("a recorded pass: zero initial state, no pooled output, all steps run")
            out, final, (Hs, Cs, Gs, packed, layouts) = recorded
            seq.prepare(cells[0].compute_dtype, packed[0].cs)
        else:
            out, final, (Hs, Cs, Gs, packed, layouts) = _run_sequence_forward(cells, seq, durations_dev, t_run,
                                                                              init_state, True, pool)
        ctx.multi, ctx.seq, ctx.durations_dev, ctx.t_run = multi, seq, durations_dev, t_run
        ctx.saved = (Hs, Cs, Gs, packed, layouts)
        ctx.in_dtype = tensors[0].dtype
        ctx.pool = pool
        ctx.state_dtypes = [(None if c is None else c.dtype, None if h is None else h.dtype) for c, h in init_state]
        # unused outputs (normally the final state) arrive as None in backward instead of zero tensors
        ctx.set_materialize_grads(False)
        flat = [out]
        for c, h in final:  # views of the saved state buffers: do not modify them in place
            flat += [c, h]
        return tuple(flat)

    @staticmethod
    def backward(ctx, d_out, *d_final):
        cells = list(ctx.multi.cells)
        L = len(cells)
        seq, durations_dev, t_run = ctx.seq, ctx.durations_dev, ctx.t_run
        if ctx.saved is None:
            raise RuntimeError("the recurrence's sequence node runs backward once (its saved states are released "
                               "after the sweep): run the forward pass again instead of retain_graph=True")
        Hs, Cs, Gs, packed, layouts = ctx.saved
        dt = cells[0].compute_dtype
        B, H, W, F, T = seq.B, seq.H, seq.W, cells[0]._filters, seq.T
        dev = Hs[0][0].device
        pool = ctx.pool
        d_out = d_out.to(dt).contiguous() if d_out is not None else \
            torch.zeros([B, T, H // pool, W // pool, F], dtype=dt, device=dev)
        dc_state, dh_state = [], []
        fused = [layouts[l] == ops.SAVE_FACTORS for l in range(L)]
        # a layer whose final state has no gradient tells the first step of its sweep so (nothing is read for it)
        # where the fused backward step runs; elsewhere it starts from zeros
        no_carry = [fused[l] and t_run > 0 and d_final[2 * l] is None and d_final[2 * l + 1] is None and
                    not os.environ.get("ISG_NO_ZERO_STATE") for l in range(L)]
        for l in range(L):
            dc, dh = d_final[2 * l], d_final[2 * l + 1]
            make = torch.empty if no_carry[l] else torch.zeros
            # the cell-state gradient is carried between steps in the activation dtype, like dL/dh
            dc_state.append(dc.to(dt).clone().contiguous() if dc is not None else make([B, H, W, F], dtype=dt, device=dev))
            dh_state.append(dh.to(dt).clone().contiguous() if dh is not None else make([B, H, W, F], dtype=dt, device=dev))
        grads = []
        # (every entry is assigned by the sweep - the history taps' rows by its first step, the rest by the closing
        # fold of the partial sums - so only a sweep of no steps needs the zeros)
        # A parameter with a gradient sink (ops.grad_sinks: a view of the trainer's flat gradient buffer) gets its
        # gradient written there by the sweep's kernels, and autograd gets None for it.
        like = torch.empty_like if t_run > 0 else torch.zeros_like
        sunk = []
        for cell in cells:
            params = (cell.input_to_state_kernel, cell.hidden_to_state_kernel, cell.bias)
            dsts = [ops.grad_destination(prm) for prm in params]
            sunk.append([d is not None for d in dsts])
            if t_run == 0:
                for d in dsts:
                    if d is not None:
                        d.zero_()
            grads.append(tuple(d if d is not None else like(prm, dtype=torch.float32) for d, prm in zip(dsts, params)))
        dy = torch.empty([B, H, W, F, 4], dtype=dt, device=dev)
        dx_mid = torch.empty([B, H, W, F], dtype=dt, device=dev)
        cs0 = packed[0].cs
        if seq.x_seq is not None:
            d_in = torch.zeros([T, B, H, W, cs0], dtype=dt, device=dev)
            dx0_channels = cs0
        else:
            cf = seq.static.shape[-1]
            # the first step of the sweep (t_run - 1) overwrites, the others add: no zero fill
            d_in = (torch.empty if t_run > 0 else torch.zeros)([B, H, W, cf], dtype=dt, device=dev)
            dx0_channels = cf
        seq._buf_t = None
        # per-workgroup weight-gradient sums stay in a private workspace over the whole sweep
        slabs = [ops.wgrad_workspace(packed[l].cs, F, dev, cached=False) for l in range(L)]
        # One launch per (step, layer) where the fused backward kernel exists (bf16): it reads the incoming state
        # gradients of neighbouring tiles' border pixels, so they alternate between two buffers per layer.
        dc_next = [torch.empty_like(dc_state[l]) if fused[l] else None for l in range(L)]
        dh_next = [torch.empty_like(dh_state[l]) if fused[l] else None for l in range(L)]
        # history taps of a fused layer: their dY rows are kept per step and folded into dw_x once after the sweep
        hist_scratch = [ops.bwd_fused_hist_begin(B, t_run, packed[l], dev) if fused[l] and packed[l].hist and t_run > 0
                        else None for l in range(L)]
        for t in reversed(range(t_run)):
            x0 = seq.get(t)
            for l in reversed(range(L)):
                ops.set_layer(l)
                x_in = x0 if l == 0 else Hs[l - 1][t + 1]
                if l == L - 1:
                    above, stride, k = d_out[:, t], T * (H // pool) * (W // pool) * F, pool
                else:
                    above, stride, k = dx_mid, H * W * F, 1
                mode = ops.WGRAD_DEFER if t < t_run - 1 else ops.WGRAD_DEFER_FIRST
                if l > 0:
                    dx_args = dict(dx=dx_mid, dx_channels=F)
                elif seq.x_seq is not None:
                    dx_args = dict(dx=d_in[t], dx_channels=dx0_channels)
                elif dx0_channels % 4 == 0:
                    dx_args = dict(dx=d_in, dx_channels=dx0_channels, dx_accumulate=t < t_run - 1)
                else:
                    raise NotImplementedError("static feature channels must be a multiple of 4")
                pts = seq.points(t) if l == 0 else None
                if fused[l]:
                    # the first step of a layer that started from the zero state: no state to read, and nobody
                    # wants the gradient of that state
                    first = t == 0 and ctx.state_dtypes[l] == (None, None) and not os.environ.get("ISG_NO_ZERO_STATE")
                    last = t == t_run - 1 and no_carry[l]
                    ops.set_layer(l, (",zs" if first else "") + (",edge" if last else ""))
                    ops.bwd_fused(Gs[l][t], None if first else Cs[l][t], Hs[l][t + 1], above, stride,
                                  None if last else dh_state[l], None if last else dc_state[l], x_in,
                                  None if first else Hs[l][t], packed[l],
                                  None if first else dc_next[l], None if first else dh_next[l], slabs[l], mode,
                                  grads[l][0], durations=durations_dev, t=t, hist_points=pts, pool=k,
                                  hist_slot=(hist_scratch[l], t) if hist_scratch[l] is not None else None, **dx_args)
                    dc_state[l], dc_next[l] = dc_next[l], dc_state[l]
                    dh_state[l], dh_next[l] = dh_next[l], dh_state[l]
                    continue
                ops.pointwise_bwd(Gs[l][t], Cs[l][t], Cs[l][t + 1], above, stride, dh_state[l], dc_state[l],
                                  durations_dev, t, dy=dy, dc_prev=dc_state[l], pool=k)
                ops.wgrad(x_in, Hs[l][t], dy, packed[l].c, grads[l][0], grads[l][1], grads[l][2], True, durations_dev, t,
                          hist_points=pts, hist=packed[l].hist, mode=mode, workspace=slabs[l])
                ops.dgrad(dy, packed[l], dh_state[l], durations=durations_dev, t=t, **dx_args)
        ops.set_layer(0)
        if t_run > 0:
            for l in range(L):
                ops.wgrad_finish(slabs[l], (B, H, W, packed[l].cs), packed[l].c, packed[l].hist, *grads[l])
                if hist_scratch[l] is not None:
                    ops.bwd_fused_hist_finish(hist_scratch[l], B, t_run, packed[l], grads[l][0])
        if seq.x_seq is not None:
            d_input = d_in[..., :seq.C].to(ctx.in_dtype)
        else:
            d_input = d_in.to(ctx.in_dtype)
        result = [None, None, None, None, None, d_input]
        for l in range(L):
            cd, hd = ctx.state_dtypes[l]
            result += [None if cd is None else dc_state[l].to(cd), None if hd is None else dh_state[l].to(hd)]
        for l in range(L):
            params = (cells[l].input_to_state_kernel, cells[l].hidden_to_state_kernel, cells[l].bias)
            for k in range(3):
                if sunk[l][k]:
                    ops.grad_delivered(params[k])
                    result.append(None)
                else:
                    result.append(grads[l][k])
        # The returned final states are entries of the saved buffers, whose grad_fn is this node: a reference cycle
        # through ctx.saved that only the garbage collector would break, gigabytes at a time.  Nothing runs
        # backward through this node twice.
        ctx.saved = ctx.seq = ctx.multi = None
        return tuple(result)


def _wide_groups(cell, C):
    """A cell of 16 k filters (k > 1) as k groups of 16: the output channels [16 s, 16 s + 16) depend on all of
    x and all of h_prev but on no other output channel, so group s is a 16-filter cell whose input is x joined
    with h_prev[16:] (their weights joined the same way), whose recurrent input is h_prev[:16], and whose gate
    columns are the group's columns of each j|i|f|o block.  The groups' packed weight images are cached until
    the weights change."""
    params = (cell.input_to_state_kernel, cell.hidden_to_state_kernel, cell.bias)
    Fh, F, dt = ops.F_HIDDEN, cell._filters, cell.compute_dtype
    reg = ops.active_derived()
    rkeys = [("wide",) + tuple(p.data_ptr() for p in params) + (dt, C, s) for s in range(F // Fh)]
    if reg is not None:  # a trainer that refreshes every derived tensor in one launch per optimiser step
        hit = [reg.get(k) for k in rkeys]
        if all(h is not None for h in hit):
            return [(h.cols, h) for h in hit]
    key = tuple((p.data_ptr(), p._version) for p in params) + (dt, ops.weights_epoch())
    cached = getattr(cell, "_wide_cache", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    cw = C + F - Fh
    if cw > 64 - Fh:
        raise NotImplementedError("a %d-filter cell takes at most %d input channels" % (F, 64 - F))

    def group_weights(cols):
        w_h_cols = params[1].detach().index_select(3, cols)
        w_x = torch.cat([params[0].detach().index_select(3, cols), w_h_cols[:, :, Fh:]], dim=2).contiguous()
        return w_x, w_h_cols[:, :, :Fh].contiguous(), params[2].detach().index_select(0, cols)

    groups = []
    with torch.no_grad():
        for s in range(F // Fh):
            cols = torch.cat([torch.arange(g * F + Fh * s, g * F + Fh * s + Fh) for g in range(4)]).to(params[0].device)
            packed = ops.PackedWeights(*group_weights(cols), ops.padded_channels(cw, dt), dt)
            packed.cols = cols
            groups.append((cols, packed))
            if reg is not None and reg.owns(*[p.detach() for p in params]):
                def again(packed=packed, cols=cols):  # the group's tensors from the parameters as they are now
                    with torch.no_grad():
                        packed._w_x, packed._w_h, bias = group_weights(cols)
                        packed.bias.copy_(bias)
                        packed.repack()
                reg.record(rkeys[s], packed, ["fwd", "dgrad", "bias"], again, params)
    cell._wide_cache = (key, groups)
    return groups


class _WideSequenceFunction(torch.autograd.Function):
    """dynamic_rnn of ONE cell of 16 k filters as one autograd node: per step one fused forward launch per group
    on a shared [x | h_prev[16:]] buffer, and a hand-written reverse sweep with the 16-filter backward kernels.
    The state is kept per group ([B, H, W, 16] each) so that nothing is sliced or re-packed inside the loops."""

    @staticmethod
    def forward(ctx, cell, durations_dev, t_run, x_seq, c0, h0, w_x, w_h, bias, recorded=None):
        T, B, H, W, C = x_seq.shape
        Fh, F, dt, dev = ops.F_HIDDEN, cell._filters, cell.compute_dtype, x_seq.device
        k = F // Fh
        if recorded is not None:
            # a pass over the same inputs from the zero state with the same weights that already ran step by step
            # (WideRecorder): adopted as it is
            if durations_dev is not None or t_run != T:
                raise NotImplementedError("a recorded pass: all steps of every sample")
            xw, Hs, Cs, Gs, groups, layout, out = recorded
            ctx.cell, ctx.durations_dev, ctx.t_run, ctx.C = cell, None, t_run, C
            ctx.saved = (xw, Hs, Cs, Gs, groups, layout)
            ctx.in_dtypes = (x_seq.dtype, c0.dtype, h0.dtype)
            ctx.batch_major = T > 1 and B > 1 and x_seq.transpose(0, 1).is_contiguous()
            ctx.x_in_slots = (x_seq.data_ptr() == xw.data_ptr() and x_seq.dtype == xw.dtype and
                              x_seq.stride() == xw[..., :C].stride())
            ctx.set_materialize_grads(False)
            # (the final state of an adopted pass is not joined: it sits in the recorder, per group, for who wants it)
            return out, None, None
        groups = _wide_groups(cell, C)
        cs = groups[0][1].cs
        keep = not getattr(ctx, "forward_only", False)  # without a backward pass no gates are stored
        xw = torch.empty([max(t_run, 1), B, H, W, cs], dtype=dt, device=dev)
        if cs > C + F - Fh or t_run == 0:
            xw[..., C + F - Fh:] = 0  # padding channels (they meet zero weights, but must be finite)
        ops.copy_slice(xw[:t_run, ..., :C], x_seq[:t_run] if x_seq.dtype == dt else x_seq[:t_run].to(dt))
        # (a batch-major source seen time-major: the gradient goes back in the same layout, so that the caller's
        # transpose and reshape of it are views)
        ctx.batch_major = T > 1 and B > 1 and x_seq.transpose(0, 1).is_contiguous()
        Hs = [[h0[..., Fh * s:Fh * s + Fh].to(dt).contiguous()] +
              list(torch.empty([t_run, B, H, W, Fh], dtype=dt, device=dev).unbind(0)) for s in range(k)]
        Cs = [[c0[..., Fh * s:Fh * s + Fh].to(torch.float32).contiguous()] +
              list(torch.empty([t_run, B, H, W, Fh], dtype=torch.float32, device=dev).unbind(0)) for s in range(k)]
        # kept for backward: the backward factors where the fused backward step exists for the groups' shape
        layout = ops.SAVE_FACTORS if ops.bwd_fused_supported(groups[0][1], B, H, W) else ops.SAVE_GATES
        Gs = [torch.empty([t_run] + ops.saved_shape(B, H, W, Fh, layout), dtype=dt, device=dev) if keep else None
              for _ in range(k)]
        # every step that runs writes its whole output slice (zeros for finished samples), each group its 16 channels
        # of the joint output: only steps past the longest sequence need the fill
        out = torch.empty([B, T, H, W, F], dtype=dt, device=dev)
        if t_run < T:
            out[:, t_run:] = 0
        live = None
        for t in range(t_run):
            for s in range(1, k):
                ops.copy_slice(xw[t, ..., C + Fh * (s - 1):C + Fh * s], Hs[s][t])
            if durations_dev is not None:
                live = (durations_dev > t).view(-1, 1, 1, 1)
            for s in range(k):
                ops.cell_fwd(xw[t], Hs[0][t], Cs[s][t], groups[s][1], durations_dev, t,
                             out=out[:, t, ..., Fh * s:Fh * s + Fh], out_batch_stride=T * H * W * F, out_pixel_stride=F,
                             save_gates=False, h_out=Hs[s][t + 1], c_out=Cs[s][t + 1],
                             gates=Gs[s][t] if keep else None, save_layout=layout)
                if s > 0 and live is not None:
                    # a finished sample's state passes through; the kernel copied its RECURRENT input (group 0's h)
                    Hs[s][t + 1].copy_(torch.where(live, Hs[s][t + 1], Hs[s][t]))
        ctx.cell, ctx.durations_dev, ctx.t_run, ctx.C = cell, durations_dev, t_run, C
        ctx.saved = (xw, Hs, Cs, Gs, groups, layout)
        ctx.in_dtypes = (x_seq.dtype, c0.dtype, h0.dtype)
        ctx.set_materialize_grads(False)
        return out, torch.cat([Cs[s][t_run] for s in range(k)], dim=-1), torch.cat([Hs[s][t_run] for s in range(k)], dim=-1)

    @staticmethod
    def backward(ctx, d_out, d_c, d_h):
        cell, durations_dev, t_run, C = ctx.cell, ctx.durations_dev, ctx.t_run, ctx.C
        if ctx.saved is None:
            raise RuntimeError("the recurrence's sequence node runs backward once (its saved states are released "
                               "after the sweep): run the forward pass again instead of retain_graph=True")
        xw, Hs, Cs, Gs, groups, layout = ctx.saved
        Fh, F, dt = ops.F_HIDDEN, cell._filters, cell.compute_dtype
        k = F // Fh
        _, B, H, W, cs = xw.shape
        dev = xw.device
        T = d_out.shape[1] if d_out is not None else t_run
        part = lambda g, s, d: (g[..., Fh * s:Fh * s + Fh].to(d).contiguous() if g is not None else
                                torch.zeros([B, H, W, Fh], dtype=d, device=dev))
        # one launch per (step, group) where the fused backward step exists for the groups' shape (its carried
        # cell-state gradient is in the compute type and ping-pongs between two buffers), else three
        fused = t_run > 0 and layout == ops.SAVE_FACTORS
        # (an initial state that takes no gradient - the zero state of a rollout: nothing is joined for it after the
        # first step's launches)
        need_state = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        dc_state = [part(d_c, s, dt if fused else torch.float32) for s in range(k)]
        dh_state = [part(d_h, s, dt) for s in range(k)]
        if fused:
            dc_next = [torch.empty_like(g) for g in dc_state]
            dh_next0 = torch.empty_like(dh_state[0])
        d_out_in_place = False
        if d_out is not None:
            d_out_c = d_out if d_out.dtype == dt else d_out.to(dt)
            # the fused step of the 48-channel kernel reads a group's 16 channels of the joint gradient where they are
            d_out_in_place = (fused and cs == 48 and tuple(d_out_c.stride()[2:]) == (W * F, F, 1) and
                              d_out_c.stride(0) % 8 == 0 and not os.environ.get("ISG_NO_STRIDED_BWD"))
        if d_out_in_place:
            d_out_g = None
        elif d_out is not None:  # group-major: one launch of 16-byte moves per group for the whole sequence
            d_out_g = torch.empty([k, B, T, H, W, Fh], dtype=dt, device=dev)
            for s in range(k):
                ops.copy_slice(d_out_g[s], d_out_c[..., Fh * s:Fh * s + Fh])
        else:
            d_out_g = torch.zeros([k, B, T, H, W, Fh], dtype=dt, device=dev)
        # the gradient of the joint input [x | h[16:]] of every step, written in place by the data-gradient launches:
        # its x channels are the node's input gradient (a view, no copy per step)
        d_xw = torch.empty([T, B, H, W, cs], dtype=dt, device=dev)
        if t_run < T:
            d_xw[t_run:] = 0
        dy = torch.empty([B, H, W, Fh, 4], dtype=dt, device=dev)
        # (one per further group: each holds its group's gradient of the shared recurrent input until the sum below)
        dh_extras = [None] + [torch.zeros([B, H, W, Fh], dtype=dt, device=dev) for _ in range(1, k)]
        gw = [(torch.zeros([3, 3, C + F - Fh, 4 * Fh], dtype=torch.float32, device=dev),
               torch.zeros([3, 3, Fh, 4 * Fh], dtype=torch.float32, device=dev),
               torch.zeros([4 * Fh], dtype=torch.float32, device=dev)) for _ in range(k)]
        slabs = [ops.wgrad_workspace(cs, Fh, dev, cached=False) for _ in range(k)]
        for t in reversed(range(t_run)):
            live = (durations_dev > t).view(-1, 1, 1, 1) if durations_dev is not None else None
            mode = ops.WGRAD_DEFER if t < t_run - 1 else ops.WGRAD_DEFER_FIRST
            dh0_sum = None
            dxw = d_xw[t]
            for s in range(k):
                dh_extra = dh_extras[s]
                if fused:
                    # the recurrent input of every group is group 0's h: group 0's gradient of it becomes the carried
                    # one (a finished sample keeps its carried value), the other groups' are added to it below
                    if d_out_in_place:
                        above = dict(dh_above=d_out_c[:, t, ..., Fh * s:Fh * s + Fh],
                                     dh_above_batch_stride=d_out_c.stride(0), dh_above_pixel_stride=F)
                    else:
                        above = dict(dh_above=d_out_g[s][:, t], dh_above_batch_stride=T * H * W * Fh)
                    ops.bwd_fused(Gs[s][t], Cs[s][t], Hs[s][t + 1], dh_state=dh_state[s], dc_state=dc_state[s], x=xw[t],
                                  h_prev=Hs[0][t], packed=groups[s][1], dc_prev=dc_next[s],
                                  dh_prev=dh_next0 if s == 0 else dh_extra, workspace=slabs[s], mode=mode, dw_x=None,
                                  dx=dxw, dx_channels=cs, dx_accumulate=s > 0, durations=durations_dev, t=t, **above)
                    dc_state[s], dc_next[s] = dc_next[s], dc_state[s]
                    if s > 0:
                        # (for a finished sample the kernel passes the group's OWN carried gradient through)
                        extra = dh_extra if live is None else torch.where(live, dh_extra, torch.zeros_like(dh_extra))
                        dh0_sum = extra if dh0_sum is None else dh0_sum + extra
                    continue
                ops.pointwise_bwd(Gs[s][t], Cs[s][t], Cs[s][t + 1], d_out_g[s][:, t], T * H * W * Fh, dh_state[s], dc_state[s],
                                  durations_dev, t, dy=dy, dc_prev=dc_state[s])
                ops.wgrad(xw[t], Hs[0][t], dy, C + F - Fh, gw[s][0], gw[s][1], gw[s][2], True, durations_dev, t, mode=mode,
                          workspace=slabs[s])
                # every group's recurrent input is group 0's h: its gradient is the sum over the groups.  Group 0
                # writes into its own carried buffer (a finished sample keeps the carried value there), the others
                # into a buffer that is zero for finished samples.
                if s > 0 and live is not None:
                    dh_extra.zero_()
                ops.dgrad(dy, groups[s][1], dh_state[0] if s == 0 else dh_extra, dx=dxw, dx_channels=cs,
                          dx_accumulate=s > 0, durations=durations_dev, t=t)
                if s > 0:
                    dh0_sum = dh_extra if dh0_sum is None else dh0_sum + dh_extra
            if fused:
                dh_state[0], dh_next0 = dh_next0, dh_state[0]
            if t == 0 and not need_state:
                break
            if dh0_sum is not None:
                dh_state[0] += dh0_sum
            for s in range(1, k):  # h_prev[16 s:] entered through the joint input
                g = dxw[..., C + Fh * (s - 1):C + Fh * s]
                if live is not None:
                    dh_state[s] = torch.where(live, g, dh_state[s]).contiguous()
                elif fused and not os.environ.get("ISG_NO_STRIDED_BWD"):
                    dh_state[s] = g  # (the fused step reads it in place, as a slice of the wider pixels)
                else:
                    dh_state[s] = ops.copy_slice(torch.empty([B, H, W, Fh], dtype=dt, device=dev), g)
        dw_x = torch.zeros_like(cell.input_to_state_kernel, dtype=torch.float32)
        dw_h = torch.zeros_like(cell.hidden_to_state_kernel, dtype=torch.float32)
        dbias = torch.zeros_like(cell.bias, dtype=torch.float32)
        for s in range(k):
            if t_run > 0:
                ops.wgrad_finish(slabs[s], (B, H, W, cs), C + F - Fh, 0, *gw[s])
            cols = groups[s][0]
            dw_x[:, :, :, cols] = gw[s][0][:, :, :C]
            dw_h[:, :, Fh:, cols] = gw[s][0][:, :, C:]
            dw_h[:, :, :Fh, cols] = gw[s][1]
            dbias[cols] = gw[s][2]
        xd, cd, hd = ctx.in_dtypes
        ctx.saved = ctx.cell = None
        if getattr(ctx, "x_in_slots", False) and not os.environ.get("ISG_NO_STRIDED_BWD"):
            # the input was produced in place, as the first channels of the joint input's wider pixels: its gradient
            # goes back the same way, as that slice of the joint gradient (the producer's backward reads it there)
            d_x = d_xw[..., :C]
        elif getattr(ctx, "batch_major", False):
            d_x = ops.copy_slice(torch.empty([B, T, H, W, C], dtype=dt, device=dev).transpose(0, 1), d_xw[..., :C])
        else:
            d_x = ops.copy_slice(torch.empty([T, B, H, W, C], dtype=dt, device=dev), d_xw[..., :C])
        return (None, None, None, d_x.to(xd), torch.cat(dc_state, dim=-1).to(cd) if need_state else None,
                torch.cat(dh_state, dim=-1).to(hd) if need_state else None, dw_x, dw_h, dbias, None)


def _dynamic_rnn_wide(multi, seq, durations_dev, t_run, initial_state):
    """Cells wider than 16 filters: one cell, time-major tensor input, the whole sequence as one autograd node
    (``_WideSequenceFunction``).  Same masking rule: past a sample's length the output is zero and the state is
    carried through."""
    if seq.x_seq is None or len(multi.cells) != 1:
        raise NotImplementedError("wide cells: a single cell over a time-major tensor input")
    cell = multi.cells[0]
    c0, h0 = initial_state[0]
    args = (cell, durations_dev, t_run, seq.x_seq, c0, h0, cell.input_to_state_kernel, cell.hidden_to_state_kernel,
            cell.bias, getattr(seq, "recorded", None))
    if torch.is_grad_enabled() and any(torch.is_tensor(a) and a.requires_grad for a in args):
        out, c, h = _WideSequenceFunction.apply(*args)
    else:
        class _Ctx:  # the forward alone; nothing is kept
            forward_only = True

            def set_materialize_grads(self, _):
                pass
        out, c, h = _WideSequenceFunction.forward(_Ctx(), *[a.detach() if torch.is_tensor(a) else a for a in args])
    return out, (LSTMStateTuple(c, h),)


def dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, time_major=True, output_pool=1, steps=None):
    """``tf.nn.dynamic_rnn`` for a ``MultiRNNCell`` (or a single cell) of ConvLSTM cells.

    inputs: a time-major tensor [T, B, H, W, C] (``time_major=False``: [B, T, H, W, C]) or a
    ``SequenceInput``.  sequence_length: int array/tensor [B] or None.  Past a sample's length the
    emitted output is zero and its state is carried through unchanged.
    Returns (outputs, final_state); outputs are [T, B, H, W, F] when time_major else [B, T, H, W, F].
    ``output_pool`` = k > 1 (not in TensorFlow): the outputs come back averaged over k x k windows,
    [.., H/k, W/k, F], and the pool's gradient is applied inside the recurrence's backward sweep.
    ``steps`` (not in TensorFlow): an upper bound of the lengths known to the caller, for lengths given as a
    device tensor - that many steps run instead of fetching the longest length from the device.
    The whole recurrence is one autograd node that releases what it saved once its backward has run: a second
    backward through the same forward pass (``retain_graph=True``) raises.
    """
    multi = cell if isinstance(cell, MultiRNNCell) else MultiRNNCell([cell])
    if isinstance(inputs, SequenceInput):
        seq = inputs
        diff_input = seq.x_seq if seq.x_seq is not None else seq.static
    else:
        x_seq = inputs if time_major else inputs.transpose(0, 1)
        seq = SequenceInput(x_seq=x_seq)
        diff_input = x_seq
    dev = diff_input.device
    multi.build(seq.C, dev)
    T, B = seq.T, seq.B
    if sequence_length is None:
        durations_dev, t_run = None, T
    else:
        if torch.is_tensor(sequence_length) and steps is not None:
            # lengths that live on the device stay there: `steps` steps run and the kernels mask per sample
            # (asking the device for the longest length stalls the host until the stream has drained)
            t_run = int(steps) if sequence_length.numel() else 0
            durations_dev = sequence_length.to(device=dev, dtype=torch.int32).contiguous()
        elif torch.is_tensor(sequence_length):
            host = sequence_length.detach().cpu()
            t_run = int(host.max()) if host.numel() else 0
            durations_dev = sequence_length.to(device=dev, dtype=torch.int32).contiguous()
        else:
            import numpy as np
            host = np.asarray(sequence_length)
            t_run = int(host.max()) if host.size else 0
            durations_dev = ops.upload(host.astype(np.int32), dev)
        t_run = max(0, min(t_run, T))
    wide = any(c._filters != ops.F_HIDDEN for c in multi.cells)
    if initial_state is None:
        # the fused node takes "no tensor" as the zero state
        initial_state = multi.zero_state(B, dev) if wide else tuple((None, None) for _ in multi.cells)
    elif not isinstance(initial_state[0], (tuple, list)):
        initial_state = (initial_state,)
    if wide:
        if output_pool != 1:
            raise NotImplementedError("output_pool is for 16-filter cells")
        out, final = _dynamic_rnn_wide(multi, seq, durations_dev, t_run, initial_state)
        if time_major:
            out = out.transpose(0, 1)
        return out, (final if isinstance(cell, MultiRNNCell) else final[0])
    tensors = [diff_input]
    for c, h in initial_state:
        tensors += [c, h]
    for c in multi.cells:
        tensors += [c.input_to_state_kernel, c.hidden_to_state_kernel, c.bias]
    if not torch.is_grad_enabled() or not any(t is not None and t.requires_grad for t in tensors):
        # nothing will run backward (validation, decode): no gate tensors, two alternating state buffers per
        # layer instead of T saved ones; the final state lives in buffers private to this call
        out, fin, _ = _run_sequence_forward(list(multi.cells), seq, durations_dev, t_run, list(initial_state), False,
                                            int(output_pool))
        flat = [out] + [s for cf, hf in fin for s in (cf, hf)]
    else:
        flat = _SequenceFunction.apply(multi, seq, durations_dev, t_run, int(output_pool), *tensors)
    out = flat[0]
    final = tuple(LSTMStateTuple(flat[1 + 2 * l], flat[2 + 2 * l]) for l in range(len(multi.cells)))
    if time_major:
        out = out.transpose(0, 1)
    if not isinstance(cell, MultiRNNCell):
        final = final[0]
    return out, final
