"""The actor-critic of the reference's polyrl/a3c directory, run synchronously over batched environments.

The reference code is a specification that never ran (its policy passes ``height=``/``width=`` to a
cell that takes ``shape=``, polyrl/a3c/model.py:23, and its worker imports a missing ``envs`` module).
What it specifies is kept:

* policy (polyrl/a3c/model.py:13-44): 4 x conv(32, 3x3, same, relu) -> ConvLSTM(32 filters, 3x3) over the
  rollout as the time axis -> ``logits = dense(ac_space)``, ``vf = dense(1)`` on the flattened output;
  sampling = multinomial over the logits (:8-10);
* returns and advantages (polyrl/a3c/a3c.py:11-32): discounted returns, GAE with lambda = 1;
* loss (polyrl/a3c/a3c.py:201-216): ``-sum(log pi(a) adv) + 0.5 * 0.5 * sum((vf - R)^2) - 0.01 * H``;
* update (polyrl/a3c/a3c.py:248-258): clip_by_global_norm(40), Adam(1e-4);
* the runner's episode rule (polyrl/a3c/a3c.py:150-159): a rollout ends at the terminal step or after 10
  steps, whichever comes first, so it always ends an episode and its bootstrap value is 0.

What is not reproduced: asynchronous workers around a parameter server.  Here every rank steps
``num_envs`` environments in lock-step (the rollouts are the batch axis, their steps the time axis),
the per-rollout losses are averaged, and ranks exchange ONE all-reduce of the flat gradient.
The env adapter ``action -> (a // P, a % P)`` (x-major, as the class index of polyrnn/Model.py:39) is
this build's; the reference hands ``action.argmax()`` to an environment that expects a point.
"""
import contextlib
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops, parallel
from ..convLSTM import (ConvLSTMCell, LSTMStateTuple, SequenceInput, WideRecorder, WideStepper, dynamic_rnn,
                        glorot_uniform_)
from ..optim import FlatAdam
from ..polyrnn import decode
from ..polyrnn.Model import (_DenseHead, _EncoderChain, _RecordedEncoder, _head_weights, conv2d_same_relu,
                             encoder_chain_supported)


def discount(x, gamma):
    """y[t] = x[t] + gamma * y[t+1] along axis 0 (reference polyrl/a3c/a3c.py:11-12, which spells it
    as scipy.signal.lfilter([1], [1, -gamma]) on the reversed sequence)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    acc = np.zeros_like(x[0]) if x.ndim > 1 else 0.0
    for t in range(len(x) - 1, -1, -1):
        acc = x[t] + gamma * acc
        y[t] = acc
    return y


def process_rollout(rewards, values, bootstrap, gamma, lambda_=1.0):
    """(advantages, returns) of one rollout (reference polyrl/a3c/a3c.py:15-32)."""
    rewards = np.asarray(rewards, dtype=np.float64)
    vpred_t = np.asarray(list(values) + [bootstrap], dtype=np.float64)
    rewards_plus_v = np.asarray(list(rewards) + [bootstrap], dtype=np.float64)
    batch_r = discount(rewards_plus_v, gamma)[:-1]
    td_error = rewards + gamma * vpred_t[1:] - vpred_t[:-1]
    return discount(td_error, gamma * lambda_), batch_r


def process_rollouts(rewards, values, lengths, gamma, lambda_=1.0, bootstrap=None):
    """The same for N rollouts held as device tensors [N, T] with per-rollout lengths: entries at
    t >= length are ignored on input and zero on output.  float64 throughout."""
    N, T = rewards.shape
    dev = rewards.device
    if (rewards.is_cuda and rewards.dtype == torch.float64 and values.dtype == torch.float32 and
            lengths.dtype == torch.int32 and (bootstrap is None or bootstrap.dtype == torch.float64)):
        # one launch (isg_gae): the same sums in the same order, one thread per rollout
        from .. import _lib
        adv = torch.empty([N, T], dtype=torch.float64, device=dev)
        ret = torch.empty([N, T], dtype=torch.float64, device=dev)
        rc = _lib.lib().isg_gae(_lib.ptr(rewards.contiguous()), _lib.ptr(values.contiguous()),
                                _lib.ptr(lengths.contiguous()), _lib.ptr(bootstrap.contiguous()) if bootstrap is not None
                                else None, N, T, float(gamma), float(lambda_), _lib.ptr(adv), _lib.ptr(ret),
                                _lib.current_stream())
        _lib.check(rc, "isg_gae")
        return adv, ret
    live = torch.arange(T, device=dev)[None, :] < lengths[:, None]
    rewards = torch.where(live, rewards.double(), torch.zeros([], dtype=torch.float64, device=dev))
    values = torch.where(live, values.double(), torch.zeros([], dtype=torch.float64, device=dev))
    boot = torch.zeros([N], dtype=torch.float64, device=dev) if bootstrap is None else bootstrap.double()
    last = (lengths.long() - 1).clamp(min=0)
    # v[t+1], with the bootstrap value after the last live step
    v_next = torch.cat([values[:, 1:], torch.zeros([N, 1], dtype=torch.float64, device=dev)], dim=1)
    v_next[torch.arange(N, device=dev), last] = boot
    v_next = torch.where(live, v_next, torch.zeros_like(v_next))
    td = rewards + gamma * v_next - values
    r_in = rewards.clone()
    r_in[torch.arange(N, device=dev), last] += gamma * boot * (lengths > 0)
    # y[t] = x[t] + g y[t+1] is y = x @ M with M[s, t] = g^(s - t) for s >= t: the entries past a rollout's length
    # are zero on input (they sit at the end, where the backward scan starts from zero), so one small float64
    # product per quantity replaces T dependent steps of three launches each
    adv = td @ _discount_matrix(T, gamma * lambda_, dev)
    ret = r_in @ _discount_matrix(T, gamma, dev)
    zero = torch.zeros_like(adv)
    return torch.where(live, adv, zero), torch.where(live, ret, zero)


_discount_matrices = {}


def _discount_matrix(T, g, device):
    """float64 [T, T]: M[s, t] = g^(s - t) for s >= t, else 0 (cached per length, factor and device)."""
    key = (int(T), float(g), str(device))
    m = _discount_matrices.get(key)
    if m is None:
        d = np.arange(T)[:, None] - np.arange(T)[None, :]
        m = torch.tensor(np.where(d >= 0, float(g) ** np.maximum(d, 0), 0.0), dtype=torch.float64, device=device)
        _discount_matrices[key] = m
    return m


class _A3CLoss(torch.autograd.Function):
    """The three sums of the loss below in three launches (isg_a3c_loss_fwd + isg_column_sums) and their gradient in
    one (isg_a3c_loss_bwd), where the tensor form takes some forty small launches over the [R, K] logits: returns
    float32 [3] = (policy loss, value loss, entropy), sums over the rows."""

    @staticmethod
    def forward(ctx, logits, vf, actions, adv, r, live):
        from .. import _lib
        R, K = logits.shape
        z = logits.contiguous()
        dev = z.device
        vf32, adv32, r32 = vf.float().contiguous(), adv.float().contiguous(), r.float().contiguous()
        a32 = actions.to(torch.int32).contiguous()
        live8 = None if live is None else live.to(torch.uint8).contiguous()
        terms = torch.empty([R, 3], dtype=torch.float32, device=dev)
        stats = torch.empty([R, 2], dtype=torch.float32, device=dev)
        rc = _lib.lib().isg_a3c_loss_fwd(_lib.ptr(z), _lib.ptr(vf32), _lib.ptr(a32), _lib.ptr(adv32), _lib.ptr(r32),
                                         _lib.ptr(live8), R, K, ops.dtype_code(z.dtype), _lib.ptr(terms), _lib.ptr(stats),
                                         _lib.current_stream())
        _lib.check(rc, "isg_a3c_loss_fwd")
        ctx.save_for_backward(z, vf32, a32, adv32, r32, stats)
        ctx.live8, ctx.vf_dtype = live8, vf.dtype
        return decode.column_sums(terms)

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        z, vf32, a32, adv32, r32, stats = ctx.saved_tensors
        R, K = z.shape
        dlogits = torch.empty_like(z)
        dvf = torch.empty([R], dtype=torch.float32, device=z.device)
        rc = _lib.lib().isg_a3c_loss_bwd(_lib.ptr(z), _lib.ptr(vf32), _lib.ptr(a32), _lib.ptr(adv32), _lib.ptr(r32),
                                         _lib.ptr(ctx.live8), _lib.ptr(stats), _lib.ptr(g.float().contiguous()), R, K,
                                         ops.dtype_code(z.dtype), _lib.ptr(dlogits), _lib.ptr(dvf), _lib.current_stream())
        _lib.check(rc, "isg_a3c_loss_bwd")
        return dlogits, dvf.to(ctx.vf_dtype), None, None, None, None


def a3c_loss(logits, vf, actions, adv, r, live=None):
    """Reference polyrl/a3c/a3c.py:201-216 over rows [R, K] (R = all steps of all rollouts); rows with
    ``live`` False contribute nothing.  Returns (loss, dict of the three parts), all sums."""
    if logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16) and not os.environ.get("ISG_NO_FUSED_A3C_LOSS"):
        sums = _A3CLoss.apply(logits, vf, actions, adv, r, live)
        pi_loss, vf_loss, entropy = sums[0], sums[1], sums[2]
        return pi_loss + 0.5 * vf_loss - entropy * 0.01, dict(pi_loss=pi_loss, vf_loss=vf_loss, entropy=entropy)
    logp = F.log_softmax(logits.float(), dim=-1)
    p = logp.exp()
    w = torch.ones_like(vf, dtype=torch.float32) if live is None else live.to(torch.float32)
    chosen = logp.gather(1, actions.long()[:, None])[:, 0]
    pi_loss = -(chosen * adv.float() * w).sum()
    vf_loss = 0.5 * (((vf.float() - r.float()) ** 2) * w).sum()
    entropy = -((p * logp).sum(dim=1) * w).sum()
    return pi_loss + 0.5 * vf_loss - entropy * 0.01, dict(pi_loss=pi_loss, vf_loss=vf_loss, entropy=entropy)


def _dense_init(shape, device, gen):
    limit = math.sqrt(6.0 / (shape[0] + shape[1]))
    return (torch.rand(shape, device=device, generator=gen) * 2 - 1) * limit


def _dense_rows(flat, w, b, slices=16, b32=None):
    """flat [M, K] @ w [K, N] + b as float32, operands in the compute type.  An acting step has few rows and a long
    reduction axis (1024 x 32768 x 1024: 32 output tiles of a library GEMM for 256 CUs): the reduction axis is cut
    into ``slices`` batched GEMMs with float32 results that are then summed (0.18 -> 0.08 ms, and the partial sums
    are never rounded to bf16; with ``b32``, the float32 bias, sum and bias are one launch).  One output column (the
    value head) is a row-wise dot product instead (isg_rows_dot)."""
    M, K = flat.shape
    N = w.shape[1]
    if (not os.environ.get("ISG_NO_SPLIT_HEAD") and flat.is_cuda and flat.dtype == torch.bfloat16 and flat.is_contiguous() and w.is_contiguous() and N >= 64 and
            M * N <= (1 << 21) and K % (slices * 8) == 0 and K // slices >= 512):
        try:
            part = torch.bmm(flat.view(M, slices, K // slices).transpose(0, 1), w.view(slices, K // slices, N),
                             out_dtype=torch.float32)
            if b32 is not None and b32.dtype == torch.float32:  # (the float32 master bias: one launch)
                return decode.sum_slices_bias(part, b32)
            return part.sum(0).add_(b)
        except TypeError:  # (a torch without float32 results from a bf16 batched GEMM)
            pass
    if N == 1 and not os.environ.get("ISG_NO_SPLIT_HEAD"):  # one column: a row-wise dot product, not a GEMM
        v = decode.rows_dot(flat, w, b)
        if v is not None:
            return v[:, None]
    return torch.addmm(b, flat, w).float()


class _JointHeads(torch.autograd.Function):
    """The action and the value head (reference polyrl/a3c/model.py:38-41) as one autograd node.  Forward: the two
    library GEMMs of ``_DenseHead``.  Backward: ONE layer of A + 1 columns (padded to a multiple of 8) - the input
    gradient as one GEMM over the joint weights instead of a GEMM, a rank-one product and the sum of the two
    [rows, K] results, the weight gradients as one GEMM, the bias gradients as one column sum (at 10,240 rows,
    K = 32,768, A = 1,024: 0.75 ms less; the forward stays apart because 1,032 columns cost the forward GEMM 0.36 ms
    more than 1,024 do)."""

    @staticmethod
    def forward(ctx, flat, wa, ba, wv, bv, caches):
        wca, bca = _head_weights(flat.dtype, wa, ba, caches[0])
        wcv, bcv = _head_weights(flat.dtype, wv, bv, caches[1])
        ctx.save_for_backward(flat, wca, wcv)
        ctx.params = (wa.detach(), ba.detach(), wv.detach(), bv.detach())
        return torch.addmm(bca, flat, wca), torch.addmm(bcv, flat, wcv)

    @staticmethod
    def backward(ctx, dlogits, dvf):
        flat, wca, wcv = ctx.saved_tensors
        M, A = dlogits.shape
        K = flat.shape[1]
        pad = (-(A + 1)) % 8
        d = torch.empty([M, A + 1 + pad], dtype=flat.dtype, device=flat.device)
        d[:, :A] = dlogits
        d[:, A:A + 1] = dvf
        parts = [wca, wcv]
        if pad:
            d[:, A + 1:] = 0
            parts.append(wca.new_zeros([K, pad]))
        wj = torch.cat(parts, dim=1)
        d_flat = d @ wj.t() if ctx.needs_input_grad[0] else None
        dwj = flat.t() @ d
        dbj = decode.column_sums(d)
        wa, ba, wv, bv = ctx.params
        out = []
        for prm, g in ((wa, dwj[:, :A]), (ba, dbj[:A]), (wv, dwj[:, A:A + 1]), (bv, dbj[A:A + 1])):
            dst = ops.grad_destination(prm)
            if dst is not None:
                dst.copy_(g)
                ops.grad_delivered(prm)
                out.append(None)
            else:
                out.append(g.to(prm.dtype))
        return (d_flat,) + tuple(out) + (None,)


class LSTMPolicy:
    """Reference polyrl/a3c/model.py:13-68 for N environments at once: ``x`` is [N, T, H, W, C]."""

    def __init__(self, ob_space, ac_space, filters=32, compute_dtype=torch.float32, device=None, seed=0):
        H, W, C = ob_space
        self.ob_space, self.ac_space, self.filters = tuple(ob_space), int(ac_space), int(filters)
        self.compute_dtype = compute_dtype
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        self.convs = []
        cin = C
        for _ in range(4):
            w = torch.empty([3, 3, cin, 32], device=self.device)
            limit = math.sqrt(6.0 / (9 * cin + 9 * 32))
            w.copy_((torch.rand(w.shape, device=self.device, generator=gen) * 2 - 1) * limit)
            self.convs.append((torch.nn.Parameter(w), torch.nn.Parameter(torch.zeros([32], device=self.device))))
            cin = 32
        # (from the policy's own seeded stream: every rank of a data-parallel run must start from the same weights)
        init = lambda t: t.copy_(glorot_uniform_(torch.empty_like(t), generator=gen))
        self.lstm = ConvLSTMCell(shape=[H, W], filters=self.filters, kernel=[3, 3], initializer=init,
                                 compute_dtype=compute_dtype, device=self.device)
        self.lstm.build(cin, self.device)
        self.state_size = self.lstm.state_size
        flat = H * W * self.filters
        self.action_kernel = torch.nn.Parameter(_dense_init((flat, self.ac_space), self.device, gen))
        self.action_bias = torch.nn.Parameter(torch.zeros([self.ac_space], device=self.device))
        self.value_kernel = torch.nn.Parameter(_dense_init((flat, 1), self.device, gen))
        self.value_bias = torch.nn.Parameter(torch.zeros([1], device=self.device))
        self._head_caches = ({}, {})  # compute-type copies of the two dense heads' weights
        self._sample_calls = 0
        self._base_seed = seed  # weights above come from `seed` itself: equal on every rank
        self._rank_seed = None

    @property
    def _seed(self):
        """The rank's own sampling seed, fixed at the FIRST sampled action: the process group may be initialised
        after the policy is built, and a seed folded before that would be the same on every rank."""
        if self._rank_seed is None:
            import os
            if int(os.environ.get("WORLD_SIZE", "1")) > 1 and parallel.world_size() == 1:
                raise RuntimeError("WORLD_SIZE > 1 but no process group is initialised: every rank would sample the "
                                   "same actions; call torch.distributed.init_process_group first")
            self._rank_seed = parallel.sampling_seed(self._base_seed)
        return self._rank_seed

    @property
    def var_list(self):
        out = []
        for w, b in self.convs:
            out += [w, b]
        out += [self.lstm.hidden_to_state_kernel, self.lstm.input_to_state_kernel, self.lstm.bias,
                self.action_kernel, self.action_bias, self.value_kernel, self.value_bias]
        return out

    def named_variables(self, scope="global"):
        names = []
        for i in range(4):
            names += ["%s/l%d/kernel" % (scope, i + 1), "%s/l%d/bias" % (scope, i + 1)]
        base = "%s/rnn/ConvLSTMCell/" % scope
        names += [base + "hidden_to_state_kernel", base + "input_to_state_kernel", base + "bias",
                  "%s/action/kernel" % scope, "%s/action/bias" % scope, "%s/value/kernel" % scope,
                  "%s/value/bias" % scope]
        return list(zip(names, self.var_list))

    def get_initial_features(self, num_envs=1):
        """Zero LSTM state (c float32, h compute dtype), reference polyrl/a3c/model.py:27-29,46-48."""
        return self.lstm.zero_state(num_envs, self.device)

    def encoder_keeps(self, x):
        """Whether ``_encode(x, keep=)`` can keep its layers' outputs for ``forward(recorded_encoder=)``: the padded
        first layer and the square kernels, bf16 on the GPU."""
        dt = self.compute_dtype
        return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and dt == torch.bfloat16 and
                ops.conv32_padded_supported(self.convs[0][0], dt, x.device) and
                all(ops.conv32_supported(w, dt, x.device) for w, _ in self.convs[1:]))

    def _encode(self, x, keep=None):
        # [R, H, W, C] -> [R, H, W, 32]
        # keep = (padded-input buffer, [one output buffer per layer]): an acting step writes its layers there, for
        # the update that follows under the same weights (``forward(recorded_encoder=)``)
        if keep is not None:
            y = ops.pad_channels(x, self.convs[0][0].shape[3], out=keep[0])
            for k, (w, b) in enumerate(self.convs):
                wimg = ops.conv32_pack_padded(w) if k == 0 else ops.conv32_pack(w)
                y = ops.conv32(y, wimg, b.detach().float().contiguous(), w.shape[0], True, out=keep[1][k])
            return y
        # the same layer routine as the polygon-RNN encoder: 32 -> 32 layers run on the conv32 kernels in bf16
        dt = self.compute_dtype
        w0 = self.convs[0][0]
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and ops.conv32_padded_supported(w0, dt, x.device):
            # conversion to the compute type and the zero channels of the padded first layer in one pass
            y = ops.pad_channels(x, w0.shape[3])
        else:
            y = x.to(dt).contiguous()
        if torch.is_grad_enabled() and encoder_chain_supported(y, self.convs, dt):
            return _EncoderChain.apply(y, *[p for wb in self.convs for p in wb]).contiguous()
        for w, b in self.convs:
            y = conv2d_same_relu(y, w, b, dt)
        return y.contiguous()

    def forward(self, x, state, recorded=None, recorded_encoder=None):
        """x [N, T, H, W, C], state (c, h) -> logits [N, T, K] float32, vf [N, T] float32, state_out.
        ``recorded``: the recurrence's pass over these very observations from the zero state with the present weights
        (``WideRecorder.record()`` of the rollout that produced them); the recurrence's node adopts it (state_out is
        then (None, None): the final state stays in the recorder).
        ``recorded_encoder``: likewise the encoder's pass, (padded observations, [the layers' outputs]), each
        time-major [T, N, H, W, 32] (``_encode(keep=)`` of the acting steps)."""
        N, T = x.shape[0], x.shape[1]
        if recorded_encoder is not None:
            x4, ys = recorded_encoder
            rows = lambda a: a.view((T * N,) + tuple(a.shape[2:]))
            feats_tm = _RecordedEncoder.apply((rows(x4), [rows(y) for y in ys]), *[p for wb in self.convs for p in wb])
            seq = SequenceInput(x_seq=feats_tm.view((T, N) + tuple(feats_tm.shape[1:])))
        else:
            feats = self._encode(x.reshape((N * T,) + tuple(x.shape[2:]))).reshape((N, T) + tuple(x.shape[2:4]) + (32,))
            seq = SequenceInput(x_seq=feats.transpose(0, 1))
        seq.recorded = recorded
        out, final = dynamic_rnn(self.lstm, seq, initial_state=LSTMStateTuple(*state), time_major=False)
        if T > 1 and N > 1 and out.transpose(0, 1).is_contiguous():
            # an adopted pass keeps its outputs time-major: the heads take the rows in that order (no copy into
            # batch-major order) and the results go out as [N, T] views of time-major memory
            logits, vf = self._heads(out.transpose(0, 1).reshape(T * N, -1))
            return logits.view(T, N, -1).transpose(0, 1), vf.view(T, N).transpose(0, 1), final
        logits, vf = self._heads(out.reshape(N * T, -1))
        return logits.reshape(N, T, -1), vf.reshape(N, T), final

    def _heads(self, flat):
        """flat [R, H W F] -> (logits float32 [R, K], value float32 [R]): two library GEMMs in the compute type on
        copies of the weights that are converted once per optimiser step (reference polyrl/a3c/model.py:38-41)."""
        dt = flat.dtype
        heads = ((self.action_kernel, self.action_bias), (self.value_kernel, self.value_bias))
        if (torch.is_grad_enabled() and flat.is_cuda and dt == torch.bfloat16 and flat.requires_grad and
                not os.environ.get("ISG_NO_JOINT_HEADS")):
            logits, vf = _JointHeads.apply(flat, *heads[0], *heads[1], self._head_caches)
            return logits.float(), vf.float()[:, 0]
        out = []
        for (w, b), cache in zip(heads, self._head_caches):
            if torch.is_grad_enabled() and flat.is_cuda and (flat.requires_grad or w.requires_grad):
                out.append(_DenseHead.apply(flat, w, b, cache).float())
            elif dt != w.dtype:
                wc, bc = _head_weights(dt, w, b, cache)
                out.append(_dense_rows(flat, wc, bc, b32=b.detach()))
            else:
                out.append(torch.addmm(b.detach(), flat, w.detach()))
        return out[0], out[1][:, 0]

    def act(self, ob, c, h):
        """One step for N environments: (action index int64 [N], value [N], c, h)
        (reference polyrl/a3c/model.py:50-63, whose action is the one-hot of the same index)."""
        with torch.no_grad():
            logits, vf, (c2, h2) = self.forward(ob[:, None], (c, h))
            self._sample_calls += 1
            a = decode.sample_rows(logits[:, 0].contiguous(), seed=self._seed * 1000003 + self._sample_calls)
        return a.long(), vf[:, 0], c2, h2

    def stepper(self, num_envs):
        """Recurrent state of ``num_envs`` environments for ``act_step``: kept on the device in the layout the cell
        kernels use, from the zero state (reference polyrl/a3c/model.py:27-29)."""
        return WideStepper(self.lstm, num_envs, self.device)

    def recorder(self, num_envs, steps):
        """``stepper`` that also keeps the recurrence's pass for the update (``WideRecorder``)."""
        return WideRecorder(self.lstm, num_envs, self.device, steps)

    def act_step(self, ob, stepper):
        """``act`` with the recurrent state held by ``stepper`` (advanced in place): (action index int64 [N],
        value [N]).  Same arithmetic and the same sampling stream as ``act``."""
        with torch.no_grad():
            feats = self._encode(ob)
            logits, vf = self._heads(stepper.step(feats).reshape(ob.shape[0], -1))
            self._sample_calls += 1
            a = decode.sample_rows(logits.contiguous(), seed=self._seed * 1000003 + self._sample_calls)
        return a.long(), vf

    def act_step_xy(self, ob, stepper, row_offset_dev, grid_side, keep=None):
        """``act_step`` for a loop that may be captured as a graph: the action arrives as int32 (x, y) [N, 2] from the
        sampling launch itself and the sampling stream's position is a device counter (int64 scalar, advanced here by
        N), so that a replay draws new numbers.  Returns ((x, y), value [N])."""
        with torch.no_grad():
            feats = self._encode(ob, keep=keep)  # (keep: see ``_encode``)
            logits, vf = self._heads(stepper.step(feats).reshape(ob.shape[0], -1))
            xy = decode.sample_actions(logits.contiguous(), seed=self._seed, grid_side=grid_side,
                                       row_offset_dev=row_offset_dev)
            row_offset_dev += ob.shape[0]
        return xy, vf

    def value(self, ob, c, h):
        with torch.no_grad():
            return self.forward(ob[:, None], (c, h))[1][:, 0]


class A3C:
    """Synchronous, sharded replacement of reference polyrl/a3c/a3c.py:168-302: ``process()`` runs one
    rollout per environment, then one update."""

    TIMESTEP_LIMIT = 10  # reference polyrl/a3c/a3c.py:150

    def __init__(self, env, policy=None, gamma=0.99, lambda_=1.0, learning_rate=1e-4, process_group=None,
                 compute_dtype=torch.float32, seed=0):
        self.env = env
        P = env.size
        if policy is None:
            policy = LSTMPolicy((P, P, 3 + env.hist), P * P, compute_dtype=compute_dtype, device=env.device, seed=seed)
        self.network = self.local_network = policy  # one copy: the sync of :251,281 is the identity
        self.gamma, self.lambda_ = gamma, lambda_
        self.opt = FlatAdam(policy.var_list, learning_rate=learning_rate, clip_norm=40.0, process_group=process_group)
        #: images and copies derived from the parameters, brought up to date by one launch per update
        #: (ops.DerivedWeights) instead of once per acting step
        self.derived = ops.DerivedWeights(self.opt.flat_param) if self.opt.flat_param.is_cuda else contextlib.nullcontext()
        self.global_step = 0
        self.local_steps = 0
        self.last = {}
        #: True: the acting loop of a 32-filter policy (T x (encoder, cell, heads, sampling, environment step,
        #: terminal IoU)) is captured as one HIP graph and replayed (the loop is a few hundred launches of a few
        #: microseconds each: the host cannot issue them as fast as the GPU retires them)
        self.rollout_graph = self.opt.flat_param.is_cuda and not os.environ.get("ISG_NO_ROLLOUT_GRAPH")
        self._rollout = None
        #: the update adopts the rollout's pass of the recurrence instead of running it again (the stepper path: a
        #: 32-filter cell on the GPU in bf16; ISG_NO_FORWARD_REUSE=1 for A/B runs)
        self.reuse_forward = (self.opt.flat_param.is_cuda and self.local_network.filters > 16 and
                              self.local_network.compute_dtype == torch.bfloat16 and
                              not os.environ.get("ISG_NO_FORWARD_REUSE"))
        #: likewise the encoder's pass (its weight gradients then sum the rows step by step instead of environment by
        #: environment: the same sums in another order of float additions)
        self.reuse_encoder = self.reuse_forward and not os.environ.get("ISG_NO_ENCODER_REUSE")

    def pull_batch(self, images=None, polygons=None, device_batch=None):
        """Every environment from reset to its terminal step or the step limit.  ``device_batch``: episodes that are
        on the device already (``DevicePolygonPool.sample``: no host work at reset), else ``images`` / ``polygons``
        from the host, else a draw from the environment's dataset."""
        env, pi = self.env, self.local_network
        N, T, P = env.num_envs, self.TIMESTEP_LIMIT, env.size
        dev = env.device
        state = env.reset(images=images, polygons=polygons, device_batch=device_batch)
        if pi.filters > 16 and dev.type == "cuda":
            return self._pull_batch_stepper(state)
        c, h = pi.get_initial_features(N)
        obs = torch.zeros((N, T) + tuple(state.shape[1:]), dtype=state.dtype, device=dev)
        actions = torch.zeros([N, T], dtype=torch.int64, device=dev)
        values = torch.zeros([N, T], dtype=torch.float32, device=dev)
        rewards = torch.zeros([N, T], dtype=torch.float64, device=dev)
        lengths = torch.zeros([N], dtype=torch.int32, device=dev)
        for t in range(T):
            # no "is anyone still running" question to the host: a finished environment ignores its action and
            # its later steps fall outside its length
            active = ~env.done
            obs[:, t] = state
            a, v, c, h = pi.act(state, c, h)
            actions[:, t], values[:, t] = a, v
            lengths += active.to(torch.int32)
            xy = torch.stack([a // P, a % P], dim=1)
            state, reward, _ = env.step(xy, host_sync=False)
            rewards[:, t] = reward
        return dict(si=obs, a=actions, values=values, rewards=rewards, lengths=lengths, terminal=env.done.clone())

    def _pull_batch_stepper(self, state):
        """The acting loop with the recurrent state on the device between steps (``WideStepper``) and nothing in it
        that needs the host: eager the first time, a graph replay from then on (``rollout_graph``).  The buffers it
        returns are reused by the next rollout."""
        env, pi = self.env, self.local_network
        N, T, P = env.num_envs, self.TIMESTEP_LIMIT, env.size
        dev = env.device
        r = self._rollout
        tensors = (env._state, env.vertices, env.num_vertices, env.predicted_vertices, env.num_predicted,
                   env.history_points, env.done)
        where = tuple(t.data_ptr() for t in tensors) + (N, T, P)
        if r is None or r["where"] != where:
            rec = pi.recorder(N, T) if self.reuse_forward else pi.stepper(N)
            r = self._rollout = dict(
                where=where, graph=None, rec=rec,
                enc=self._encoder_buffers(state, N, T, rec) if self.reuse_forward and self.reuse_encoder else None,
                offset=torch.zeros([], dtype=torch.int64, device=dev),
                obs=torch.zeros((N, T) + tuple(state.shape[1:]), dtype=state.dtype, device=dev),
                xy=torch.zeros([N, T, 2], dtype=torch.int32, device=dev),
                actions=torch.zeros([N, T], dtype=torch.int64, device=dev),
                values=torch.zeros([N, T], dtype=torch.float32, device=dev),
                rewards=torch.zeros([N, T], dtype=torch.float64, device=dev),
                lengths=torch.zeros([N], dtype=torch.int32, device=dev))

        def body():
            r["rec"].reset()
            r["lengths"].zero_()
            state = env._state
            fold = not os.environ.get("ISG_NO_ROLLOUT_RECORD")  # (the switch: the tensor-operation form, for A/B runs)
            for t in range(T):
                r["obs"][:, t] = state
                keep = None if r["enc"] is None else (r["enc"][0][t], [y[t] for y in r["enc"][1]])
                xy, v = pi.act_step_xy(state, r["rec"], r["offset"], P, keep=keep)
                if fold and v.dtype == torch.float32 and v.is_contiguous():
                    # action, length, reward and value of the step filed by the environment's own launches
                    state = env.step_recorded(xy, t, r["xy"], r["lengths"], r["rewards"], v, r["values"])
                    continue
                active = ~env.done
                r["xy"][:, t], r["values"][:, t] = xy, v
                r["lengths"] += active.to(torch.int32)
                state, reward, _ = env.step(xy, host_sync=False)
                r["rewards"][:, t] = reward
            torch.add(r["xy"][..., 0] * P, r["xy"][..., 1], out=r["actions"])  # x-major class index, as ``act``

        derived = self.derived if isinstance(self.derived, ops.DerivedWeights) else None
        g = r["graph"]
        if g is not None and derived is not None and g["generation"] != derived.generation:
            g = r["graph"] = None  # (a compile moved the weight images the graph reads)
        if g is not None:
            if derived is not None:
                derived.ensure_fresh()
            g["graph"].replay()
        elif self.rollout_graph and r.get("warm"):
            # (the first rollout ran eagerly: library handles, workspaces and the recorded weight images exist)
            generation = None
            if derived is not None:
                derived.compile()
                generation = derived.generation
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, capture_error_mode="relaxed"):
                    body()
                r["graph"] = dict(graph=graph, generation=generation)
                graph.replay()  # (a capture records, it does not run)
            except RuntimeError as err:
                import warnings
                warnings.warn("acting loop not captured as a graph (%s): launching it step by step" % err)
                self.rollout_graph = False
                body()
        else:
            body()
            r["warm"] = True
        return dict(si=r["obs"], a=r["actions"], values=r["values"], rewards=r["rewards"], lengths=r["lengths"],
                    terminal=env.done.clone(), forward=r["rec"].record() if self.reuse_forward else None,
                    encoder=r["enc"] if self.reuse_forward and self.reuse_encoder else None)

    def _encoder_buffers(self, state, N, T, rec=None):
        """Where the acting steps keep the encoder's padded input and layer outputs for the update, time-major
        [T, N, H, W, 32] each; None where the encoder does not run on the kernels that can keep them."""
        pi = self.local_network
        if not pi.encoder_keeps(state):
            return None
        shape = [T, N] + list(state.shape[1:3]) + [int(pi.convs[0][0].shape[3])]
        new = lambda: torch.empty(shape, dtype=torch.bfloat16, device=state.device)
        ys = [new() for _ in pi.convs]
        if rec is not None and not os.environ.get("ISG_NO_INPUT_SLOTS"):
            # the top layer writes its features where the recurrent cell reads them: no copy per acting step
            slots = rec.input_slots()
            if slots.dtype == torch.bfloat16 and list(slots.shape) == shape and shape[-1] == 32:
                ys[-1] = slots
        return new(), ys

    def process(self, images=None, polygons=None, device_batch=None):
        with self.derived:
            batch = self._process(images, polygons, device_batch)
        if isinstance(self.derived, ops.DerivedWeights):
            self.derived.compile()
        return batch

    def _process(self, images=None, polygons=None, device_batch=None):
        pi = self.local_network
        batch = self.pull_batch(images, polygons, device_batch)
        N, T = batch["a"].shape
        adv, r = process_rollouts(batch["rewards"], batch["values"], batch["lengths"], self.gamma, self.lambda_)
        live = torch.arange(T, device=adv.device)[None, :] < batch["lengths"][:, None]
        self.opt.zero_grad()
        # (the weights have not changed since the rollout: the update adopts its pass of the recurrence)
        logits, vf, _ = pi.forward(batch["si"], pi.get_initial_features(N), recorded=batch.get("forward"),
                                   recorded_encoder=batch.get("encoder"))
        # (the loss is a sum over rows: they go in the order the logits are stored in, time-major after an adopted pass)
        if T > 1 and N > 1 and logits.transpose(0, 1).is_contiguous():
            rows = lambda a: a.transpose(0, 1).reshape((N * T,) + tuple(a.shape[2:]))
        else:
            rows = lambda a: a.reshape((N * T,) + tuple(a.shape[2:]))
        loss, parts = a3c_loss(rows(logits), rows(vf), rows(batch["a"]), rows(adv), rows(r), rows(live))
        (loss / N).backward()
        norm = self.opt.step()
        # one device-to-host copy for all the scalars of the update
        total = batch["lengths"].sum().double()
        stats = torch.stack([total, loss.detach().double(), norm.detach().double().reshape(()),
                             parts["pi_loss"].detach().double(), parts["vf_loss"].detach().double(),
                             parts["entropy"].detach().double(), batch["rewards"].sum(dim=1).mean().double()]).tolist()
        self.global_step += int(stats[0])
        self.local_steps += 1
        bs = max(stats[0], 1.0)
        self.last = dict(loss=stats[1] / N, grad_global_norm=stats[2], policy_loss=stats[3] / bs,
                         value_loss=stats[4] / bs, entropy=stats[5] / bs, episode_reward=stats[6])
        self.last_env_steps = int(stats[0])
        return batch
