"""Softmax cross-entropy, argmax and sampling over the vertex class grid, on the libisg kernels
(isg_softmax_xent_fwd / _bwd, isg_argmax_rows, isg_sample_rows)."""
import torch

from .. import _lib
from ..ops import dtype_code


def _rows(logits):
    cols = logits.shape[-1]
    z = logits.reshape(-1, cols)
    if z.dtype not in (torch.float32, torch.bfloat16):
        z = z.float()
    return z.contiguous(), z.shape[0], cols


class _XentRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weights, argmax_out=None):
        z, rows, cols = _rows(logits)
        lab = labels.reshape(-1).to(torch.int32).contiguous()
        w = weights.reshape(-1).to(torch.float32).contiguous() if weights is not None else None
        loss = torch.empty([rows], dtype=torch.float32, device=z.device)
        lse = torch.empty([rows], dtype=torch.float32, device=z.device)
        rc = _lib.lib().isg_softmax_xent_fwd(_lib.ptr(z), _lib.ptr(lab), _lib.ptr(w), rows, cols, dtype_code(z.dtype),
                                             _lib.ptr(loss), _lib.ptr(lse), _lib.ptr(argmax_out),
                                             _lib.current_stream())
        _lib.check(rc, "isg_softmax_xent_fwd")
        ctx.save_for_backward(z, lab, lse, w)
        ctx.shape, ctx.in_dtype = logits.shape, logits.dtype
        return loss.reshape(logits.shape[:-1])

    @staticmethod
    def backward(ctx, dloss):
        z, lab, lse, w = ctx.saved_tensors
        rows, cols = z.shape
        scale = dloss.reshape(-1).to(torch.float32)
        if w is not None:
            scale = scale * w
        scale = scale.contiguous()
        dz = torch.empty_like(z)
        rc = _lib.lib().isg_softmax_xent_bwd(_lib.ptr(z), _lib.ptr(lab), _lib.ptr(lse), _lib.ptr(scale), 1.0, rows,
                                             cols, dtype_code(z.dtype), _lib.ptr(dz), _lib.current_stream())
        _lib.check(rc, "isg_softmax_xent_bwd")
        return dz.reshape(ctx.shape).to(ctx.in_dtype), None, None, None


class _SequenceLoss(torch.autograd.Function):
    """Mean cross-entropy over every step of every sequence together with the step's masked metrics (reference
    polyrnn/Model.py:44-54, 98-100) in two launches - the per-row pass and one summary launch that finishes the
    mean and the quotients on the device - and a backward pass of one launch."""

    @staticmethod
    def forward(ctx, logits, targets, durations, prediction_size):
        z, rows, cols = _rows(logits)
        B, T = targets.shape[0], targets.shape[1]
        tg = targets.to(torch.int32).contiguous()
        dur = durations.to(torch.int32).contiguous()
        dev = z.device
        loss_rows = torch.empty([rows], dtype=torch.float32, device=dev)
        lse = torch.empty([rows], dtype=torch.float32, device=dev)
        pred = torch.empty([rows], dtype=torch.int32, device=dev)
        lab = torch.empty([rows], dtype=torch.int32, device=dev)
        rc = _lib.lib().isg_sequence_xent_fwd(_lib.ptr(z), _lib.ptr(tg), rows, cols, int(prediction_size),
                                              dtype_code(z.dtype), _lib.ptr(loss_rows), _lib.ptr(lse), _lib.ptr(pred),
                                              _lib.ptr(lab), _lib.current_stream())
        _lib.check(rc, "isg_sequence_xent_fwd")
        out6 = torch.empty([6], dtype=torch.float64, device=dev)
        mean = torch.empty([], dtype=torch.float32, device=dev)
        rc = _lib.lib().isg_sequence_summary(_lib.ptr(pred), _lib.ptr(tg), _lib.ptr(dur), _lib.ptr(loss_rows), B, T,
                                             int(prediction_size), _lib.ptr(out6), _lib.ptr(mean), _lib.current_stream())
        _lib.check(rc, "isg_sequence_summary")
        ctx.save_for_backward(z, lab, lse)
        ctx.shape, ctx.in_dtype = logits.shape, logits.dtype
        ctx.mark_non_differentiable(out6)
        ctx.set_materialize_grads(False)  # (else a zero gradient for out6 is filled in on every backward pass)
        return mean, out6

    @staticmethod
    def backward(ctx, dmean, _):
        z, lab, lse = ctx.saved_tensors
        rows, cols = z.shape
        if dmean is None:
            return None, None, None, None
        up = dmean.to(torch.float32).contiguous()
        dz = torch.empty_like(z)
        rc = _lib.lib().isg_softmax_xent_mean_bwd(_lib.ptr(z), _lib.ptr(lab), _lib.ptr(lse), _lib.ptr(up), rows, cols,
                                                  dtype_code(z.dtype), _lib.ptr(dz), _lib.current_stream())
        _lib.check(rc, "isg_softmax_xent_mean_bwd")
        return dz.reshape(ctx.shape).to(ctx.in_dtype), None, None, None


def sequence_loss(logits, targets, durations, prediction_size):
    """(mean loss, float64 [6] = correct, error sum, error max, live steps, accuracy, mean error); logits [B,T,K],
    targets int [B,T,2] as (x, y) (class = x * prediction_size + y), durations int [B]."""
    return _SequenceLoss.apply(logits, targets, durations, prediction_size)


def softmax_xent_rows(logits, labels, weights=None, argmax_out=None):
    """Per-row sparse softmax cross-entropy (optionally times a per-row weight).  ``argmax_out``
    (int32, one per row) receives the first-maximum class of each row from the same pass."""
    return _XentRows.apply(logits, labels, weights, argmax_out)


def softmax_xent_mean(logits, labels, weights=None, argmax_out=None):
    """mean over every row (reference polyrnn/Model.py:98-100)."""
    return softmax_xent_rows(logits, labels, weights, argmax_out).mean()


def sequence_metrics(pred, targets, durations, prediction_size):
    """Masked accuracy / mean L2 error / max L2 error of reference polyrnn/Model.py:44-54 from ONE kernel:
    pred int32 [B, T] argmax classes, targets int [B, T, 2] (x, y), durations int [B].
    Returns a float64 device tensor [4] = (correct, error sum, error max, number of live steps)."""
    B, T = targets.shape[0], targets.shape[1]
    pred = pred.reshape(-1).to(torch.int32).contiguous()
    tg = targets.to(torch.int32).contiguous()
    dur = durations.to(torch.int32).contiguous()
    out = torch.empty([4], dtype=torch.float64, device=pred.device)
    rc = _lib.lib().isg_sequence_metrics(_lib.ptr(pred), _lib.ptr(tg), _lib.ptr(dur), B, T, int(prediction_size),
                                         _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_sequence_metrics")
    return out


def argmax_rows(logits):
    """int64 argmax over the last axis, first maximum wins (tf.argmax)."""
    z, rows, cols = _rows(logits)
    out = torch.empty([rows], dtype=torch.int32, device=z.device)
    rc = _lib.lib().isg_argmax_rows(_lib.ptr(z), rows, cols, dtype_code(z.dtype), _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_argmax_rows")
    return out.to(torch.int64).reshape(logits.shape[:-1])


def prediction_max(logits, prediction_size):
    """[..., 2] = (argmax // P, argmax % P), i.e. (x, y) (reference polyrnn/Model.py:91-94)."""
    a = argmax_rows(logits)
    return torch.stack([a // prediction_size, a % prediction_size], dim=-1)


def sample_rows(logits, seed, row_offset=0):
    """One categorical sample per row of softmax(logits) (tf.multinomial's role)."""
    z, rows, cols = _rows(logits)
    out = torch.empty([rows], dtype=torch.int32, device=z.device)
    rc = _lib.lib().isg_sample_rows(_lib.ptr(z), rows, cols, dtype_code(z.dtype), int(seed), int(row_offset),
                                    _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_sample_rows")
    return out.to(torch.int64).reshape(logits.shape[:-1])


def sample_actions(logits, seed, grid_side, row_offset=0, row_offset_dev=None, out=None):
    """One categorical sample per row, delivered as the action (x, y) = (class // P, class % P), int32
    [rows, 2], from one launch (reference polyrl/ReinforceOverfitExperiment.py:101-105).
    ``row_offset_dev`` (uint64/int64 device scalar) is added to ``row_offset`` on the device."""
    z, rows, cols = _rows(logits)
    if out is None:
        out = torch.empty([rows, 2], dtype=torch.int32, device=z.device)
    rc = _lib.lib().isg_sample_actions(_lib.ptr(z), rows, cols, dtype_code(z.dtype), int(seed), int(row_offset),
                                       _lib.ptr(row_offset_dev), int(grid_side), _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_sample_actions")
    return out


def decode_begin(true_vertices, num_true, hist, steps):
    """isg_decode_begin: what the decode loop starts from, one launch (reference polyrnn/Model.py:218-230).
    true_vertices int32 [B, vmax, 2], num_true int32 [B] -> (close_xy [B,2], first history points [B,hist,2], the
    loop's own copy of them, required [B] = 1, durations [B] = 0, predicted [B,steps,2] = 0), all int32."""
    B, vmax = true_vertices.shape[0], true_vertices.shape[1]
    dev = true_vertices.device
    assert true_vertices.dtype == torch.int32 and num_true.dtype == torch.int32
    new = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    close, pts0, pts = new(B, 2), new(B, hist, 2), new(B, hist, 2)
    required, durations, predicted = new(B), new(B), new(B, steps, 2)
    rc = _lib.lib().isg_decode_begin(_lib.ptr(true_vertices.contiguous()), _lib.ptr(num_true.contiguous()), B, vmax,
                                     int(hist), int(steps), _lib.ptr(close), _lib.ptr(pts0) if hist else None,
                                     _lib.ptr(pts) if hist else None, _lib.ptr(required), _lib.ptr(durations),
                                     _lib.ptr(predicted), _lib.current_stream())
    _lib.check(rc, "isg_decode_begin")
    return close, pts0, pts, required, durations, predicted


def decode_polygons(predicted, durations, first_points, scale, prepend_history, slice_by_duration, vmax):
    """isg_decode_polygons: the decoded polygons as the rasteriser takes them (reference polyrnn/Model.py:252-260):
    (polygons int32 [B, vmax, 2], counts int32 [B])."""
    B, T = predicted.shape[0], predicted.shape[1]
    hist = first_points.shape[1]
    vp = torch.empty([B, vmax, 2], dtype=torch.int32, device=predicted.device)
    count = torch.empty([B], dtype=torch.int32, device=predicted.device)
    rc = _lib.lib().isg_decode_polygons(_lib.ptr(predicted), _lib.ptr(durations), _lib.ptr(first_points) if hist else None,
                                        B, T, hist, int(bool(prepend_history)), int(bool(slice_by_duration)), int(scale),
                                        int(vmax), _lib.ptr(vp), _lib.ptr(count), _lib.current_stream())
    _lib.check(rc, "isg_decode_polygons")
    return vp, count


def column_sums(x, out=None):
    """isg_column_sums: float32 [cols] = sum over the rows of x [rows, cols] (float32 or bfloat16), a fixed order of
    additions - the dense head's bias gradient over all (sample, step) rows (a library reduction over a tall matrix
    takes 0.27 ms for 20,480 x 785, this 0.02).  ``out``: where to write (float32 [cols], contiguous)."""
    rows, cols = x.shape
    x = x.contiguous()
    if out is None:
        out = torch.empty([cols], dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == cols
    ws = torch.empty([_lib.lib().isg_column_sums_workspace_bytes(int(cols)) // 4], dtype=torch.float32, device=x.device)
    rc = _lib.lib().isg_column_sums(_lib.ptr(x), int(rows), int(cols), dtype_code(x.dtype), _lib.ptr(out), _lib.ptr(ws),
                                    _lib.current_stream())
    _lib.check(rc, "isg_column_sums")
    return out


def dense_few_rows(flat, w, b, slices=None):
    """flat [M, K] @ w [K, N] + b in the operands' type for few rows against a long reduction axis (a decode or
    rollout step's head): one library GEMM has M N / tile output tiles for 256 CUs, so the reduction axis is cut
    into ``slices`` batched GEMMs with float results, summed with the bias by one launch (isg_sum_slices_bias).
    None when the shape does not call for it (the caller's plain GEMM then)."""
    M, K = flat.shape
    N = b.numel()  # (w may carry zero columns beyond the bias's classes: a copy padded to a multiple of 8)
    if slices is None:  # (measured at K = 12,544, N = 785: 16 rows 26 -> 19 us, 2,048 rows 125 -> 82 us)
        slices = 8 if M <= 256 else 4
    # (the rows may lie apart - one step's outputs inside a sequence's kept outputs -, each contiguous in itself)
    if not (flat.is_cuda and flat.dtype == torch.bfloat16 and w.dtype == flat.dtype and flat.stride(1) == 1 and
            (M == 1 or (flat.stride(0) >= K and flat.stride(0) % 8 == 0)) and flat.data_ptr() % 16 == 0 and
            w.is_contiguous() and M * N <= (1 << 21) and K % (slices * 8) == 0 and K // slices >= 512):
        return None
    try:
        part = torch.bmm(flat.unflatten(1, (slices, K // slices)).transpose(0, 1),
                         w.view(slices, K // slices, w.shape[1]), out_dtype=torch.float32)
    except TypeError:  # (a torch without float32 results from a bf16 batched GEMM)
        return None
    return sum_slices_bias(part, b)


def sum_slices_bias(part, b):
    """part float32 [slices, M, Np] summed over the slices + b [N], N <= Np (the first N columns are kept), in b's type
    (float32 or bf16), in one launch."""
    slices, M, Np = part.shape
    N = b.numel()
    assert part.dtype == torch.float32 and part.is_contiguous() and N <= Np
    out = torch.empty([M, N], dtype=b.dtype, device=part.device)
    rc = _lib.lib().isg_sum_slices_bias_padded(_lib.ptr(part), slices, M, N, Np, _lib.ptr(b.contiguous()),
                                               dtype_code(out.dtype), _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "isg_sum_slices_bias_padded")
    return out


def rows_dot(flat, w, b=None):
    """flat [M, K] . w [K] (or [K, 1]) + b as float32 [M], operands bf16 (isg_rows_dot): a dense layer with one output
    column.  None when the operands do not fit (the caller's library GEMM then)."""
    M, K = flat.shape
    if not (flat.is_cuda and flat.dtype == torch.bfloat16 and w.dtype == flat.dtype and flat.is_contiguous() and
            w.is_contiguous() and w.numel() == K and K % 8 == 0 and (b is None or b.dtype == flat.dtype) and
            flat.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0):
        return None
    out = torch.empty([M], dtype=torch.float32, device=flat.device)
    rc = _lib.lib().isg_rows_dot(_lib.ptr(flat), _lib.ptr(w), _lib.ptr(b), M, K, dtype_code(flat.dtype), _lib.ptr(out),
                                 _lib.current_stream())
    _lib.check(rc, "isg_rows_dot")
    return out


def decode_step(logits, grid_side, end_token, close_xy, required, durations, predicted, step, history_points):
    """isg_decode_step: argmax of every row, then the decode loop's bookkeeping (stop rule, predicted polygon,
    history points of the next step) on the device.  logits [B, K]; all other tensors int32 and updated in
    place: close_xy [B,2], required [B], durations [B], predicted [B,T,2], history_points [B,hist,2]."""
    z, rows, cols = _rows(logits)
    hist = history_points.shape[1]
    for t in (close_xy, required, durations, predicted, history_points):
        assert t.dtype == torch.int32 and t.is_contiguous()
    rc = _lib.lib().isg_decode_step(_lib.ptr(z), rows, cols, dtype_code(z.dtype), int(grid_side), int(bool(end_token)),
                                    _lib.ptr(close_xy), _lib.ptr(required), _lib.ptr(durations), _lib.ptr(predicted),
                                    predicted.shape[1], int(step), _lib.ptr(history_points) if hist else None, hist,
                                    _lib.current_stream())
    _lib.check(rc, "isg_decode_step")
