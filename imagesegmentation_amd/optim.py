"""Flat gradient bucket + TensorFlow-form Adam, shared by trainers that are not a polyrnn Model."""
import math

import torch

from . import parallel


def flatten_parameters_(params):
    """Moves the parameters into ONE float32 buffer (each keeps its shape as a view of it) and returns the
    buffer: the optimiser step is then a single pass over it."""
    params = list(params)
    n = sum(p.numel() for p in params)
    flat = torch.empty([n], dtype=torch.float32, device=params[0].device)
    off = 0
    with torch.no_grad():
        for p in params:
            k = p.numel()
            flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + k].view_as(p)
            off += k
    return flat


def adam_tf_step_(flat_param, m, v, flat_grad, t, learning_rate, clip_norm=None, norm=None, b1=0.9, b2=0.999,
                  eps=1e-8):
    """tf.train.AdamOptimizer's update after tf.clip_by_global_norm (reference polyrnn/Model.py:106-112) on flat
    buffers: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps).  One HIP launch on the
    GPU; the same arithmetic with torch ops for host tensors (the CPU tests of the data-parallel step)."""
    lr_t = learning_rate * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    if flat_param.is_cuda:
        from . import ops
        ops.adam_tf_(flat_param, m, v, flat_grad, lr_t, b1, b2, eps, grad_norm=norm if clip_norm else None,
                     clip_norm=clip_norm)
        return
    with torch.no_grad():
        g = flat_grad if not clip_norm else flat_grad * (clip_norm / torch.clamp(norm, min=clip_norm))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        flat_param.addcdiv_(m, v.sqrt().add_(eps), value=-lr_t)


class FlatAdam:
    """Parameters and their ``.grad`` are views of one float32 buffer each, so that the data-parallel
    exchange is ONE all-reduce, the global norm one reduction and the update one pass.  ``step`` =
    all-reduce, clip_by_global_norm, Adam as tf.train.AdamOptimizer computes it
    (lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps))."""

    def __init__(self, params, learning_rate=1e-4, clip_norm=None, process_group=None):
        self.params = list(params)
        self.learning_rate, self.clip_norm, self.process_group = learning_rate, clip_norm, process_group
        self.flat_param = flatten_parameters_(self.params)
        self.flat_grad = torch.zeros_like(self.flat_param)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.m = torch.zeros_like(self.flat_param)
        self.v = torch.zeros_like(self.flat_param)
        self.t = 0

    def zero_grad(self):
        self.flat_grad.zero_()

    def step(self, all_reduce=True, average=True):
        flat = self.flat_grad
        if all_reduce:
            parallel.allreduce_gradients_(flat, self.process_group, average=average)
        norm = flat.norm()
        self.t += 1
        adam_tf_step_(self.flat_param, self.m, self.v, flat, self.t, self.learning_rate, self.clip_norm, norm)
        return norm
