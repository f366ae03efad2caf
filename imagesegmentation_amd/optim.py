"""Flat gradient bucket + TensorFlow-form Adam, shared by trainers that are not a polyrnn Model."""
import math

import torch

from . import parallel


class FlatAdam:
    """Parameters' ``.grad`` are views of one float32 bucket, so that the data-parallel exchange is
    ONE all-reduce and the global norm one reduction.  ``step`` = all-reduce, clip_by_global_norm,
    Adam as tf.train.AdamOptimizer computes it (lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);
    p -= lr_t m / (sqrt(v) + eps))."""

    def __init__(self, params, learning_rate=1e-4, clip_norm=None, process_group=None):
        self.params = list(params)
        self.learning_rate, self.clip_norm, self.process_group = learning_rate, clip_norm, process_group
        n = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros([n], dtype=torch.float32, device=self.params[0].device)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def zero_grad(self):
        self.flat_grad.zero_()

    def step(self, all_reduce=True, average=True):
        flat = self.flat_grad
        if all_reduce:
            parallel.allreduce_gradients_(flat, self.process_group, average=average)
        norm = flat.norm()
        if self.clip_norm is not None:
            flat.mul_(self.clip_norm / torch.clamp(norm, min=self.clip_norm))
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = self.learning_rate * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
        grads = [p.grad for p in self.params]
        with torch.no_grad():
            torch._foreach_mul_(self.m, b1)
            torch._foreach_add_(self.m, grads, alpha=1 - b1)
            torch._foreach_mul_(self.v, b2)
            torch._foreach_addcmul_(self.v, grads, grads, value=1 - b2)
            denom = torch._foreach_sqrt(self.v)
            torch._foreach_add_(denom, eps)
            torch._foreach_addcdiv_(self.params, self.m, denom, value=-lr_t)
        return norm
