#!/usr/bin/env python3
"""Writes tests/golden/convlstm_golden.npz: the float64 oracle's outputs (logits, loss, metrics and a
three-number signature of every parameter gradient) for small instances of the experiment graphs whose
weights and inputs are rebuilt from fixed seeds (tests/helpers_convlstm.py::golden_case).
The reference cannot produce these (TensorFlow 1.x is not importable), so the vectors freeze the
ORACLE: they guard the restatement and the kernels against silent drift, they do not pin the
restatement to the reference.

    python tests/golden/make_convlstm_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers_convlstm import GOLDEN_CASES, GOLDEN_P, flat_grad_list, golden_case, grad_signature  # noqa: E402
from oracle import convlstm as oc  # noqa: E402


def main():
    out = {}
    for name in sorted(GOLDEN_CASES):
        params, images, histories, durations, targets = golden_case(name)
        loss, logits, grads = oc.model_loss_and_grads(params, images, histories, durations, targets, GOLDEN_P)
        acc, avg, mx = oc.metrics(logits, targets, durations, GOLDEN_P)
        out[name + "/logits"] = logits
        out[name + "/scalars"] = np.array([loss, acc, avg, mx])
        out[name + "/grad_signatures"] = np.stack([grad_signature(g) for g in flat_grad_list(grads)])
        out[name + "/input_checksum"] = np.array([images.sum(), histories.sum(), durations.sum(), targets.sum(),
                                                  sum(np.abs(p["w"]).sum() for p in params["encoder"])])
    path = os.path.join(ROOT, "tests", "golden", "convlstm_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
