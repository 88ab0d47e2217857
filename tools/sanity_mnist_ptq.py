#!/usr/bin/env python3
"""Functional sanity run: PTQ-style training of the MNIST dense model from the reference's shipped
baseline weights on a synthetic task whose labels come from the baseline itself.  Prints the number of unique integers
and max|q| of the first layer, and the agreement with the teacher, as training proceeds."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402
from learned_quantization_amd.train import Trainer  # noqa: E402

lam = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
orient = sys.argv[2] if len(sys.argv) > 2 else "rowwise"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
dev = torch.device("cuda:0")
w = np.load(os.path.join(ROOT, "tests", "golden", "mnist_baseline_weights.npz"))
W1, b1, W2, b2 = (torch.tensor(w[k], device=dev) for k in ("W1", "b1", "W2", "b2"))


def teacher(x):
    h = torch.relu(torch.flatten(x, 1) @ W1 + b1)
    return (h @ W2 + b2).argmax(1)


tr = Trainer("mnist", "nq", lam, orient, None, device=dev, seed=42, batched=True)
with torch.no_grad():
    tr.model.dense_1.W.copy_(W1); tr.model.dense_1.b.copy_(b1); tr.model.dense_2.W.copy_(W2); tr.model.dense_2.b.copy_(b2)
g = torch.Generator(device=dev).manual_seed(0)
print_hist = True
cb = lq.NestedScaleTrackingCallback(tr.model.dense_1, "/tmp/lq_sanity_logs")
def images(n):
    """MNIST-like synthetic inputs: ~19 % of the pixels lit (the real digits' ink fraction), values in (0, 1]."""
    m = (torch.rand(n, 1, 28, 28, device=dev, generator=g) < 0.19).float()
    return m * torch.rand(n, 1, 28, 28, device=dev, generator=g)


xv = images(4096)
yv = teacher(xv)
print(json.dumps({"teacher_label_histogram": torch.bincount(yv, minlength=10).tolist()}))
for step in range(steps + 1):
    if step % (steps // 10) == 0:
        st = cb.stats()
        _, acc = tr.evaluate(xv, yv)
        print(json.dumps({"step": step, "unique_ints_W1": st["unique_k"], "max_abs_q_W1": float(st["max_k"].max()),
                          "scale_W1_mean": float(tr.model.dense_1.nested_q_w_layer.scale.mean()), "agreement_with_teacher": round(acc, 4)}), flush=True)
    x = images(32)
    tr.step(x, teacher(x))
