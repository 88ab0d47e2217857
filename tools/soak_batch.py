#!/usr/bin/env python3
"""Soak: tests/test_gpu_batch.py::test_batch_of_odd_conv_shapes_equals_single_tensor_ops (22 random conv kernels per seed and
orientation through the multi-tensor batch, against the single-tensor ops bit for bit and the oracle) with many seeds, in ONE process.
usage: python tools/soak_batch.py [first_seed] [n_seeds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
first, n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 2), (2, 24)))
import torch  # noqa: E402

import test_gpu_batch as T  # noqa: E402

dev = torch.device("cuda:0")
for seed in range(first, first + n):
    t0 = time.time()
    for orient in ("channelwise", "rowwise", "columnwise", "scalar"):
        T.test_batch_of_odd_conv_shapes_equals_single_tensor_ops.__wrapped__(dev, orient, seed) if hasattr(
            T.test_batch_of_odd_conv_shapes_equals_single_tensor_ops, "__wrapped__") else T.test_batch_of_odd_conv_shapes_equals_single_tensor_ops(dev, orient, seed)
    print(f"seed {seed}: ok ({time.time() - t0:.1f} s)", flush=True)
print("soak passed")
