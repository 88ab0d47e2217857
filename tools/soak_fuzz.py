#!/usr/bin/env python3
"""Soak: the randomized GPU-vs-oracle sweeps of tests/test_gpu_fuzz.py with many seeds, in ONE process.
usage: python tools/soak_fuzz.py [first_seed] [n_seeds] [scale] [streaming 0|1]
streaming = 1 adds test_fuzz_streaming_size_forms (10 * scale random 4-6 M element descriptors per seed: the round-2 kernels)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
first, n, scale, streaming = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 1), (2, 8), (3, 2), (4, 0)))
import torch  # noqa: E402

dev = torch.device("cuda:0")
os.environ["LQ_FUZZ_SCALE"] = str(scale)
mod = None
for seed in range(first, first + n):
    os.environ["LQ_FUZZ_SEED"] = str(seed * 1000)
    mod = importlib.import_module("test_gpu_fuzz") if mod is None else importlib.reload(mod)
    t0 = time.time()
    mod.test_fuzz_forward_backward(dev)
    mod.test_fuzz_misaligned_views(dev)
    mod.test_fuzz_penalty_terms(dev)
    if streaming:
        mod.test_fuzz_streaming_size_forms(dev)
    print(f"seed shift {seed * 1000}: ok ({time.time() - t0:.1f} s)", flush=True)
print("soak passed")
