#!/usr/bin/env python3
"""Development sweep: K1 / K2+K3 / K4 bandwidth over the (outer, G, inner) descriptor space, to find performance
cliffs of the traversal modes.  ~150 MB tensors unless noted.  Prints one line per descriptor."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)

dev = torch.device("cuda:0")
CASES = [
    ("NCHW per-channel (BENCH)", (256, 3, 50176)),
    ("per-tensor flat", (1, 1, 38535168)),
    ("NHWC per-channel C=3", (12845056, 3, 1)),
    ("NHWC per-channel C=64", (602112, 64, 1)),
    ("NHWC per-channel C=256", (150528, 256, 1)),
    ("columnwise dense 6144x6144", (6144, 6144, 1)),
    ("rowwise 1M rows x 32", (1, 1048576, 32)),
    ("rowwise 65536 rows x 512", (1, 65536, 512)),
    ("rowwise 16384 rows x 2048", (1, 16384, 2048)),
    ("rowwise 8192 rows x 4100 (L%4==0)", (1, 8192, 4100)),
    ("rowwise 8192 rows x 4099 (scalar path)", (1, 8192, 4099)),
    ("channelwise HWIO 3x3x2048x2048", (9, 2048, 2048)),
    ("inner=8, outer=2048, G=2048", (2048, 2048, 8)),
]


def timed(fn, n=60):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if len(sys.argv) > 1 and sys.argv[1] == "fill":
    # chunk-fill probe: rows of 12 / 12.25 / 12.5 / 13 chunks of 4096 elements
    CASES = [(f"fill probe (256, 3, {L})", (256, 3, L)) for L in (49152, 50176, 51200, 53248, 50176)]
for name, (outer, G, inner) in CASES:
    n = outer * G * inner
    sets = []
    for k in range(3):
        P = (torch.rand(n, device=dev) * 200 - 100)
        dy = torch.randn(n, device=dev) * 1e-3
        sets.append((P, dy, torch.empty(n, device=dev)))
    s = torch.rand(G, device=dev) + 0.5
    ds = torch.empty(G, device=dev)
    lib = lq._hip.load()
    ws = torch.empty(lib.lq_workspace_bytes(outer, G, inner), dtype=torch.uint8, device=dev)
    it = [0]

    def fwd():
        P, dy, out = sets[it[0] % 3]
        it[0] += 1
        lib.lq_fq_forward(P.data_ptr(), s.data_ptr(), out.data_ptr(), None, 0, outer, G, inner, None)

    def bwd():
        P, dy, out = sets[it[0] % 3]
        it[0] += 1
        lib.lq_fq_scale_grad(P.data_ptr(), s.data_ptr(), dy.data_ptr(), 1e-11, ds.data_ptr(), None, ws.data_ptr(), ws.numel(), outer, G, inner, None)

    def fused():
        P, dy, out = sets[it[0] % 3]
        it[0] += 1
        lib.lq_fq_fwd_bwd_fused(P.data_ptr(), s.data_ptr(), dy.data_ptr(), 1e-11, out.data_ptr(), ds.data_ptr(), ws.data_ptr(), ws.numel(), outer, G, inner, None)

    tf, tb, tk = timed(fwd), timed(bwd), timed(fused)
    print(f"{name:42s} n={n/1e6:6.1f}M  K1 {8*n/tf/1e9:6.0f} GB/s  K2+K3 {8*n/tb/1e9:6.0f} GB/s  K4 {12*n/tk/1e9:6.0f} GB/s  ws={ws.numel()/1e6:.2f} MB",
          flush=True)
    del sets
