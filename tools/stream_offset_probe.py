#!/usr/bin/env python3
"""Does the RELATIVE placement of the streams of a BENCH-sized launch matter?  (profiles/r04/colbench_second_stream_offset.txt found one
column-tile shape whose two streams, 2 MB-aligned alike, ran three times slower.)  K1 (read P, write out), K2 (read P, read dy) and K4 on
the BENCH tensor with the second / third stream shifted by a few offsets inside one big allocation; hipEvent time over back-to-back launches.

    python3 tools/stream_offset_probe.py [--iters 60]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
args = ap.parse_args()
dev = torch.device("cuda:0")
outer, G, inner = 256, 3, 224 * 224
n = outer * G * inner
lib = lq._hip.load()
pad = (8 << 20) // 4
big = [torch.empty(n + pad, device=dev) for _ in range(3)]          # P, dy, out live in their own allocations, shifted inside them
big[0][:n].uniform_(-100, 100)
big[1].normal_().mul_(1e-3)
s = torch.tensor([0.5, 1.0, 2.0], device=dev)
ds = torch.empty(G, device=dev)
ws = torch.empty(lib.lq_workspace_bytes(outer, G, inner), dtype=torch.uint8, device=dev)
print("# base addresses mod 2 MiB (P, dy, out):", [hex(b.data_ptr() % (2 << 20)) for b in big], "distances MiB:",
      (big[1].data_ptr() - big[0].data_ptr()) / 2**20, (big[2].data_ptr() - big[0].data_ptr()) / 2**20)
offs = [0, 256, 4096, 65536, (1 << 20) + 4096, (2 << 20) + 8192, (3 << 20) + 256 * 37]


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters * 1e3


P = big[0][:n]
for name in ("k1", "k2", "k4"):
    for o_dy in ([0] if name == "k1" else offs):
        for o_out in ([0] if name == "k2" else offs):
            if name == "k4" and o_dy not in (0, offs[4]) :
                continue
            dy = big[1][o_dy // 4:o_dy // 4 + n]
            out = big[2][o_out // 4:o_out // 4 + n]
            if name == "k1":
                fn = lambda: lib.lq_fq_forward(P.data_ptr(), s.data_ptr(), out.data_ptr(), None, 0, outer, G, inner, None)       # noqa: E731
            elif name == "k2":
                fn = lambda: lib.lq_fq_scale_grad(P.data_ptr(), s.data_ptr(), dy.data_ptr(), 1e-11, ds.data_ptr(), None, ws.data_ptr(), ws.numel(), outer, G, inner, None)   # noqa: E731
            else:
                fn = lambda: lib.lq_fq_fwd_bwd_fused(P.data_ptr(), s.data_ptr(), dy.data_ptr(), 1e-11, out.data_ptr(), ds.data_ptr(), ws.data_ptr(), ws.numel(), outer, G, inner, None)   # noqa: E731
            t = timed(fn)
            print(f"{name} dy+{o_dy:>8d} out+{o_out:>8d}  {t:7.2f} us per call (incl. the finalize launch for k2 / k4)", flush=True)
