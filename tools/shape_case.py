#!/usr/bin/env python3
"""One descriptor of the (outer, G, inner) space through K1 / K2+K3 / K4, N launches each, on rotating buffers -- the
program rocprofv3 is pointed at by tools/prof_shapes.sh (kernel durations and PMC counters per traversal mode).

    python3 tools/shape_case.py <outer> <G> <inner> [--iters 20] [--ops k1,k2,k4] [--lam 1e-11]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)

ap = argparse.ArgumentParser()
ap.add_argument("outer", type=int)
ap.add_argument("G", type=int)
ap.add_argument("inner", type=int)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--ops", default="k1,k2,k4")
ap.add_argument("--lam", type=float, default=1e-11)
ap.add_argument("--sets", type=int, default=3)
args = ap.parse_args()

dev = torch.device("cuda:0")
outer, G, inner = args.outer, args.G, args.inner
n = outer * G * inner
lib = lq._hip.load()
if os.environ.get("LQ_DEV_FLAGS"):          # development library only: lq_dev_set_flags (lq_stream2.hpp)
    import ctypes
    lib.lq_dev_set_flags.restype = ctypes.c_int
    lib.lq_dev_set_flags.argtypes = [ctypes.c_int]
    assert lib.lq_dev_set_flags(int(os.environ["LQ_DEV_FLAGS"])) == 0
sets = []
for k in range(args.sets):
    P = torch.rand(n, device=dev) * 200 - 100
    dy = torch.randn(n, device=dev) * 1e-3
    sets.append((P, dy, torch.empty(n, device=dev)))
s = torch.rand(G, device=dev) + 0.5
ds = torch.empty(G, device=dev)
ws = torch.empty(lib.lq_workspace_bytes(outer, G, inner), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for op in args.ops.split(","):
    for i in range(args.iters):
        P, dy, out = sets[i % args.sets]
        if op == "k1":
            rc = lib.lq_fq_forward(P.data_ptr(), s.data_ptr(), out.data_ptr(), None, 0, outer, G, inner, None)
        elif op == "k2":
            rc = lib.lq_fq_scale_grad(P.data_ptr(), s.data_ptr(), dy.data_ptr(), args.lam, ds.data_ptr(), None, ws.data_ptr(),
                                      ws.numel(), outer, G, inner, None)
        else:
            rc = lib.lq_fq_fwd_bwd_fused(P.data_ptr(), s.data_ptr(), dy.data_ptr(), args.lam, out.data_ptr(), ds.data_ptr(),
                                         ws.data_ptr(), ws.numel(), outer, G, inner, None)
        assert rc == 0, lib.lq_last_error()
    torch.cuda.synchronize()
print("done", outer, G, inner, n)
