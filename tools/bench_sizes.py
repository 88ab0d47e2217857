#!/usr/bin/env python3
"""Development sweep: forward / scale-gradient call time against tensor size for each traversal mode (looks for
latency cliffs between the launch-bound and the streaming regime).  Prints microseconds per call (wall, asynchronous queue)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learned_quantization_amd as lq

dev = torch.device("cuda:0")
lib = lq._hip.load()


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


LAYOUTS = [("flat scalar (1,1,n)", lambda n: (1, 1, n)),
           ("rows of 512 (1,n/512,512)", lambda n: (1, n // 512, 512)),
           ("rows of 4608 (1,n/4608,4608)", lambda n: (1, max(1, n // 4608), 4608)),
           ("channelwise (9,n/9/512,512)", lambda n: (9, max(1, n // 9 // 512), 512)),
           ("column C=128 (n/128,128,1)", lambda n: (n // 128, 128, 1)),
           ("column C=10 (n/10,10,1)", lambda n: (n // 10, 10, 1)),
           ("column C=512 inner=1 (n/512,512,1)", lambda n: (n // 512, 512, 1))]
for name, mk in LAYOUTS:
    row = []
    for n in (65536, 262144, 1048576, 4194304, 16777216):
        outer, G, inner = mk(n)
        m = outer * G * inner
        P = torch.randn(m, device=dev) * 0.05
        dy = torch.randn(m, device=dev) * 1e-3
        out = torch.empty(m, device=dev)
        s = torch.full((G,), 1e-3, device=dev)
        ds = torch.empty(G, device=dev)
        ws = torch.empty(max(lib.lq_workspace_bytes(outer, G, inner), 16), dtype=torch.uint8, device=dev)
        tf = timed(lambda: lib.lq_fq_forward(P.data_ptr(), s.data_ptr(), out.data_ptr(), None, 0, outer, G, inner, None))
        tb = timed(lambda: lib.lq_fq_scale_grad(P.data_ptr(), s.data_ptr(), dy.data_ptr(), 1e-11, ds.data_ptr(), None, ws.data_ptr(), ws.numel(), outer, G, inner, None))
        row.append(f"{m/1e6:5.2f}M {tf:6.1f}/{tb:6.1f}")
    print(f"{name:36s} " + " | ".join(row), flush=True)
