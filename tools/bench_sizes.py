#!/usr/bin/env python3
"""Development sweep: forward / scale-gradient call time against tensor size for each traversal mode (looks for
latency cliffs between the launch-bound and the streaming regime).  Prints microseconds per call (wall, asynchronous queue)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learned_quantization_amd as lq
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)

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
           ("column C=512 inner=1 (n/512,512,1)", lambda n: (n // 512, 512, 1)),
           ("rows of 32 (1,n/32,32)", lambda n: (1, n // 32, 32)),
           ("rows of 100 (1,n/100,100)", lambda n: (1, n // 100, 100)),
           ("rows of 1000 (1,n/1000,1000)", lambda n: (1, n // 1000, 1000)),
           ("rows of 1031 (1,n/1031,1031)", lambda n: (1, n // 1031, 1031)),
           ("rows of 20000 (1,n/20000,20000)", lambda n: (1, max(1, n // 20000), 20000)),
           ("G=16 inner=64 (n/1024,16,64)", lambda n: (n // 1024, 16, 64)),
           ("G=256 inner=8 (n/2048,256,8)", lambda n: (n // 2048, 256, 8)),
           ("G=3 inner=big (1,3,n/3)", lambda n: (1, 3, n // 3)),
           ("outer=64 G=3 (64,3,n/192)", lambda n: (64, 3, n // 192))]
if len(sys.argv) > 1 and sys.argv[1] == "more":
    LAYOUTS = LAYOUTS[7:]
else:
    LAYOUTS = LAYOUTS[:7]
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
