#!/usr/bin/env python3
"""Development probe: host-side cost per call of the Python layer above the C ABI (tiny tensors, launches are asynchronous,
so the loop time is the host time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learned_quantization_amd as lq
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)
from learned_quantization_amd import _hip

dev = torch.device("cuda:0")
lib = _hip.load()
P = torch.randn(128, device=dev) * 0.05
s = torch.full((1,), 1e-3, device=dev)
dy = torch.randn(128, device=dev) * 1e-3
out = torch.empty_like(P)
ds = torch.empty_like(s)
ws = torch.empty(lib.lq_workspace_bytes(1, 1, 128), dtype=torch.uint8, device=dev)
N = 3000


def loop(fn):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / N * 1e6, (time.perf_counter() - t0) / N * 1e6


pp, sp, op, dp, dsp, wp, wn = P.data_ptr(), s.data_ptr(), out.data_ptr(), dy.data_ptr(), ds.data_ptr(), ws.data_ptr(), ws.numel()
print("raw ctypes lq_fq_forward         host %.1f us  (with drain %.1f us)" % loop(lambda: lib.lq_fq_forward(pp, sp, op, None, 0, 1, 1, 128, None)))
print("raw ctypes lq_fq_scale_grad      host %.1f us  (with drain %.1f us)" % loop(lambda: lib.lq_fq_scale_grad(pp, sp, dp, 1e-11, dsp, None, wp, wn, 1, 1, 128, None)))
print("ops.fq_forward                   host %.1f us  (with drain %.1f us)" % loop(lambda: lq.fq_forward(P, s)))
print("ops.fq_scale_grad                host %.1f us  (with drain %.1f us)" % loop(lambda: lq.fq_scale_grad(P, s, dy, 1e-11)))
Pg = P.clone().requires_grad_(True)
sg = s.clone().requires_grad_(True)


def fb():
    o = lq.my_custom_gradient(Pg, sg, 1e-11)
    o.backward(dy)


print("my_custom_gradient fwd+bwd       host %.1f us  (with drain %.1f us)" % loop(fb))
print("torch reference: (P/s).floor()*s host %.1f us  (with drain %.1f us)" % loop(lambda: torch.floor(P / s) * s))
