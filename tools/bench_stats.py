#!/usr/bin/env python3
"""Development probe: throughput of the tracking-statistics kernels (SURVEY f-2): q min/max, histogram/unique count, max|q| per slice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learned_quantization_amd as lq
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, shape, sshape, sval in (("ResNet conv 3x3x512x512 channelwise", (3, 3, 512, 512), (1, 1, 512, 1), 1e-3),
                                 ("BENCH activation per-channel", (256, 3, 224, 224), (1, 3, 1, 1), 1.0),
                                 ("ResNet conv, init scale (wide integer range)", (3, 3, 512, 512), (1, 1, 512, 1), 1.1920929e-05)):
    P = torch.randn(shape, device=dev) * 0.05 if sval < 1 else torch.rand(shape, device=dev) * 255
    s = torch.full(sshape, sval, device=dev)
    n = P.numel()
    t_u = timed(lambda: lq.q_unique(P, s))
    nu = lq.q_unique(P, s)
    t_t = timed(lambda: torch.unique(torch.floor(P / s)).numel())
    print(f"{name:48s} n={n/1e6:5.1f}M  q_unique {t_u*1e6:8.1f} us ({4*n/t_u/1e9:6.0f} GB/s)  torch.unique {t_t*1e6:9.1f} us  -> {t_t/t_u:5.1f}x   unique={nu if isinstance(nu, int) else int(nu[0].numel())}", flush=True)

# max|q| over an axis (custom_callbacks.py:98-99): the coalesced flat-stream kernel against the torch expression it replaces
for name, shape, sshape, axis in (("Dense 784x128 columnwise, axis=1 (post = 1)", (784, 128), (1, 128), 1),
                                 ("conv 3x3x512x512 channelwise, axis=1", (3, 3, 512, 512), (1, 1, 512, 1), 1),
                                 ("conv 3x3x64x128 channelwise, axis=1", (3, 3, 64, 128), (1, 1, 64, 1), 1),
                                 ("BENCH activation per-channel, axis=1", (256, 3, 224, 224), (1, 3, 1, 1), 1)):
    P = torch.randn(shape, device=dev) * 0.05
    s = torch.full(sshape, 1e-3, device=dev)
    t_k = timed(lambda: lq.q_absmax_over_axis(P, s, axis))
    t_t = timed(lambda: torch.floor(P / s).abs().amax(dim=axis))
    print(f"{name:48s} n={P.numel()/1e6:5.2f}M  lq_q_absmax_over_axis {t_k*1e6:8.1f} us ({4*P.numel()/t_k/1e9:6.0f} GB/s)  torch {t_t*1e6:8.1f} us", flush=True)
