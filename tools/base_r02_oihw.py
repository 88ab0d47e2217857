#!/usr/bin/env python3
"""Round-2 sources (tree at ad7e194 extracted to _base_r02/, built there): the multi-tensor scale gradient with OIHW gradients
(lq_batch_scale_grad_oihw: element-wise gather, 4-byte accesses 4*ci*hw bytes apart) on one weight set -- the "before" of
lq_conv_tile.hpp's K2.  Run from _base_r02/ under rocprofv3:  python3 ../tools/base_r02_oihw.py imagenette:channelwise"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)

assert "_base_r02" in lq.__file__, lq.__file__
config, orient = sys.argv[1].split(":")
dev = torch.device("cuda:0")
lam = 1e-11
model = lq.build_model(config, mode="nq", value=(1e-10, lam) if config == "resnet50" else lam, seed=42, orientation=orient, device=dev, kernel_storage="hwio")
batch = lq.FakeQuantBatch(model)
g = torch.Generator(device=dev).manual_seed(42)
dys = [torch.randn(e.out.shape, device=dev, generator=g) * 1e-3 for e in batch.entries]
dys_o = [d.permute(3, 2, 0, 1).contiguous() if e.out_oihw is not None else d for e, d in zip(batch.entries, dys)]
lib = lq._hip.load()
sp = lq._hip.stream_ptr(dev)
ptrs_o = (ctypes.c_void_p * len(dys))(*[d.data_ptr() for d in dys_o])


def step():
    lib.lq_batch_forward(batch._handle, sp)
    lib.lq_batch_scale_grad_oihw(batch._handle, ptrs_o, batch.ws.data_ptr(), batch.ws.numel(), sp)


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
torch.cuda.synchronize()
print(json.dumps({"config": config, "orientation": orient, "sources": "round 2 (ad7e194)", "us_per_fwd_plus_oihw_scale_grad": (time.perf_counter() - t0) / 100 * 1e6}))
