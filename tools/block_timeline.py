#!/usr/bin/env python3
"""Block timeline of one multi-tensor batch launch (development library: make -C learned_quantization_amd/csrc dev).

    LQ_HIP_LIB=learned_quantization_amd/csrc/liblq_hip_dev.so python3 tools/block_timeline.py imagenette:channelwise fwd|bwd|bwd_oihw

Every block records wall_clock64() (100 MHz) at its start and end; prints when blocks start and finish (histogram in microseconds
from the first start), the distribution of block lifetimes and the number of blocks alive over time."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)


def main():
    config, orient = sys.argv[1].split(":")
    which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    dev = torch.device("cuda:0")
    lam = 1e-11
    model = lq.build_model(config, mode="nq", value=(1e-10, lam) if config == "resnet50" else lam, seed=42, orientation=orient, device=dev,
                           kernel_storage=os.environ.get("LQ_TIMELINE_STORAGE", "hwio"))     # oihw: the trainer's default (plain launches)
    batch = lq.FakeQuantBatch(model, hwio_out=os.environ.get('LQ_TIMELINE_HWIO_OUT', '1') != '0')
    g = torch.Generator(device=dev).manual_seed(42)
    dys = [torch.empty_like(e.param.data).normal_(generator=g) * 1e-3 for e in batch.entries]
    dys_o = [d.permute(3, 2, 0, 1).contiguous() if e.out_oihw is not None else d for e, d in zip(batch.entries, dys)]
    lib = lq._hip.load()
    lib.lq_dev_set_trace.restype = ctypes.c_int
    lib.lq_dev_set_trace.argtypes = [ctypes.c_void_p]
    sp = lq._hip.stream_ptr(dev)
    ptrs = (ctypes.c_void_p * len(dys))(*[d.data_ptr() for d in dys])
    ptrs_o = (ctypes.c_void_p * len(dys))(*[d.data_ptr() for d in dys_o])

    def launch():
        if which == "fwd":
            lib.lq_batch_forward(batch._handle, sp)
        elif which == "bwd":
            lib.lq_batch_scale_grad(batch._handle, ptrs, batch.ws.data_ptr(), batch.ws.numel(), sp)
        else:
            lib.lq_batch_scale_grad_oihw(batch._handle, ptrs_o, batch.ws.data_ptr(), batch.ws.numel(), sp)

    for _ in range(5):
        lib.lq_batch_forward(batch._handle, sp)
        launch()
    torch.cuda.synchronize()
    nmax = 1 << 16
    buf = torch.zeros(2 * nmax, dtype=torch.int64, device=dev)
    assert lib.lq_dev_set_trace(buf.data_ptr()) == 0
    launch()
    torch.cuda.synchronize()
    assert lib.lq_dev_set_trace(None) == 0
    t = buf.cpu().numpy().reshape(-1, 2)
    used = t[:, 0] != 0
    t = t[used]
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0
    end = (t[:, 1] - t0) / 100.0
    life = end - start
    print(f"# {config}:{orient} {which}: {len(t)} blocks, first start 0, last end {end.max():.2f} us")
    print(f"# lifetime us: min {life.min():.2f} p10 {np.percentile(life,10):.2f} median {np.median(life):.2f} p90 {np.percentile(life,90):.2f} max {life.max():.2f}")
    edges = np.arange(0, end.max() + 2, 2.0)
    hs, _ = np.histogram(start, edges)
    he, _ = np.histogram(end, edges)
    print("# t_us   starts   ends   alive_at_t")
    for i, e in enumerate(edges[:-1]):
        alive = int(((start <= e) & (end > e)).sum())
        print(f"{e:6.1f} {hs[i]:8d} {he[i]:6d} {alive:8d}")
    # by block index (tasks occupy contiguous index ranges, heaviest first): when do the blocks of each range end
    nb = len(t)
    step = max(1, nb // 16)
    print("# block range: mean start / mean end / max end (us)")
    for lo in range(0, nb, step):
        sl = slice(lo, min(lo + step, nb))
        print(f"  {lo:5d}-{min(lo + step, nb) - 1:5d}  {start[sl].mean():6.2f} {end[sl].mean():6.2f} {end[sl].max():6.2f}")
    # by block index: which blocks started late
    order = np.argsort(start)
    idx = np.nonzero(used)[0]
    late = idx[order[-10:]]
    print("# last 10 blocks to start (index, start, lifetime):", [(int(i), round(float(start[order[-10 + k]]), 1), round(float(life[order[-10 + k]]), 1)) for k, i in enumerate(late)])


if __name__ == "__main__":
    main()
