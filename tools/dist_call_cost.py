#!/usr/bin/env python3
"""Host-side cost of one asynchronous 3-float all-reduce on a one-rank RCCL group (what bench.py --gpus N issues per step):
python3 tools/dist_call_cost.py  -> microseconds per call, issue-only and with the stream wait bench.py does two steps later."""
import os
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
bufs = [torch.zeros(3, device=dev) for _ in range(2)]
for _ in range(20):
    dist.all_reduce(bufs[0], op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
pend = [None, None]
for i in range(N):
    b = i & 1
    if pend[b] is not None:
        pend[b].wait()
    pend[b] = dist.all_reduce(bufs[b], op=dist.ReduceOp.AVG, async_op=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per async all-reduce (+ wait of the one two steps back): {(t1 - t0) / N * 1e6:.1f} us; "
      f"including the drain of the queue: {(t2 - t0) / N * 1e6:.1f} us")
dist.destroy_process_group()
