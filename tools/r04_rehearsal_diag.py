#!/usr/bin/env python3
"""What differed between the three ResNet-18 runs of round 3's red test (GPUTEST_r03: test_one_rank_rccl_rehearsal)?

Runs, in ONE process with a one-rank RCCL group (as tests/tools/ddp_rehearsal.py did): plain batched trainer (R0), the same
again (R0b), exact mode B (R1), and plain once more (R0c) -- each two steps on the same two images -- and records for every
step of every run: sha256 of the model output, the task loss and the regulariser sum as float32 bit patterns, the loss; at
the end the parameter differences.  A marker kernel (torch.cumsum over 7777 floats) is launched before each run so that a
rocprofv3 kernel trace of this process can be cut into the runs (tools/r04_rehearsal_diag_kernels.py).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/diag -- python3 tools/r04_rehearsal_diag.py 29551
"""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29551")
os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
torch.cuda.set_device(0)

from learned_quantization_amd.losses import sparse_categorical_crossentropy  # noqa: E402
from learned_quantization_amd.train import Trainer, synthetic_batch  # noqa: E402

res18 = ("imagenette", "nq", 1e-11, "channelwise", None)
d18 = synthetic_batch("imagenette", 2, dev, torch.Generator(device=dev).manual_seed(1))
marker_src = torch.ones(7777, device=dev)


def bits(t):
    return struct.pack("<f", float(t)).hex()


def run(tag, **kw):
    torch.cuda.synchronize()
    torch.cumsum(marker_src, 0)                       # marker kernel: cuts the kernel trace into runs
    torch.cuda.synchronize()
    tr = Trainer(*res18, device=dev, seed=42, batched=True, **kw)
    rec = []
    orig_loss = tr.loss

    def loss(y, p):
        with torch.no_grad():
            task = sparse_categorical_crossentropy(y, p).mean()
            regs = [float(l.regularization_loss()) for l in tr.regularized]
            rec.append({"out_sha": hashlib.sha256(p.detach().cpu().numpy().tobytes()).hexdigest()[:16], "task_bits": bits(task),
                        "task": float(task), "reg_sum_host_f64": sum(regs), "reg_first_bits": bits(regs[0])})
        total = orig_loss(y, p)
        rec[-1]["total_bits"] = bits(total.detach())
        rec[-1]["total"] = float(total.detach())
        return total
    tr.loss = loss
    for _ in range(2):
        tr.step(*d18)
    torch.cuda.synchronize()
    return rec, {n: p.detach().clone() for n, p in tr.model.named_parameters()}


def mean_diff(a, b):
    return sum(float((a[k] - b[k]).abs().sum()) for k in a) / sum(a[k].numel() for k in a)


def max_diff(a, b):
    return max(float((a[k] - b[k]).abs().max()) for k in a)


out = {"torch": torch.__version__, "benchmark": torch.backends.cudnn.benchmark, "deterministic": torch.backends.cudnn.deterministic}
r0, p0 = run("R0_plain")
r0b, p0b = run("R0b_plain")
r1, p1 = run("R1_modeB", ddp_mode="B", force_collectives=True)
r0c, p0c = run("R0c_plain")
out["steps"] = {"R0_plain": r0, "R0b_plain": r0b, "R1_modeB": r1, "R0c_plain": r0c}
out["param_mean_diff"] = {"R0b_vs_R0": mean_diff(p0b, p0), "R1_vs_R0": mean_diff(p1, p0), "R0c_vs_R0": mean_diff(p0c, p0), "R0c_vs_R1": mean_diff(p0c, p1)}
out["param_max_diff"] = {"R0b_vs_R0": max_diff(p0b, p0), "R1_vs_R0": max_diff(p1, p0), "R0c_vs_R0": max_diff(p0c, p0), "R0c_vs_R1": max_diff(p0c, p1)}
print("DIAG " + json.dumps(out))
dist.destroy_process_group()
