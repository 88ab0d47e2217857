#!/usr/bin/env python3
"""One-rank RCCL rehearsal of the data-parallel step on ONE GPU (run as its own process by tests/test_gpu_ddp.py: the
process group is initialised before anything touches the GPU).  With one rank ReduceOp.AVG is the identity, so every
data-parallel variant must reproduce the plain single-process trainer -- while exercising backend "nccl" (= RCCL),
`device_id=` init, the async hook path of DataParallel (tiny buckets), mode B through the batch, and the two graphed
forms of the step (graph / eager all-reduce / graph, and ONE graph containing the RCCL all-reduce).
Prints one JSON object {variant: {"max_param_diff": ..., "losses": [...]}, ...}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29541")
os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)          # before any GPU call of this process
torch.cuda.set_device(0)

from learned_quantization_amd.train import Trainer, synthetic_batch  # noqa: E402

STEPS = 4
ONLY_CAPTURED_COLLECTIVES = len(sys.argv) > 2 and sys.argv[2] == "collectives"
x, y = synthetic_batch("mnist", 32, dev, torch.Generator(device=dev).manual_seed(0))


def run(graph=False, steps=STEPS, **kw):
    tr = Trainer("mnist", "nq", 2e-4, "rowwise", None, device=dev, seed=42, graph=graph, **kw)
    tr.model.eval()                                      # no dropout in the dense model anyway; keeps BN-free determinism explicit
    step = tr.step_graphed if graph else tr.step
    losses = [float(step(x, y).detach()) for _ in range(steps)]
    torch.cuda.synchronize()
    return tr, losses, {n: p.detach().clone() for n, p in tr.model.named_parameters()}


def diff(a, b):
    return max(float((a[k] - b[k]).abs().max()) for k in a)


out = {}
if not ONLY_CAPTURED_COLLECTIVES:
    _, ref_losses, ref = run()                           # plain single-process trainer (no DataParallel)
    _, refb_losses, refb = run(batched=True)
variants = {} if ONLY_CAPTURED_COLLECTIVES else {
    "A_hooks_tiny_buckets": dict(ddp_mode="A", bucket_mb=1e-3, force_collectives=True),
    "A_no_overlap": dict(ddp_mode="A", overlap=False, force_collectives=True),
    "A_batched": dict(ddp_mode="A", batched=True, force_collectives=True),
    "B": dict(ddp_mode="B", bucket_mb=0.05, force_collectives=True),
    "B_batched": dict(ddp_mode="B", batched=True, force_collectives=True),
}
for name, kw in variants.items():
    tr, losses, params = run(**kw)
    assert tr.dp is not None and tr.dp._collectives and tr.dp._avg is not None, "RCCL collectives must run in this rehearsal"
    if name == "A_hooks_tiny_buckets":
        assert tr.dp.overlap and len(tr.dp._ranges) > 2, "the async hook path needs several sub-buckets"
    base = refb if kw.get("batched") else ref
    out[name] = {"max_param_diff": diff(params, base), "losses": losses, "buckets": len(tr.dp._ranges)}
# graphed steps against their eager data-parallel counterparts
graphed = {"graph_split_A_batched": dict(ddp_mode="A", batched=True, force_collectives=True),
           "graph_split_B_batched": dict(ddp_mode="B", batched=True, force_collectives=True),
           "graph_split_A": dict(ddp_mode="A", force_collectives=True)}
if ONLY_CAPTURED_COLLECTIVES:      # its own process: a collective that cannot be captured on this stack aborts the process group
    graphed = {"graph_collectives_A_batched": dict(ddp_mode="A", batched=True, force_collectives=True, graph_collectives=True),
               "graph_collectives_B_batched": dict(ddp_mode="B", batched=True, force_collectives=True, graph_collectives=True)}
for name, kw in graphed.items():
    try:
        tr, losses, params = run(graph=True, **kw)
        ek = {k: v for k, v in kw.items() if k != "graph_collectives"}
        _, el, ep = run(steps=STEPS + 3, **ek)           # step_graphed runs 3 eager warm-up steps before it captures
        el = el[3:]
        # the graphed optimizers keep the step counter on the device: same arithmetic, compared at float32 resolution
        rels = {k: float((params[k] - ep[k]).abs().max() / ep[k].abs().max()) for k in params}     # per tensor, against its magnitude
        worst = max(rels, key=rels.get)
        out[name] = {"max_rel_param_diff_vs_eager": rels[worst], "worst_tensor": worst, "losses": losses, "eager_losses": el,
                     "graphs": 1 + (tr.graph_update is not None)}
    except Exception as e:                               # reported, judged by the test
        out[name] = {"error": repr(e)[:300]}
print("REHEARSAL " + json.dumps(out))
dist.destroy_process_group()
