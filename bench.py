#!/usr/bin/env python3
"""bench.py -- images/sec of the fake-quant training path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload: synthetic conv-activation batch
256 x 3 x 224 x 224 fp32 per GPU (x ~ U[0,255), dy ~ N(0,1e-3), seed 42), per-channel scales
s = [0.5, 1, 2] (outer=256, G=3, inner=50176), threshold lambda = 1e-11 -- the best threshold the
reference publishes for CIFAR-10 (thesis chapter4.tex:225-226), the dataset BASELINE.json's metric names.
("extras" in the JSON line also reports lambda = 1e-3, where every element takes the exact-ratio +
tanh branch, the single-pass fused kernel K4, and the per-tensor scale variant.)
One STEP = what one training iteration does to that tensor on the hot path, through the C ABI:
    forward   lq_fq_forward      out = floor(x/s)*s                    (K1, 8 B/element)
    backward  lq_fq_scale_grad   ds = mean_g(vote) * max_g|q|          (K2+K3, 8 B/element; dP aliases dy)
  = 16 algorithmic bytes per element, 616.6 MB per step.  Inputs are resident in HBM before the
timed region; >= 4 buffer sets (>= 1.8 GB) rotate so the 256 MiB Infinity Cache cannot serve re-reads.
With N > 1 ranks (one process per GPU, torch.distributed "nccl" = RCCL over xGMI) every rank
processes its own 256-image shard (weak scaling) and the learned-scale gradient is all-reduced
each step -- the only exchange this path has.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline":     K1, K2 and K4 each timed live with HIP events on the launch stream in a dedicated region right after
                  the timed loop: a FIXED number of isolated launches per kernel (independent of --steps), the
                  events stamped by the kernel dispatch itself (lq_profile_events -> hipExtLaunchKernelGGL), so
                  the interval is the kernel's own duration as rocprofv3 reports it.  "kernel" names the one with
                  the largest total per step.  "traffic" = HBM bytes per launch from the committed rocprofv3 PMC passes
                  (profiles/traffic.json), reported only while the kernel sources still hash to what was profiled
  "cpu_baseline": the op-for-op torch-CPU restatement of the reference path (oracle/lq_oracle_torch.py)
                  timed on this box's host cores on a bounded sample -- a baseline, not a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BATCH, CH, H, W = 256, 3, 224, 224
ELEMS = BATCH * CH * H * W                      # 38,535,168
BYTES_FWD = 8 * ELEMS                           # 4 R + 4 W
BYTES_BWD = 8 * ELEMS                           # 4 R (dy) + 4 R (x recompute); dP aliases dy
BYTES_FUSED = 12 * ELEMS                        # K4 single pass: x and dy read once, out written
HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--lam", type=float, default=1e-11)
    ap.add_argument("--variant", choices=["split", "fused"], default="split",
                    help="split = K1 then K2+K3 (what autograd runs); fused = K4 single pass")
    ap.add_argument("--scale", choices=["per_channel", "per_tensor"], default="per_channel")
    ap.add_argument("--sets", type=int, default=4, help="rotating buffer sets (>= 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=0, help="(ignored: the roofline leg now runs after the timed loop)")
    ap.add_argument("--graph", action="store_true", help="replay the steps from captured hipGraphs of --graph-steps steps each "
                    "(with torch.distributed: the scale-gradient all-reduce sits on a forked branch of the graph)")
    ap.add_argument("--graph-edges", choices=["fork_join", "fork_only", "linear"], default="linear",
                    help="captured exchange: linear = the all-reduce in the compute chain of the graph (default: no cross-branch edge, no "
                         "event traffic; +2.5 us per step on a one-rank communicator); fork_join = all-reduce on a side branch joined before "
                         "its buffer is rewritten two steps later (+17 us per step: a cross-branch edge of a hipGraph costs as much as the "
                         "eager event pair); fork_only = one gradient buffer per step, side branches joined once at the end of the graph "
                         "(+20 us).  profiles/r03/exchange/")
    ap.add_argument("--graph-steps", type=int, default=8, help="steps per captured graph (a multiple of the buffer sets keeps the rotation)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the informational extra measurements (N=1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; nccl = RCCL over xGMI (default). "
                    "'gloo' + --share-gpu rehearses the N>1 code path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--exchange", choices=["auto", "graph", "async", "sync"], default="auto",
                    help="scale-gradient all-reduce: graph (captured on a forked branch of the step graph: cross-stream edges cost no "
                         "events; the default on RCCL, implies --graph), sync (eager, on the compute stream: ~10 us/step measured on a "
                         "one-rank communicator; the fallback when the capture fails, and the default on other backends) or async "
                         "(eager, RCCL side stream + events: ~27 us/step)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (exercises the RCCL path)")
    return ap.parse_args()


def cpu_baseline(lam: float, budget_s: float):
    """Reference-equivalent CPU path (restated; TF 2.11 unavailable): unfused op sequence in torch-CPU."""
    from oracle import lq_oracle_torch as OT
    n_img = 32
    # a PINNED sample (129 / 409 / 471 images/s on three boxes with torch's default of one thread per visible
    # core, 128, on a 16-core share of the host): a fixed thread count within the one-GPU job's CPU share, one warm pass discarded
    prev_threads = torch.get_num_threads()
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(42)
    x = torch.rand(n_img, CH, H, W, generator=g) * 255.0
    dy = torch.randn(n_img, CH, H, W, generator=g) * 1e-3
    s = torch.tensor([0.5, 1.0, 2.0]).view(1, CH, 1, 1)
    OT.nq_forward_backward(x, s, lam, dy)                       # warm
    # a sample bounded in TIME (a rep count derived from one timed call ran 58 s on a loaded box: the first call after the
    # warm-up is not representative of the sustained rate)
    reps = 0
    t0 = time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < budget_s and reps < 400):
        OT.nq_forward_backward(x, s, lam, dy)
        reps += 1
    dt = time.perf_counter() - t0
    torch.set_num_threads(prev_threads)
    return {
        "value": n_img * reps / dt,
        "unit": "images/s",
        "label": f"images/s on {threads} torch CPU threads (pinned with torch.set_num_threads; os.cpu_count() = {os.cpu_count()}), "
                 "one warm pass discarded",
        "cores": threads,
        "kind": "port",
        "sample": f"{reps} x (fwd+bwd of {n_img}x3x224x224 fp32, lambda={lam:g}) = {dt:.1f} s after one discarded warm pass; "
                  f"oracle/lq_oracle_torch.py unfused op sequence; {threads} threads; os.cpu_count()={os.cpu_count()}",
    }


def csrc_sha() -> str:
    """sha256 over the kernel sources: ties profiles/traffic.json to the code it was collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "learned_quantization_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "lq_hip.h")]):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    # read by the HSA runtime when it initialises (the first torch.cuda call below); the host driver only supports dmabuf IPC
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    if args.share_gpu:
        if args.backend == "nccl":
            raise SystemExit("--share-gpu is a rehearsal mode and needs --backend gloo (RCCL cannot put two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # a bounded collective timeout: ranks that ever disagree about the sequence of collectives abort within minutes with
        # RCCL's own message instead of hanging until the driver's limit
        tmo = datetime.timedelta(minutes=5)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    from learned_quantization_amd import _hip
    lib = _hip.load()

    # ---------------- inputs resident in HBM before the timed region
    nsets = max(4, args.sets)
    g = torch.Generator(device=dev).manual_seed(42 + rank)
    xs = [torch.rand(BATCH, CH, H, W, device=dev, generator=g) * 255.0 for _ in range(nsets)]
    dys = [torch.randn(BATCH, CH, H, W, device=dev, generator=g) * 1e-3 for _ in range(nsets)]
    outs = [torch.empty(BATCH, CH, H, W, device=dev) for _ in range(nsets)]
    if args.scale == "per_channel":
        s = torch.tensor([0.5, 1.0, 2.0], device=dev).view(1, CH, 1, 1)
        outer, G, inner = BATCH, CH, H * W
    else:
        s = torch.tensor([1.0], device=dev)
        outer, G, inner = 1, 1, ELEMS
    # two gradient buffers: the (tiny) scale-gradient all-reduce of step i runs asynchronously on RCCL's
    # stream and overlaps the kernels of step i+1; a buffer is reused only after its collective completed
    dss = [torch.zeros_like(s) for _ in range(max(2, args.graph_steps if args.graph_edges == "fork_only" else 2))]
    ds = dss[0]
    pending = [None, None]
    ws_bytes = lib.lq_workspace_bytes(outer, G, inner)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream or None
    lam = float(args.lam)
    px = [t.data_ptr() for t in xs]
    pdy = [t.data_ptr() for t in dys]
    pout = [t.data_ptr() for t in outs]
    ps, pws = s.data_ptr(), ws.data_ptr()
    pdss = [t.data_ptr() for t in dss]
    pds = pdss[0]

    fwd, bwd, fused = lib.lq_fq_forward, lib.lq_fq_scale_grad, lib.lq_fq_fwd_bwd_fused

    avg_op = dist.ReduceOp.AVG if (use_dist and args.backend == "nccl") else None

    def exchange(i):
        """learned-scale gradient exchange (mode A): mean over ranks, asynchronous."""
        b = i & 1
        if exchange_form != "async":
            if avg_op is not None:
                dist.all_reduce(dss[b], op=avg_op)
            else:
                dss[b].div_(world)
                dist.all_reduce(dss[b], op=dist.ReduceOp.SUM)
        elif avg_op is not None:     # RCCL averages in the collective: no extra kernel on the compute stream
            pending[b] = dist.all_reduce(dss[b], op=avg_op, async_op=True)
        else:
            dss[b].div_(world)
            pending[b] = dist.all_reduce(dss[b], op=dist.ReduceOp.SUM, async_op=True)

    def step(i, ev=None):
        k = i % nsets
        b = i & 1
        if use_dist and pending[b] is not None:
            pending[b].wait()          # stream-level dependency: the buffer's previous collective is done
            pending[b] = None
        if args.variant == "split":
            if ev:
                ev[0].record(stream)
            rc = fwd(px[k], ps, pout[k], None, 0, outer, G, inner, sp)
            if ev:
                ev[1].record(stream)
            rc |= bwd(px[k], ps, pdy[k], lam, pdss[b], None, pws, ws_bytes, outer, G, inner, sp)
            if ev:
                ev[2].record(stream)
        else:
            if ev:
                ev[0].record(stream)
            rc = fused(px[k], ps, pdy[k], lam, pout[k], pdss[b], pws, ws_bytes, outer, G, inner, sp)
            if ev:
                ev[2].record(stream)
        if rc:
            _hip.check(rc, "bench step")
        if use_dist:
            exchange(i)

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- the graphed form.  A captured graph holds S consecutive steps, each followed by the all-reduce of its scale gradient
    # (custom_layers.py:116-118) in the SAME chain: captured, the collective is RCCL's kernel alone -- none of
    # the event records ProcessGroupNCCL puts around an eager call (eager sync: +10 us per step on a one-rank communicator,
    # eager async: +27 us; captured: +2.5 us, the one-rank reduce kernel itself).  Putting the collective on a forked
    # branch of the graph (--graph-edges fork_join / fork_only) would hide its latency at N > 1 but costs 17-20 us per
    # step here: hipGraph executes parallel branches on separate queues and every cross-branch edge is a signal between
    # them (measured, profiles/r03/exchange/).
    exchange_form = args.exchange
    if exchange_form == "auto":
        exchange_form = "graph" if (use_dist and args.backend == "nccl") else "sync"
    if not use_dist:
        exchange_form = None
    want_graph = args.graph or exchange_form == "graph"
    if exchange_form == "graph" and args.backend != "nccl":
        raise SystemExit("--exchange graph needs --backend nccl (only RCCL collectives can be stream-captured)")
    graph_note = None

    def capture(S):
        gph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(dev)
        sides = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        # a process group's watchdog THREAD polls events while this thread captures: "thread_local" keeps that legal
        with torch.cuda.graph(gph, stream=cap, capture_error_mode="thread_local" if use_dist else "global"):
            main = torch.cuda.current_stream(dev)
            cs = main.cuda_stream or None
            edges = args.graph_edges if use_dist else "linear"
            for j in range(S):
                k = j % nsets
                b = j if edges == "fork_only" else (j & 1)             # fork_only: a gradient buffer per step, no join until the end
                side = sides[j & 1]
                join = use_dist and edges == "fork_join" and j >= 2    # the all-reduce of step j-2 is done with dss[b]
                if args.variant == "split":
                    rc = fwd(px[k], ps, pout[k], None, 0, outer, G, inner, cs)
                    if join:
                        main.wait_stream(side)
                    rc |= bwd(px[k], ps, pdy[k], lam, pdss[b], None, pws, ws_bytes, outer, G, inner, cs)
                else:
                    if join:
                        main.wait_stream(side)
                    rc = fused(px[k], ps, pdy[k], lam, pout[k], pdss[b], pws, ws_bytes, outer, G, inner, cs)
                if rc:
                    _hip.check(rc, "graph capture")
                if use_dist:
                    if edges != "linear":
                        side.wait_stream(main)                         # fork
                    with torch.cuda.stream(side if edges != "linear" else main):
                        if avg_op is not None:
                            dist.all_reduce(dss[b], op=avg_op)
                        else:
                            dss[b].div_(world)
                            dist.all_reduce(dss[b], op=dist.ReduceOp.SUM)
            if use_dist and edges != "linear":
                main.wait_stream(sides[0])
                main.wait_stream(sides[1])
        return gph

    graphs = None
    if want_graph:
        from learned_quantization_amd.ddp import CaptureRefused, capture_with_agreement, control_group
        S = max(1, args.graph_steps)
        ctl = control_group() if use_dist else None        # gloo side group: host-side agreement on the outcome of the capture
        if use_dist:
            # two eager steps first: RCCL sets up its channels, buffers and kernels at the first collective of a communicator
            # (allocations and IPC exchanges that must not happen inside a capture)
            for i in range(2):
                step(i)
            fence()
        torch.cuda.synchronize(dev)
        captured = {}

        def attempt():
            try:
                captured[S] = capture(S)
                for r in {args.warmup % S, args.steps % S} - {0}:
                    captured[r] = capture(r)               # the remainder of a loop whose length is not a multiple of S
                torch.cuda.synchronize(dev)
            except _hip.LQError:
                raise                                      # an error of the path itself, not a refused capture
            except RuntimeError as e:
                raise CaptureRefused(f"{e!r}"[:300]) from e

        def health():
            import datetime
            t = torch.ones(1, device=dev)
            dist.all_reduce(t, async_op=True).wait(timeout=datetime.timedelta(seconds=60))
            torch.cuda.synchronize(dev)
            if int(t.item()) != world:
                raise RuntimeError(f"health check: all-reduce over {world} ranks returned {float(t)}")

        # every rank keeps its graphs, or every rank drops them for the eager sync exchange, or every rank stops: a rank that
        # fell back on its own would issue another sequence of collectives than the others and hang them
        if capture_with_agreement(attempt, ctl, health_check=health if use_dist else None):
            graphs = captured
        else:
            if args.graph or args.exchange == "graph":
                raise SystemExit("the step graph (with its all-reduce) could not be captured on every rank; asked for explicitly")
            graphs = None
            exchange_form = "sync"
            graph_note = "graph capture was refused on at least one rank: eager sync exchange on every rank"
    if exchange_form == "graph" and graphs is None:
        exchange_form = "sync"

    def run_steps(n):
        """n steps of the path: eager launches, or replays of the S-step graph plus one remainder graph."""
        if graphs is None:
            for i in range(n):
                step(i)
            return
        S = max(graphs)
        for _ in range(n // S):
            graphs[S].replay()
        if n % S:
            graphs[n % S].replay()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: every kernel of the path, ROOF_SAMPLES launches each, independent of --steps.  The kernels are
    # launched through the ABI with lq_profile_events(start, stop): hipExtLaunchKernelGGL stamps the two events with the
    # kernel's own begin and end (the duration rocprofv3 reports) -- no event-record cost inside the interval, the finalize
    # launch of the two-launch calls outside it, and no marker packets between the launches: the stream sees the same
    # back-to-back kernel sequence as in the timed loop.
    ROOF_SAMPLES = 64
    Ev = lambda: torch.cuda.Event(enable_timing=True)          # noqa: E731

    def stamped(n):
        evs = [(Ev(), Ev()) for _ in range(n)]
        for a, b in evs:                                       # torch creates the hipEvent_t lazily, at the first record
            a.record(stream)
            b.record(stream)
        return evs

    nroof = ROOF_SAMPLES + 4                                   # the first four warm the path and are dropped
    e1, e2, e4 = stamped(nroof), stamped(nroof), stamped(nroof)
    torch.cuda.synchronize(dev)
    rc = 0
    for j in range(nroof):                                     # the split step as in the timed loop, back to back: K1, K2 (+K3), ...
        k = j % nsets
        lib.lq_profile_events(e1[j][0].cuda_event, e1[j][1].cuda_event)
        rc |= fwd(px[k], ps, pout[k], None, 0, outer, G, inner, sp)
        lib.lq_profile_events(e2[j][0].cuda_event, e2[j][1].cuda_event)
        rc |= bwd(px[k], ps, pdy[k], lam, pds, None, pws, ws_bytes, outer, G, inner, sp)
    for j in range(nroof):                                     # the fused single-pass step, back to back
        k = j % nsets
        lib.lq_profile_events(e4[j][0].cuda_event, e4[j][1].cuda_event)
        rc |= fused(px[k], ps, pdy[k], lam, pout[k], pds, pws, ws_bytes, outer, G, inner, sp)
    lib.lq_profile_events(None, None)
    if rc:
        _hip.check(rc, "roofline leg")
    torch.cuda.synchronize(dev)
    mean_us = lambda prs: sum(a.elapsed_time(b) for a, b in prs[4:]) / len(prs[4:]) * 1e3      # noqa: E731
    t_k1, t_k2, t_k4 = mean_us(e1), mean_us(e2), mean_us(e4)
    kernels = {}
    for name, t_us, nbytes in (("K1 k_flat_fwd<OP_FWD> (lq_fq_forward)", t_k1, BYTES_FWD),
                               ("K2 k_row_stream<OP_BWD> (lq_fq_scale_grad, traversal)", t_k2, BYTES_BWD),
                               ("K4 k_row_stream<OP_FUSED> (lq_fq_fwd_bwd_fused, traversal)", t_k4, BYTES_FUSED)):
        gbs = nbytes / (t_us * 1e-6) / 1e9
        kernels[name] = {"avg_launch_us": t_us, "algorithmic_bytes_per_launch": nbytes, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS}
    if args.variant == "split":
        # the step runs K1, K2, K3: the dominant kernel is the one with the largest total per step
        kname = max(list(kernels)[:2], key=lambda k: kernels[k]["avg_launch_us"])
        step_bytes = BYTES_FWD + BYTES_BWD
    else:
        kname = list(kernels)[2]
        step_bytes = BYTES_FUSED
    kt = kernels[kname]["avg_launch_us"] * 1e-6
    kbytes = kernels[kname]["algorithmic_bytes_per_launch"]
    extra = {"kernels": kernels, "event_samples_per_kernel": ROOF_SAMPLES,
             "method": "mean over back-to-back launches of hipEventElapsedTime(start, stop) with the events stamped by the kernel "
                       "dispatch itself (hipExtLaunchKernelGGL through lq_profile_events); region placed after the timed loop, "
                       "independent of --steps"}

    # ---- informational extras (not part of `value`): other variants of the same step, 100 steps each, no events
    extras = {}
    if world == 1 and not args.no_extras and graphs is None:
        def timed(fn, n=100):
            for i in range(10):
                fn(i)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / n

        def split_step(lm, sc):
            sp_, o_, g_, in_ = sc
            def f(i):
                k = i % nsets
                fwd(px[k], sp_, pout[k], None, 0, o_, g_, in_, sp)
                bwd(px[k], sp_, pdy[k], lm, pds, None, pws, ws_bytes, o_, g_, in_, sp)
            return f

        def fused_step(lm, sc):
            sp_, o_, g_, in_ = sc
            def f(i):
                k = i % nsets
                fused(px[k], sp_, pdy[k], lm, pout[k], pds, pws, ws_bytes, o_, g_, in_, sp)
            return f

        s1 = torch.tensor([1.0], device=dev)
        ws1 = lib.lq_workspace_bytes(1, 1, ELEMS)
        assert ws1 <= ws_bytes or args.scale == "per_tensor"
        pc = (ps, outer, G, inner)
        pt = (s1.data_ptr(), 1, 1, ELEMS)
        for name, fn, nbytes in (
                ("split_lambda_1e-3", split_step(1e-3, pc), BYTES_FWD + BYTES_BWD),
                ("split_lambda_0", split_step(0.0, pc), BYTES_FWD + BYTES_BWD),
                (f"fused_K4_lambda_{lam:g}", fused_step(lam, pc), BYTES_FUSED),
                ("fused_K4_lambda_1e-3", fused_step(1e-3, pc), BYTES_FUSED),
                (f"split_per_tensor_scale_lambda_{lam:g}", split_step(lam, pt), BYTES_FWD + BYTES_BWD)):
            if "per_tensor" in name and ws1 > ws_bytes:
                continue
            dt = timed(fn)
            extras[name] = {"images_per_s": BATCH / dt, "us_per_step": dt * 1e6, "step_GBs": nbytes / dt / 1e9,
                            "algorithmic_bytes_per_step": nbytes}

        # BASELINE.json configs[1] end to end (informational; MIOpen convolutions dominate it and are out of scope):
        # CIFAR-10 small CNN, nested quantization layer, bs 256, multi-tensor batch + whole step in one hipGraph
        try:
            from learned_quantization_amd.train import Trainer, synthetic_batch
            tr = Trainer("cifar", "nq", 1e-11, "channelwise", None, device=dev, seed=42, graph=True, batched=True,
                         log_dir=os.path.join("/tmp", "lq_bench_logs"))
            gb = torch.Generator(device=dev).manual_seed(7)
            bx, by = synthetic_batch("cifar", 256, dev, gb)
            for _ in range(5):
                tr.step_graphed(bx, by)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(30):
                tr.step_graphed(bx, by)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 30
            extras["e2e_cifar_cnn_bs256_nq_channelwise_hipgraph_batched"] = {
                "images_per_s": 256 / dt, "ms_per_step": dt * 1e3, "quantized_elements": 287008,
                "note": "full training step (MIOpen convs + lq kernels + optimizers), synthetic data"}
            del tr
        except Exception as e:      # never let the informational extra break the bench line
            extras["e2e_cifar_cnn_bs256_nq_channelwise_hipgraph_batched"] = {"error": repr(e)[:200]}
        # The per-step fake-quant work of the ResNet weight sets (BASELINE configs[2], [4]; SURVEY 8d: 16 B per element) through
        # the multi-tensor batch ABI: lq_batch_forward + lq_batch_scale_grad_step = three launches for 40 / 108 tensors, conv
        # kernels stored in OIHW order (layers.py kernel_storage).  Same loop as tools/bench_weights.py --kernel-storage oihw.
        try:
            import ctypes as _ct

            import learned_quantization_amd as _lq
            _lib, _sp = _lq._hip.load(), _lq._hip.stream_ptr(dev)
            for _cfg, _val in (("imagenette", 1e-11), ("resnet50", (1e-10, 1e-11))):
                _lq.reset_layer_names()
                _m = _lq.build_model(_cfg, mode="nq", value=_val, seed=42, orientation="channelwise", device=dev)
                _b = _lq.FakeQuantBatch(_m, hwio_out=False)
                _g = torch.Generator(device=dev).manual_seed(42)
                _dys = [torch.empty_like(e.param.data).normal_(generator=_g) * 1e-3 for e in _b.entries]
                _ptrs = (_ct.c_void_p * len(_dys))(*[d.data_ptr() for d in _dys])
                _n_el = sum(e.param.numel() for e in _b.entries)

                def _step():
                    _lib.lq_batch_forward(_b._handle, _sp)
                    _lib.lq_batch_scale_grad_step(_b._handle, _ptrs, 0, _b.ws.data_ptr(), _b.ws.numel(), 1e-4, 0.9, 0.999, 1e-7, 1, None, 0, _sp)
                for _ in range(20):
                    _step()
                torch.cuda.synchronize(dev)
                _t0 = time.perf_counter()
                for _ in range(300):
                    _step()
                torch.cuda.synchronize(dev)
                _dt = (time.perf_counter() - _t0) / 300
                extras[f"weight_set_step_{_cfg}_channelwise_batch_abi"] = {
                    "us_per_step": _dt * 1e6, "tensors": len(_b.entries), "elements": _n_el, "algorithmic_bytes_per_step": 16 * _n_el,
                    "step_GBs": 16 * _n_el / _dt / 1e9, "frac_of_8TBs": 16 * _n_el / _dt / 8e12,
                    "note": "fake-quant forward + scale gradient + scale Adam of every kernel and bias of the model in three launches"}
                del _b, _m, _dys
        except Exception as e:      # informational only
            extras["weight_set_step"] = {"error": repr(e)[:200]}

    # HBM bytes per launch of the dominant kernel from the committed PMC passes -- only while the kernel sources still are
    # the ones that were profiled (tools/prof_pmc.sh stores their hash next to the numbers)
    traffic, traffic_note = None, "profiles/traffic.json absent"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("csrc_sha") != csrc_sha():
                traffic_note = f"profiles/traffic.json was collected on other kernel sources (csrc_sha {tj.get('csrc_sha')} != {csrc_sha()}): not reported"
            else:
                key = "K1" if kname.startswith("K1") else ("K2" if kname.startswith("K2") else "K4")
                traffic = tj.get("hbm_bytes_per_launch", {}).get(key)
                traffic_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of {tj.get('collected_with', 'tools/prof_pmc.sh')}, csrc_sha {tj.get('csrc_sha')}"
        except Exception as e:
            traffic, traffic_note = None, f"profiles/traffic.json unreadable: {e!r}"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * BATCH * args.steps / elapsed
        achieved = kbytes / kt / 1e9
        line = {
            "metric": "images/sec training with fake-quant layer, CIFAR-10 bs=256, 1/2/4/8 MI355X",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"fake-quant fwd+bwd ({args.variant}) on conv-activation batch 256x3x224x224 fp32 per GPU, "
                                   f"{args.scale} scales, lambda={lam:g}",
                       "per_gpu_batch": BATCH, "global_batch": BATCH * world, "parallelism": f"dp{world}",
                       "buffer_sets": nsets, "algorithmic_bytes_per_step": step_bytes,
                       "launch": (f"hipGraph of {max(graphs)} steps" if graphs is not None else "eager"),
                       "exchange": ({"graph": f"RCCL all-reduce of ds captured in the step graph ({args.graph_edges})",
                                     "sync": "eager all-reduce of ds on the compute stream",
                                     "async": "eager all-reduce of ds on RCCL's stream"}.get(exchange_form) if use_dist else None),
                       **({"note": graph_note} if graph_note else {}),
                       "step_GBs": step_bytes / (elapsed / args.steps) / 1e9},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": kbytes, "avg_launch_us": kt * 1e6, **extra,
                         # math-free kernels of the same shape on this device class (tools/membench, re-measured in round 3:
                         # 512 threads, nontemporal): what a streaming kernel can reach of the 8 TB/s spec peak.  Box to box the
                         # figures move by 2-3 % (round 1: 6579 / 6808 / 6507)
                         "stream_ceiling_GBs": {"read1_write1": 6637, "read2": 6689, "read2_write1": 6385,
                                                "source": "profiles/r03/membench.txt"}},
        }
        if world == 1 and not args.no_extras and graphs is None:
            line["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(lam, args.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
