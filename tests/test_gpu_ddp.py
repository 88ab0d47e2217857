"""Two-process data-parallel training on ONE MI355X (both ranks use cuda:0, gloo transport): the DataParallel bucket,
modes A/B and the batched path run with the real HIP kernels.  (RCCL needs one GPU per rank; the 8-GPU run is the
driver's.)  Mode B must reproduce the single-process global-batch result; mode A the mean of the local scale grads."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data(dev):
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(16, 1, 28, 28, generator=g) * 255.0).to(dev)
    y = torch.randint(0, 10, (16,), generator=g).to(dev)
    return x, y


def _worker(rank, world, port, mode, batched, out_dir, graph=False, steps=2):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from learned_quantization_amd.train import Trainer
    tr = Trainer("mnist", "nq", 2e-4, "rowwise", None, device=dev, ddp_mode=mode, batched=batched, seed=42 + 7 * rank, graph=graph)
    x, y = _data(dev)
    xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]
    step = tr.step_graphed if graph else tr.step
    losses = [float(step(xs, ys).detach()) for _ in range(steps)]
    torch.cuda.synchronize()
    res = {n: p.detach().cpu().clone() for n, p in tr.model.named_parameters()}
    res["losses"] = losses
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,batched", [("A", False), ("B", False), ("A", True), ("B", True)])
def test_two_rank_training_on_one_gpu(tmp_path, mode, batched):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, batched, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for k in r0:
        if k != "losses":
            assert torch.equal(r0[k], r1[k]), f"replicas diverged at {k} (mode {mode})"   # same init (broadcast) + same update
    if mode == "B":
        # exact mode == single process on the global batch
        sys.path.insert(0, ROOT)
        from learned_quantization_amd.train import Trainer
        dev = torch.device("cuda:0")
        tr = Trainer("mnist", "nq", 2e-4, "rowwise", None, device=dev, seed=42)       # rank 0's seed: the broadcast source
        x, y = _data(dev)
        for _ in range(2):
            tr.step(x, y)
        for n, p in tr.model.named_parameters():
            np.testing.assert_allclose(r0[n].numpy(), p.detach().cpu().numpy(), rtol=2e-5, atol=1e-9, err_msg=n)


@pytest.mark.parametrize("mode", ["A", "B"])
def test_two_rank_graphed_step_equals_eager(tmp_path, mode):
    """world_size 2 (gloo transport, both ranks on the one GPU): graph(backward) -> eager bucketed all-reduce -> graph(update)
    gives the parameters of the eager data-parallel step, bit for bit, on both ranks."""
    runs = {}
    for tag, graph, steps in (("eager", False, 5), ("graph", True, 2)):        # step_graphed: 3 eager warm-up steps, then replays
        d = tmp_path / tag
        d.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), mode, True, str(d), graph, steps), nprocs=2, join=True)
        runs[tag] = (torch.load(d / "r0.pt"), torch.load(d / "r1.pt"))
    for k in runs["eager"][0]:
        if k == "losses":
            continue
        assert torch.equal(runs["graph"][0][k], runs["graph"][1][k]), f"graphed replicas diverged at {k}"
        assert torch.equal(runs["graph"][0][k], runs["eager"][0][k]), f"graphed != eager at {k} (mode {mode})"


def _hold_worker(rank, world, port, overlap, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    import learned_quantization_amd as lq
    from learned_quantization_amd.ddp import DataParallel
    lq.reset_layer_names()
    model = lq.build_model("imagenette", mode="nq", value=1e-11, seed=42, orientation="channelwise", device=dev)
    layers = lq.custom_layers_of(model)
    # exact mode B (scales stay outside the bucket) with the regularisers INSIDE the differentiated objective: the hooks of the
    # regularised kernels and biases fire during backward, and a sub-bucket can consist of nothing but such parameters
    dp = DataParallel(model, mode="B", bucket_mb=1e-3, overlap=overlap)      # ~ one sub-bucket per parameter: the big kernels are alone in theirs
    batch = lq.FakeQuantBatch(model, hwio_out=False, autograd=False)
    dp.attach_batch(batch)
    assert len(dp._ranges) > 8 and dp._hold
    if os.environ.get("LQ_TEST_MUTATE_NO_HOLD") == "1":               # mutation check of this test itself (tools/r04_job14.sh): must FAIL
        dp._hold = set()
    g = torch.Generator().manual_seed(100 + rank)                      # every rank its own upstream gradients
    dp.zero_grad()
    batch.quantize_all()
    total = None
    for l in layers:
        w, qb = l.quantized_parameters()
        for o in (w, qb):
            if o is not None:
                t = (o * (torch.randn(tuple(o.shape), generator=g) * 1e-3).to(dev)).sum()
                total = t if total is None else total + t
        if l.regularizer is not None:
            total = total + l.regularization_loss()
    total.backward()
    batch.finish_backward()
    dp.exchange()
    torch.cuda.synchronize()
    torch.save({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None and not getattr(p, "lq_is_scale", False)},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_regularised_kernels_are_exchanged_after_finish_backward(tmp_path):
    """Leaf mode (FakeQuantBatch(autograd=False)): dP reaches the quantised parameters in finish_backward(), after loss.backward() has
    returned -- but the hook of an l2-regularised kernel fires during backward, for the regulariser's gradient alone.  A sub-bucket
    that holds such a kernel must not be exchanged from that hook (DataParallel.attach_batch: _hold).  Two ranks with DIFFERENT
    upstream gradients on the ResNet-18-like net, exact mode B with the regularisers inside the objective, tiny sub-buckets: the
    overlapping exchange must give the averaged gradients of the exchange after backward, bit for bit, on both ranks.  (A one-rank
    group cannot see this: its all-reduce is the identity.  Checked once with the hold removed, LQ_TEST_MUTATE_NO_HOLD=1: fails.)"""
    res = {}
    for overlap in (False, True):
        d = tmp_path / str(overlap)
        d.mkdir()
        mp.spawn(_hold_worker, args=(2, _free_port(), overlap, str(d)), nprocs=2, join=True)
        res[overlap] = (torch.load(d / "r0.pt"), torch.load(d / "r1.pt"))
    assert len(res[True][0]) > 40
    for k in res[True][0]:
        assert torch.equal(res[True][0][k], res[True][1][k]), f"ranks hold different gradients for {k}: a sub-bucket was exchanged before dP arrived"
        assert torch.equal(res[True][0][k], res[False][0][k]), f"overlapping exchange != exchange after backward at {k}"


def _run_script(args, timeout=600):
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return res.stdout


def test_one_rank_rccl_rehearsal():
    """backend "nccl" (= RCCL) with `device_id=` init on the one GPU of the box: DataParallel's async hook path, modes A/B,
    the batch, and both graphed forms of the step -- all must reproduce the plain trainer (ReduceOp.AVG over one rank is
    the identity).  tests/tools/ddp_rehearsal.py runs in its own process so that the process group exists before any GPU call."""
    import json
    out = _run_script([os.path.join(ROOT, "tests", "tools", "ddp_rehearsal.py"), str(_free_port())])
    line = [l for l in out.splitlines() if l.startswith("REHEARSAL ")][-1]
    res = json.loads(line[len("REHEARSAL "):])
    for name in ("A_hooks_tiny_buckets", "A_no_overlap", "A_batched", "B", "B_batched"):
        assert res[name]["max_param_diff"] == 0.0, (name, res[name])
        assert all(np.isfinite(l) for l in res[name]["losses"])
    assert res["A_hooks_tiny_buckets"]["buckets"] > 2
    # gradients that are not dy (loss term, regularisers) in exact mode B: applied after the recompute
    assert res["nqcl_vs_nq_scale_diff"] > 0.0, "the loss term must change the scales for this comparison to mean anything"
    for name in ("nqcl_B_batched", "nqcl_A_batched"):
        assert res[name]["max_param_diff"] == 0.0, (name, res[name])
        assert res[name]["losses"] == res[name]["ref_losses"], (name, res[name])
    # Exact mode B on the l2-regularised ResNet-18-like net (11.2 M parameters).  The contract -- ds is recomputed from the pure
    # task-loss P.grad, regulariser gradients are added afterwards (NQ-L:116-118, 327) -- is asserted BIT FOR BIT on
    # injected upstream gradients (tests/_linear_task.py: no convolution library between the objective and the parameters) ...
    assert res["linear_task_regularizers_move_the_parameters"] > 0.0, "the regularisers must matter for this comparison to mean anything"
    for name in ("linear_task_B_batched_regularized_resnet18", "linear_task_A_batched_regularized_resnet18",
                 "linear_task_B_per_tensor_regularized_resnet18"):
        assert res[name]["max_param_diff"] == 0.0, (name, res[name])
        assert res[name]["losses"] == res[name]["ref_losses"], (name, res[name])
    # ... and THROUGH MIOpen only as a sanity check of the losses, with an absolute bound.  No parameter comparison: MIOpen's
    # weight gradients reduce with atomics, its solver choice depends on the process's history, and Adam turns an ulp of a
    # gradient into a whole step (GPUTEST_r03; DESIGN section 4 "what the red test of round 3 was")
    r18 = res["miopen_B_batched_regularized_resnet18"]
    assert all(np.isfinite(l) for l in r18["losses"] + r18["ref_losses"])
    np.testing.assert_allclose(r18["losses"], r18["ref_losses"], rtol=1e-3)
    for name in ("graph_split_A_batched", "graph_split_B_batched", "graph_split_A"):
        assert "error" not in res[name], (name, res[name])
        assert res[name]["graphs"] == 2 and res[name]["max_rel_param_diff_vs_eager"] < 1e-5, (name, res[name])
        np.testing.assert_allclose(res[name]["losses"], res[name]["eager_losses"], rtol=1e-5)


def test_one_rank_rccl_all_reduce_captured_inside_the_step_graph():
    """``graph_collectives=True``: ONE hipGraph holding forward, backward, the RCCL all-reduce and both optimizers.  The
    capability is established on this torch/RCCL stack (GPUTEST_r02: passed), so every failure of the child -- a non-zero exit,
    a signal, a timeout, a missing result line, an ``error`` entry -- fails the test with the child's stderr."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "ddp_rehearsal.py"), str(_free_port()), "collectives"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    lines = [l for l in res.stdout.splitlines() if l.startswith("REHEARSAL ")]
    assert res.returncode == 0 and lines, f"rc={res.returncode}\n--- stderr\n{res.stderr[-4000:]}\n--- stdout\n{res.stdout[-2000:]}"
    out = json.loads(lines[-1][len("REHEARSAL "):])
    for name in ("graph_collectives_A_batched", "graph_collectives_B_batched"):
        assert "error" not in out[name], (name, out[name])
        assert out[name]["graphs"] == 1 and out[name]["max_rel_param_diff_vs_eager"] < 1e-5, (name, out[name])


def test_bench_force_dist_one_rank_rccl():
    """bench.py with torch.distributed initialised on RCCL (one rank): the scale-gradient all-reduce (ReduceOp.AVG) runs every step."""
    import json
    out = _run_script([os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"])
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e5 and line["roofline"]["frac"] > 0.5


def test_nqcl_unbatched_mode_b_is_refused():
    """Mode B reads P.grad as the global-batch dy; the per-tensor autograd path adds the loss term's gradient to it first."""
    import torch.distributed as dist
    from learned_quantization_amd.train import Trainer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        with pytest.raises(ValueError, match="batched=True"):
            Trainer("mnist", "nqcl", (2e-4, 1e-7), "rowwise", "maxbin", device=torch.device("cuda:0"), ddp_mode="B", force_collectives=True)
    finally:
        dist.destroy_process_group()


def test_bench_exchange_captured_in_the_step_graph_costs_a_few_percent():
    """bench.py --force-dist (default exchange: the RCCL all-reduce captured in the chain of the step graph) against bench.py
    --graph (the same graphs without any collective), one-rank communicator: 2.5 % measured (profiles/r03/exchange/); the
    eager sync exchange costs 10 %.  The structural part (the default IS the captured form) is asserted exactly; the ratio is
    printed and bounded loosely."""
    import json
    common = ["--steps", "160", "--warmup", "32", "--no-cpu-baseline", "--no-extras"]
    def run(extra):
        out = _run_script([os.path.join(ROOT, "bench.py")] + extra + common)
        return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    plain = run(["--graph"])
    captured = run(["--force-dist"])
    assert "hipGraph" in captured["config"]["launch"] and "captured" in captured["config"]["exchange"], captured["config"]
    ratio = captured["ms_per_step"] / plain["ms_per_step"]
    print(f"captured exchange / collective-free graph = {ratio:.4f} ({captured['ms_per_step']:.5f} / {plain['ms_per_step']:.5f} ms per step)")
    # a wall-clock ratio of two separate processes: measured +0.8-2.5 %; profiles/README.md records +-4 % between runs of the same
    # command, so the bound is the measurement plus three times that spread.  What it still catches: the eager forms creeping back
    # (sync +8-10 % is inside the bound on a bad day, async +27 % and a forked branch +14-20 % are not)
    assert ratio < 1.15, (ratio, captured["ms_per_step"], plain["ms_per_step"])


@pytest.mark.parametrize("mode,loss", [("nqcl", "maxbin"), ("cl", "difference")])
def test_one_rank_rccl_penalty_injection_on_bucket_views(mode, loss):
    """The batched penalty kernels write straight into the data-parallel bucket's gradient views: every view must satisfy the
    C ABI's 16-byte alignment (one-element scales sit between the tensors), eager and graphed."""
    import json
    for extra in ([], ["--graph"]):
        out = _run_script(["-m", "learned_quantization_amd.train", "--config", "cifar", "--mode", mode, "--loss", loss, "--value", "1e-11",
                           "--rate", "1e-7", "--batch", "32", "--steps", "4", "--warmup", "4", "--batched", "--force-dist"] + extra)
        line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        assert line["backend"] == "nccl" and np.isfinite(line["final_loss"])


def test_bench_two_ranks_through_the_driver_launcher():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` -- the launcher line the driver uses for N > 1 --
    with both ranks on the one GPU of the box (gloo transport: RCCL needs a GPU per rank): RANK / LOCAL_RANK / WORLD_SIZE from the
    environment, barrier + max-over-ranks timing, ONE JSON line from rank 0 whose value counts both ranks' images."""
    import json
    out = _run_script(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                       "--backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--no-extras"])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 512 and line["config"]["parallelism"] == "dp2"
    assert abs(line["value"] - 512 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    assert line["cpu_baseline"] is None                      # timed on rank 0 at N = 1 only
