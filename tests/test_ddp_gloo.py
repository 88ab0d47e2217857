"""World-size-2 gloo tests of the data-parallel exchange step (CPU; the kernel is replaced by the ORACLE,
in the test only, so that the bucket / ordering logic of learned_quantization_amd/ddp.py is exercised)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeNested(torch.nn.Module):
    def __init__(self, shape, lam):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.full(shape, 0.01))
        self.scale.lq_is_scale = True
        self.penalty_threshold = lam


class _FakeLayer(torch.nn.Module):
    """Same attribute surface as CustomDenseLayer (W, b, nested_q_w_layer, nested_q_b_layer); the fake-quant op is
    replaced by an oracle-backed autograd function so the test runs without a GPU."""

    def __init__(self, lam, permuted=False):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        w = torch.randn(12, 5, generator=g) * 0.05
        if permuted:            # same shape and values, memory in (5, 12) order: what kernel_storage="oihw" does to a conv kernel
            w = w.t().contiguous().t()
        self.W = torch.nn.Parameter(w)
        self.b = torch.nn.Parameter(torch.randn(5, generator=g) * 0.05)
        self.nested_q_w_layer = _FakeNested((12, 1), lam)
        self.nested_q_b_layer = _FakeNested((1,), lam)

    def forward(self, x):
        return x @ _OracleNQ.apply(self.W, self.nested_q_w_layer.scale, self.nested_q_w_layer.penalty_threshold) + \
            _OracleNQ.apply(self.b, self.nested_q_b_layer.scale, self.nested_q_b_layer.penalty_threshold)


class _OracleNQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, s, lam):
        from oracle import lq_oracle as O
        ctx.save_for_backward(P, s)
        ctx.lam = lam
        return torch.from_numpy(O.fq_forward(P.detach().numpy(), s.detach().numpy())[1])

    @staticmethod
    def backward(ctx, dy):
        from oracle import lq_oracle as O
        P, s = ctx.saved_tensors
        _, ds = O.nq_backward(P.detach().numpy(), s.detach().numpy(), ctx.lam, dy.numpy())
        return dy, torch.from_numpy(ds), None


def _oracle_scale_grad(P, s, dy, lam):
    from oracle import lq_oracle as O
    return torch.from_numpy(O.nq_backward(P.numpy(), s.numpy(), lam, dy.numpy())[1])


def _worker(rank, world, port, mode, out_dir, permuted=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from learned_quantization_amd.ddp import DataParallel
    lam = 2e-2
    model = _FakeLayer(lam, permuted)
    assert model.W.is_contiguous() != permuted
    if rank == 1:   # replicas start different on purpose: the wrapper must broadcast rank 0's
        with torch.no_grad():
            model.W.add_(1.0)
    # mode A with tiny sub-buckets: every parameter is its own overlapped all-reduce; mode B: one bucket
    dp = DataParallel(model, mode=mode, scale_grad_fn=_oracle_scale_grad, bucket_mb=(1e-5 if mode == "A" else 25.0))
    assert len(dp._ranges) == (4 if mode == "A" else 1)       # views are padded to 64 floats: W, b and both scales
    assert model.W.grad.stride() == model.W.stride(), "the bucket view carries the parameter's strides"
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 12, generator=g)
    Y = torch.randn(8, 5, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    dp.zero_grad()
    loss = ((dp(xs) - ys) ** 2).sum() / 8.0        # global-batch mean, local shard contribution * world below
    (loss * world).backward()                       # so that the rank-mean of gradients equals the global gradient
    dp.sync_gradients()
    res = {n: p.grad.clone() for n, p in model.named_parameters()}
    res["W_param"] = model.W.detach().clone()
    assert model.W.grad.stride() == model.W.stride()
    res = {k: v.contiguous() for k, v in res.items()}
    torch.save(res, os.path.join(out_dir, f"rank{rank}_{mode}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,permuted", [("A", False), ("B", False), ("A", True), ("B", True)])
def test_data_parallel_two_ranks(tmp_path, mode, permuted):
    """``permuted``: W is a dense permuted view (the memory layout of a conv kernel stored in OIHW order): broadcast, bucket
    views and the exchange must treat it index by index like the contiguous parameter."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path), permuted), nprocs=2, join=True)
    r0 = torch.load(tmp_path / f"rank0_{mode}.pt")
    r1 = torch.load(tmp_path / f"rank1_{mode}.pt")
    assert torch.equal(r0["W_param"], r1["W_param"])                 # broadcast happened
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"{k} differs across ranks ({mode})"

    # single-process reference on the full batch
    sys.path.insert(0, ROOT)
    lam = 2e-2
    model = _FakeLayer(lam)
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 12, generator=g)
    Y = torch.randn(8, 5, generator=g)
    (((model(X) - Y) ** 2).sum() / 8.0).backward()
    np.testing.assert_allclose(r0["W"].numpy(), model.W.grad.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["b"].numpy(), model.b.grad.numpy(), rtol=1e-5, atol=1e-7)
    if mode == "B":
        # exact mode: ds equals the single-device global-batch scale gradient
        np.testing.assert_allclose(r0["nested_q_w_layer.scale"].numpy(), model.nested_q_w_layer.scale.grad.numpy(), rtol=1e-5)
        np.testing.assert_allclose(r0["nested_q_b_layer.scale"].numpy(), model.nested_q_b_layer.scale.grad.numpy(), rtol=1e-5)
    else:
        # mode A: mean over ranks of the local scale gradients (non-linear in dy -> generally != global)
        from oracle import lq_oracle as O
        locals_ = []
        for r in range(2):
            m = _FakeLayer(lam)
            xs, ys = X[r * 4:(r + 1) * 4], Y[r * 4:(r + 1) * 4]
            ((((m(xs) - ys) ** 2).sum() / 8.0) * 2).backward()
            locals_.append(m.nested_q_w_layer.scale.grad.numpy())
        np.testing.assert_allclose(r0["nested_q_w_layer.scale"].numpy(), (locals_[0] + locals_[1]) / 2, rtol=1e-5)


def _accum_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from learned_quantization_amd.ddp import DataParallel
    model = _FakeLayer(2e-2)
    dp = DataParallel(model, mode="A", scale_grad_fn=_oracle_scale_grad, bucket_mb=1e-5)
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 5, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    res = {}
    # (1) a second backward into buckets that were already exchanged must raise, not silently diverge
    dp.zero_grad()
    ((dp(xs) - ys) ** 2).sum().backward()
    dp.sync_gradients()
    try:
        ((dp(xs) - ys) ** 2).sum().backward()
        res["second_backward"] = "no error"
    except RuntimeError as e:
        res["second_backward"] = str(e)
    # ... also without overlap (the configuration of the graphed steps): the guard does not depend on the hooks launching anything
    dp2 = DataParallel(_FakeLayer(2e-2), mode="A", scale_grad_fn=_oracle_scale_grad, overlap=False)
    dp2.zero_grad()
    ((dp2(xs) - ys) ** 2).sum().backward()
    dp2.sync_gradients()
    try:
        ((dp2(xs) - ys) ** 2).sum().backward()
        res["second_backward_no_overlap"] = "no error"
    except RuntimeError as e:
        res["second_backward_no_overlap"] = str(e)
    # (2) gradient accumulation: earlier passes under no_sync(), the last one exchanges the accumulated bucket
    dp.zero_grad()
    with dp.no_sync():
        ((dp(xs[:2]) - ys[:2]) ** 2).sum().backward()
    ((dp(xs[2:]) - ys[2:]) ** 2).sum().backward()
    dp.sync_gradients()
    res["W_accum"] = model.W.grad.clone()
    torch.save(res, os.path.join(out_dir, f"acc{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_and_double_backward_guard(tmp_path):
    port = _free_port()
    mp.spawn(_accum_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "acc0.pt"), torch.load(tmp_path / "acc1.pt")
    assert "no_sync" in r0["second_backward"] and "no_sync" in r1["second_backward"]
    assert "no_sync" in r0["second_backward_no_overlap"] and "no_sync" in r1["second_backward_no_overlap"]
    assert torch.equal(r0["W_accum"], r1["W_accum"])
    # reference: mean over ranks of the per-rank summed-loss gradients
    sys.path.insert(0, ROOT)
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 5, generator=g)
    grads = []
    for r in range(2):
        m = _FakeLayer(2e-2)
        xs, ys = X[r * 4:(r + 1) * 4], Y[r * 4:(r + 1) * 4]
        ((m(xs[:2]) - ys[:2]) ** 2).sum().backward()
        ((m(xs[2:]) - ys[2:]) ** 2).sum().backward()
        grads.append(m.W.grad)
    np.testing.assert_allclose(r0["W_accum"].numpy(), ((grads[0] + grads[1]) / 2).numpy(), rtol=1e-5, atol=1e-7)


def test_grad_bucket_single_process():
    sys.path.insert(0, ROOT)
    from learned_quantization_amd.ddp import GradBucket
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    b = GradBucket(ps)
    assert b.offsets == [0, 64] and b.flat.numel() == 128                      # every view starts on a 256-byte boundary
    assert all(v.data_ptr() % 16 == 0 for v in b.views)
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    assert torch.all(b.flat[:12] == 2) and torch.all(b.flat[64:69] == 3)      # autograd accumulated into the bucket
    assert torch.count_nonzero(b.flat) == 17                                   # the padding stays zero
    ps[1].grad = torch.full((5,), 7.0)                                         # someone re-allocated a grad
    b.gather_()
    assert torch.all(b.flat[64:69] == 7) and ps[1].grad.data_ptr() == b.views[1].data_ptr()
    b.zero_()
    assert torch.count_nonzero(b.flat) == 0
    # a dense permuted parameter (conv kernel shaped HWIO, stored OIHW): the view has the parameter's strides, so that gradient
    # and parameter are read with one memory-order descriptor
    k = torch.nn.Parameter(torch.randn(4, 3, 2, 2).permute(2, 3, 1, 0))
    b2 = GradBucket([ps[0], k])
    assert k.grad.shape == k.shape and k.grad.stride() == k.stride() and not k.grad.is_contiguous()
    (k * torch.arange(48.0).view(4, 3, 2, 2).permute(2, 3, 1, 0)).sum().backward()
    assert torch.equal(b2.flat[64:112], torch.arange(48.0)), "bucket memory in the parameter's memory order"


# ---------------------------------------------------------------------------------------------------------------------
# Collective fallback decisions: the outcome of a capture attempt is agreed on by all ranks.

def _agreement_worker(rank, world, port, scenario, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from learned_quantization_amd.ddp import CaptureRefused, DataParallel, capture_with_agreement, control_group
    ctl = control_group()
    model = _FakeLayer(2e-2)
    dp = DataParallel(model, mode="A", force_collectives=True)
    state = {"graph": None}

    def attempt():
        # a "capture": rank 1 fails in the way the scenario says, rank 0 succeeds
        if rank == 1 and scenario == "refused_on_rank_1":
            raise CaptureRefused("operation not permitted when stream is capturing")
        if rank == 1 and scenario == "error_on_rank_1":
            raise ValueError("a shape bug in the update phase")
        state["graph"] = "captured"

    result = {"rank": rank}
    try:
        ok = capture_with_agreement(attempt, ctl, health_check=dp.health_check)
        result["ok"] = ok
        if not ok:
            state["graph"] = None                    # every rank drops what it captured ...
        # ... and every rank issues the SAME sequence of collectives afterwards: one eager exchange of the bucket
        dp.zero_grad()
        x = torch.ones(4, 12) * (rank + 1)
        model(x).sum().backward()
        dp.exchange()
        result["form"] = "graph" if state["graph"] else "eager"
        result["grad_sum"] = float(model.W.grad.sum())
    except BaseException as e:                       # noqa: BLE001
        result["raised"] = type(e).__name__ + ": " + str(e)[:120]
    torch.save(result, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["all_captured", "refused_on_rank_1", "error_on_rank_1"])
def test_capture_outcome_is_agreed_on_by_all_ranks(tmp_path, scenario):
    """A capture that fails on ONE rank must not leave the ranks on different forms of the step (graph replay on one, eager
    collectives on the other: a hang).  Refused on rank 1 -> both ranks end on the eager path and their next exchange completes with
    identical results; an error of the step itself on rank 1 -> rank 1 re-raises it, rank 0 raises too (nobody waits for a dead peer)."""
    mp.spawn(_agreement_worker, args=(2, _free_port(), scenario, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    if scenario == "all_captured":
        assert r0["ok"] is True and r1["ok"] is True and r0["form"] == r1["form"] == "graph"
    elif scenario == "refused_on_rank_1":
        assert r0["ok"] is False and r1["ok"] is False, (r0, r1)
        assert r0["form"] == r1["form"] == "eager"
        assert r0["grad_sum"] == r1["grad_sum"], "the exchange after the agreed fallback completed on both ranks with the same result"
    else:
        assert r1["raised"].startswith("ValueError"), r1
        assert r0["raised"].startswith("RuntimeError") and "another rank failed" in r0["raised"], r0


def test_control_group_of_a_gloo_default_group_is_that_group():
    sys.path.insert(0, ROOT)
    from learned_quantization_amd.ddp import agree, control_group
    assert control_group() is None and agree(1, None) == 1          # torch.distributed not initialised: a single process decides alone
