"""Conv kernels stored in the order MIOpen consumes (layers.py ``kernel_storage="oihw"``): shape and values are the
reference's HWIO kernel (custom_layers.py:321), only the element order in memory differs, and the kernels are told so through the
descriptor (descriptor.memory_descriptor).  Everything here compares INDEX BY INDEX with the contiguous HWIO tensor -- whose own
parity with the oracle is what the rest of the suite establishes -- and once directly with the oracle.

  * single-tensor ops (K1 with q, K2+K3, K4, K5a/b values and gradients) on the permuted tensor == on its contiguous copy, bit for bit
    (vote sums are exact for lambda < 4e-4, lq_common.hpp: the traversal order does not matter; lambda = 2e-2 within 1e-6);
  * the multi-tensor batch, the layer call and a whole training step of the CIFAR CNN give the same numbers for either storage;
  * no transposition kernel and no LDS-tile kernel is involved: the gradient MIOpen returns is dP itself (custom_layers.py:118).
"""
import numpy as np
import pytest
import torch

from oracle import lq_oracle as O
from test_gpu_parity import dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

SHAPES = [(3, 3, 8, 16), (3, 3, 64, 128), (1, 1, 64, 256), (7, 7, 3, 64), (3, 3, 13, 10), (1, 1, 5, 7), (5, 5, 16, 32)]
ORIENTS = ["rowwise", "columnwise", "channelwise", "scalar"]


def _oihw_stored(t):
    """Same shape and values, elements in (co, ci, kh, kw) order."""
    return t.permute(3, 2, 0, 1).contiguous().permute(2, 3, 1, 0)


def _case(dev, shape, orient, seed):
    import learned_quantization_amd as lq
    g = torch.Generator(device=dev).manual_seed(seed)
    k = torch.randn(shape, device=dev, generator=g) * 0.05
    k.view(-1)[::97] = 0.0                                   # exact zeros: out == 0 -> eps (custom_layers.py:63)
    s = torch.rand(lq.scale_shape(shape, orient), device=dev, generator=g) * 9e-3 + 1e-3
    dy = torch.randn(shape, device=dev, generator=g) * 1e-3
    dy.view(-1)[::5] *= 1e-9                                 # ratios on both sides of lambda
    return k, s, dy


@pytest.mark.parametrize("orient", ORIENTS)
@pytest.mark.parametrize("shape", SHAPES)
def test_single_tensor_ops_on_an_oihw_stored_kernel(dev, shape, orient):
    import learned_quantization_amd as lq
    from learned_quantization_amd import ops
    k, s, dy = _case(dev, shape, orient, 11)
    kp, dyp = _oihw_stored(k), _oihw_stored(dy)
    assert lq.memory_descriptor(kp.shape, kp.stride(), s.shape) is not None
    # K1 with the integer view
    out, q = lq.fq_forward(k, s, q_dtype=torch.int32)
    outp, qp = lq.fq_forward(kp, s, q_dtype=torch.int32)
    assert outp.stride() == kp.stride() and qp.stride() == kp.stride(), "outputs carry the parameter's strides"
    assert torch.equal(outp, out) and torch.equal(qp, q)
    # ... and directly against the oracle (custom_layers.py:55-60)
    q_o, out_o = O.fq_forward(k.cpu().numpy(), s.cpu().numpy())
    assert np.array_equal(outp.cpu().numpy(), out_o) and np.array_equal(qp.cpu().numpy(), q_o.astype(np.int32))
    for lam in (1e-10, 2e-2):
        ds, parts = lq.fq_scale_grad(k, s, dy, lam, return_parts=True)
        for d_in in (dyp, dy):                               # a gradient in the other element order is brought into the parameter's
            dsp, partsp = lq.fq_scale_grad(kp, s, d_in, lam, return_parts=True)
            assert torch.equal(partsp[0], parts[0]) and torch.equal(partsp[2], parts[2]), "max|q| and vote counts"
            if lam < 4e-4:
                assert torch.equal(dsp, ds), f"lam={lam}: exact vote sums do not depend on the traversal"
            else:
                np.testing.assert_allclose(dsp.cpu().numpy(), ds.cpu().numpy(), rtol=1e-6, atol=0)
        _, ds_o = O.nq_backward(k.cpu().numpy(), s.cpu().numpy(), lam, dy.cpu().numpy())
        np.testing.assert_allclose(dsp.cpu().numpy(), ds_o, rtol=1e-5, atol=0)
        o4, ds4 = lq.fq_fwd_bwd_fused(kp, s, dyp, lam)
        assert torch.equal(o4, out) and o4.stride() == kp.stride()
        if lam < 4e-4:
            assert torch.equal(ds4, ds)
    # K5a / K5b: values and gradients through autograd
    for term in (ops.maxbin_term, ops.difference_term):
        res = []
        for kk in (k, kp):
            kk = kk.detach().requires_grad_(True)
            ss = s.detach().clone().requires_grad_(True)
            t = term(kk, ss)
            (t * 0.37).backward()
            res.append((t.detach(), kk.grad, ss.grad))
        assert res[1][1].stride() == kp.stride()
        assert torch.equal(res[0][0], res[1][0]) or abs(float(res[0][0]) - float(res[1][0])) <= 1e-6 * abs(float(res[0][0]))
        assert torch.equal(res[0][1], res[1][1]), f"{term.__name__}: dP"
        np.testing.assert_allclose(res[1][2].cpu().numpy(), res[0][2].cpu().numpy(), rtol=2e-6, atol=0)


@pytest.mark.parametrize("orient", ORIENTS)
def test_batch_and_layers_give_the_same_numbers_for_either_storage(dev, orient):
    """Two CIFAR CNNs from one seed, conv kernels stored HWIO / OIHW: FakeQuantBatch forward, the layer calls, the backward
    through MIOpen and the batched Adam step agree index by index; the OIHW-stored model launches no companion / tile work."""
    import learned_quantization_amd as lq
    models, batches = {}, {}
    for st in ("hwio", "oihw"):
        lq.reset_layer_names()
        m = lq.build_model("cifar", mode="nq", value=1e-4, seed=3, orientation=orient, device=dev, kernel_storage=st)   # lambda < 4e-4: exact vote sums
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for s in lq.scale_parameters(m):
                s.copy_((torch.rand(s.shape, generator=g) * 9e-3 + 1e-3).to(dev))
        m.eval()                                             # no dropout masks between the two models
        models[st], batches[st] = m, lq.FakeQuantBatch(m, hwio_out=False)
    assert all(e.out_oihw is None for e in batches["oihw"].entries), "an OIHW-stored kernel needs no companion"
    assert any(e.out_oihw is not None for e in batches["hwio"].entries)
    for (n0, p0), (n1, p1) in zip(models["hwio"].named_parameters(), models["oihw"].named_parameters()):
        assert n0 == n1 and torch.equal(p0, p1), f"{n0}: same values from the same seed"
        if p1.dim() == 4:
            assert p1.permute(3, 2, 0, 1).is_contiguous() and p0.is_contiguous()
    # (a) given upstream gradients (no MIOpen in between): everything is bit-identical
    res = {}
    for st in ("hwio", "oihw"):
        m, b = models[st], batches[st]
        opt = lq.BatchedScaleAdam(b)
        opt.zero_grad()
        outs = b.quantize_all()
        g = torch.Generator(device=dev).manual_seed(9)
        dys = [torch.randn(tuple(o.shape), device=dev, generator=g) * 1e-3 for o in outs]      # the same logical values for both
        torch.autograd.backward(outs, dys)
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        assert len(grads) == 2 * len(b.entries)
        opt.step()
        res[st] = ([o.detach().clone() for o in outs], grads, {n: p.detach().clone() for n, p in m.named_parameters()})
        for p in m.parameters():
            p.grad = None
    for o0, o1 in zip(res["hwio"][0], res["oihw"][0]):
        assert torch.equal(o0, o1), "fake-quantised tensors"
    for n, g0 in res["hwio"][1].items():
        assert torch.equal(g0, res["oihw"][1][n]), f"{n}: gradient"
    for n, p0 in res["hwio"][2].items():
        assert torch.equal(p0, res["oihw"][2][n]), f"{n} after the Adam step of the scales"
    # (b) through the layers and MIOpen: the forward is the same computation on the same OIHW tensors; the weight gradients agree up
    # to MIOpen's own run-to-run noise (atomics), the scale gradients follow them
    x = torch.randn(8, 3, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    y = torch.randint(0, 10, (8,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    res = {}
    for st in ("hwio", "oihw"):
        m, b = models[st], batches[st]
        b.quantize_all()
        probs = m(x)
        lq.sparse_categorical_crossentropy(y, probs).mean().backward()
        for n, p in m.named_parameters():
            assert p.grad is None or p.grad.stride() == p.stride() or p.dim() != 4, f"{n}: the gradient has the parameter's strides"
        res[st] = (probs.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    # MIOpen saw the same OIHW tensors; two executions of one convolution need not agree bit for bit (some of its forward kernels
    # split the reduction and accumulate with atomics: profiles/r04/rehearsal_diag/), an element-order mix-up changes the outputs grossly
    np.testing.assert_allclose(res["oihw"][0].cpu().numpy(), res["hwio"][0].cpu().numpy(), rtol=1e-4, atol=1e-6)
    for n, g0 in res["hwio"][1].items():
        if "scale" in n:
            continue            # a vote flips when a ratio crosses lambda: ds is not continuous in dW; checked against its own dW below
        # norm-wise, three orders of magnitude above the noise measured (1e-6 of the tensor's largest gradient): an element-order
        # mix-up moves elements by the size of the gradients themselves
        g0, g1 = g0.cpu().numpy(), res["oihw"][1][n].cpu().numpy()
        assert np.abs(g1 - g0).max() <= 1e-3 * np.abs(g0).max(), n
    for st in ("hwio", "oihw"):
        for e in batches[st].entries:
            assert torch.equal(e.nested.scale.grad, lq.fq_scale_grad(e.param.data, e.nested.scale.data, e.param.grad, e.nested.penalty_threshold)), \
                f"{st}: ds of the batch == the single-tensor op on the gradient that arrived"


def _steps(dev, cls, storage, n, batched, setup, data):
    tr = cls(*setup, device=dev, seed=7, batched=batched, kernel_storage=storage)
    if hasattr(tr, "coefficients"):
        from _linear_task import make_coefficients
        tr.coefficients = make_coefficients(tr, n)
    losses = [float(tr.step(*data).detach()) for _ in range(n)]
    return tr, losses, {k: p.detach().clone() for k, p in tr.model.named_parameters()}


@pytest.mark.parametrize("setup", [("cifar", "nqcl", (1e-4, 1e-3), "channelwise", "maxbin"), ("cifar", "nq", 1e-10, "rowwise", None)],
                         ids=["nqcl-maxbin-channelwise", "nq-rowwise"])
def test_training_steps_agree_between_storages(dev, setup):
    """Three steps of the batched trainer on the CIFAR CNN: the same parameters for either storage, BIT FOR BIT, when the upstream
    gradients are given (tests/_linear_task.py: the product's whole step -- batch forward, STE, scale gradients, MaxBin injection,
    both optimizers -- with no convolution library in the loop; MaxBin's ds = -(c/G) max/s and the vote sums, exact for lambda < 4e-4,
    do not depend on the traversal order).  Through MIOpen the same comparison is a sanity check of the losses only: its weight gradients are not
    run-to-run stable and Adam turns an ulp of a gradient into a whole step (GPUTEST_r03)."""
    from _linear_task import LinearTaskTrainer
    from learned_quantization_amd.train import Trainer, synthetic_batch
    out = {st: _steps(dev, LinearTaskTrainer, st, 3, True, setup, (None, None)) for st in ("hwio", "oihw")}
    assert out["oihw"][1] == out["hwio"][1], "losses"
    for n, p0 in out["hwio"][2].items():
        assert torch.equal(out["oihw"][2][n], p0), f"{n}: parameters after three steps"
    ref = _steps(dev, LinearTaskTrainer, "hwio", 0, True, setup, (None, None))[2]
    moved = max(float((ref[n] - p0).abs().max()) for n, p0 in out["hwio"][2].items() if "scale" in n)
    assert moved > 0.0, "the scales moved"
    x, y = synthetic_batch("cifar", 16, dev, torch.Generator(device=dev).manual_seed(0))
    e2e = {st: _steps(dev, Trainer, st, 3, True, setup, (x, y)) for st in ("hwio", "oihw")}
    assert all(np.isfinite(l) for st in e2e for l in e2e[st][1])
    np.testing.assert_allclose(e2e["oihw"][1], e2e["hwio"][1], rtol=1e-3)


def test_unbatched_trainer_and_export_on_oihw_storage(dev, tmp_path):
    """The per-tensor path (ops.my_custom_gradient inside the layer call) and the integer export see the logical HWIO tensor: given
    upstream gradients, parameters and exported integers are identical for either storage; through MIOpen the losses agree."""
    import learned_quantization_amd as lq
    from _linear_task import LinearTaskTrainer
    from learned_quantization_amd.train import Trainer, synthetic_batch
    setup = ("cifar", "nq", 1e-3, "channelwise", None)
    out = {}
    for st in ("hwio", "oihw"):
        tr, losses, params = _steps(dev, LinearTaskTrainer, st, 2, False, ("cifar", "nq", 1e-10, "channelwise", None), (None, None))
        layer = [l for l in lq.custom_layers_of(tr.model) if hasattr(l, "kernel")][0]
        qi = lq.quantized_integers(layer.kernel.data, layer.nested_q_k_layer.scale.data, torch.int8).cpu().numpy()
        assert qi.shape == tuple(layer.kernel.shape)
        out[st] = (losses, qi, params)
    assert out["oihw"][0] == out["hwio"][0]
    for n, p0 in out["hwio"][2].items():
        assert torch.equal(out["oihw"][2][n], p0), f"{n}: parameters after two steps"
    assert np.array_equal(out["oihw"][1], out["hwio"][1]), "exported integers"
    x, y = synthetic_batch("cifar", 8, dev, torch.Generator(device=dev).manual_seed(0))
    e2e = {st: _steps(dev, Trainer, st, 2, False, setup, (x, y))[1] for st in ("hwio", "oihw")}
    np.testing.assert_allclose(e2e["oihw"], e2e["hwio"], rtol=1e-3)


def test_grad_bucket_keeps_the_parameters_strides(dev):
    import learned_quantization_amd as lq
    k = torch.nn.Parameter(_oihw_stored(torch.randn(3, 3, 8, 16, device=dev)))
    w = torch.nn.Parameter(torch.randn(5, 7, device=dev))
    b = lq.GradBucket([w, k])
    assert k.grad.stride() == k.stride() and k.grad.shape == k.shape and w.grad.is_contiguous()
    assert k.grad.data_ptr() == b.flat.data_ptr() + 4 * b.offsets[1]
    (k * 2.0).sum().backward()
    assert torch.all(b.flat[b.offsets[1]:b.offsets[1] + k.numel()] == 2.0) and k.grad.data_ptr() == b.views[1].data_ptr()


def test_keras_adam_state_saved_from_one_storage_loads_into_the_other(dev):
    """torch's load_state_dict keeps the checkpoint's strides; KerasAdam updates flat memory.  Moments saved from an
    HWIO-stored model and loaded into an OIHW-stored one must continue the SAME trajectory, bit for bit."""
    import learned_quantization_amd as lq
    g = torch.Generator().manual_seed(4)
    w0 = torch.randn(3, 3, 8, 16, generator=g) * 0.05
    grads = [torch.randn(3, 3, 8, 16, generator=g) * 1e-3 for _ in range(4)]

    def param(storage):
        w = w0.clone().to(dev)
        return torch.nn.Parameter(_oihw_stored(w) if storage == "oihw" else w)
    a = param("hwio")
    opt_a = lq.KerasAdam([a])
    for gr in grads[:2]:
        a.grad = gr.to(dev)
        opt_a.step()
    sd = opt_a.state_dict()
    assert sd["lq_step"] == 2
    b = param("oihw")
    with torch.no_grad():
        b.copy_(a)
    opt_b = lq.KerasAdam([b])
    opt_b.load_state_dict(sd)
    for gr in grads[2:]:
        a.grad = gr.to(dev)
        opt_a.step()
        b.grad = _oihw_stored(gr.to(dev))
        opt_b.step()
    assert opt_b.state[b]["m"].stride() == b.stride(), "the loaded moments were brought into the parameter's element order"
    assert torch.equal(a.detach(), b.detach()) and torch.equal(opt_a.state[a]["m"], opt_b.state[b]["m"]) \
        and torch.equal(opt_a.state[a]["v"], opt_b.state[b]["v"])
    # ... and a reload into the SAME optimizer after it has run rebuilds its launch tables (fresh state tensors)
    opt_a.load_state_dict(opt_b.state_dict())
    a.grad = grads[0].to(dev)
    b.grad = _oihw_stored(grads[0].to(dev))
    opt_a.step()
    opt_b.step()
    assert torch.equal(a.detach(), b.detach())


def test_export_bytes_do_not_depend_on_the_kernel_storage(dev, tmp_path):
    """The reference's artefact (utils/log_scripts.py:61-97: weights.npy -> zip -> file_sizes.log) for a model with 1x1 convs (the
    ResNet-18-like net's shortcut convs): identical bytes for either storage -- an OIHW-stored 1x1 kernel is F-contiguous as a numpy
    array and would otherwise be pickled in Fortran order."""
    import learned_quantization_amd as lq
    blobs = {}
    for st in ("hwio", "oihw"):
        lq.reset_layer_names()
        m = lq.build_model("imagenette", mode="nq", value=1e-11, seed=3, orientation="channelwise", device=dev, kernel_storage=st)
        with torch.no_grad():
            for s in lq.scale_parameters(m):
                s.fill_(2e-3)
        d = tmp_path / st
        sizes = lq.save_compress_parameters(m, str(d))
        w = np.load(d / "weights.npy", allow_pickle=True).item()          # written by this test a moment ago
        assert any(v.ndim == 4 and v.shape[:2] == (1, 1) for v in w.values()), "the model has 1x1 kernels"
        assert all(v.flags["C_CONTIGUOUS"] for v in w.values())
        blobs[st] = (open(d / "weights.npy", "rb").read(), open(d / "file_sizes.log").read(), sizes)
    assert blobs["hwio"][0] == blobs["oihw"][0], "weights.npy bytes"
    assert blobs["hwio"][1] == blobs["oihw"][1], "file_sizes.log"
