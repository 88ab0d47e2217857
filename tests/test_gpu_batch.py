"""GPU tests of the multi-tensor batch (lq_batch_*): bit-identical to the single-tensor entry points."""
import numpy as np
import pytest
import torch

from oracle import lq_oracle as O
from oracle import lq_oracle_f64 as O64

from _bounds import assert_within_terms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(dev, config, orient, mode="nq", value=1e-3, seed=3, kernel_storage=None):
    import learned_quantization_amd as lq
    lq.reset_layer_names()
    m = lq.build_model(config, mode=mode, value=value, seed=seed, orientation=orient, device=dev, kernel_storage=kernel_storage)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for s in lq.scale_parameters(m):
            s.copy_((torch.rand(s.shape, generator=g) * 9e-3 + 1e-3).to(dev))
    return m


@pytest.mark.parametrize("config,orient", [("mnist", "rowwise"), ("mnist", "columnwise"), ("cifar", "channelwise"),
                                           ("cifar", "rowwise"), ("cifar", "columnwise"), ("cifar", "scalar"),
                                           ("imagenette", "channelwise"), ("resnet50", "channelwise"),
                                           ("resnet50", "columnwise"),
                                           # round 4: conv kernels stored OIHW -- row-wise / column-wise scales are narrow column matrices
                                           # (periodic float4 stream), per-tensor scales one long row (units of 4096 elements)
                                           ("imagenette", "rowwise"), ("imagenette", "columnwise"), ("imagenette", "scalar"),
                                           ("resnet50", "scalar")])
def test_batch_equals_single_tensor_ops_bitwise(dev, config, orient):
    import learned_quantization_amd as lq
    m = _model(dev, config, orient)
    batch = lq.FakeQuantBatch(m)
    outs = batch.quantize_all()
    g = torch.Generator(device=dev).manual_seed(1)
    dys = [torch.randn(o.shape, device=dev, generator=g) * 1e-3 for o in outs]
    torch.autograd.backward(outs, dys)
    for e, o, d in zip(batch.entries, outs, dys):
        ref = lq.fq_forward(e.param.data, e.nested.scale.data)
        assert torch.equal(o, ref), f"{e.layer.name} slot {e.slot}: forward"
        ds_ref = lq.fq_scale_grad(e.param.data, e.nested.scale.data, d, e.nested.penalty_threshold)
        assert torch.equal(e.nested.scale.grad, ds_ref), f"{e.layer.name} slot {e.slot}: scale grad"
        assert torch.equal(e.param.grad, d), "dP must be dy"
        # and against the oracle (sampled: the largest tensors are checked through the single-tensor op above)
        if e.param.numel() <= 200000:
            _, out_o = O.fq_forward(e.param.detach().cpu().numpy(), e.nested.scale.detach().cpu().numpy())
            np.testing.assert_array_equal(o.detach().cpu().numpy(), out_o)
    # batched Adam == per-tensor K6
    s_before = [e.nested.scale.detach().clone() for e in batch.entries]
    grads = [e.nested.scale.grad.clone() for e in batch.entries]
    opt = lq.BatchedScaleAdam(batch)
    opt.step()
    for e, s0, gr in zip(batch.entries, s_before, grads):
        s_ref, mm, vv = s0.clone(), torch.zeros_like(s0), torch.zeros_like(s0)
        lq.ops.scale_adam_step_(s_ref, gr, mm, vv, 1, lr=1e-4, min_value=lq.SCALE_INIT)
        np.testing.assert_allclose(e.nested.scale.detach().cpu().numpy(), s_ref.cpu().numpy(), rtol=1e-6)
        assert float(e.nested.scale.min()) >= O.SCALE_MIN


def test_batched_training_matches_unbatched(dev, tmp_path):
    from learned_quantization_amd.train import Trainer, synthetic_batch
    # the dense model: rocBLAS GEMMs are run-to-run deterministic, MIOpen's convolution weight-gradients are not, and the
    # thresholded scale gradient amplifies last-bit differences of dy (a group flipping "all above" changes ds by orders
    # of magnitude) -- the bitwise equality of every fake-quant result is covered by the previous test
    x, y = synthetic_batch("mnist", 32, dev, torch.Generator(device=dev).manual_seed(0))
    results = []
    for batched in (False, True):
        tr = Trainer("mnist", "nq", 1e-3, "rowwise", None, device=dev, log_dir=str(tmp_path), batched=batched)
        tr.model.eval()             # no dropout randomness / BN batch statistics in the comparison
        losses = []
        for _ in range(3):
            if tr.batch is not None:
                tr.batch.quantize_all()
            tr.opt.zero_grad(set_to_none=True)
            tr.scale_opt.zero_grad(set_to_none=True)
            loss = tr.loss(y, tr.model(x))
            loss.backward()
            if tr.batch is not None:
                tr.batch.finish_backward()      # the trainer's batch hands out leaves (FakeQuantBatch(autograd=False))
            tr.opt.step()
            tr.scale_opt.step()
            losses.append(float(loss))
        results.append((losses, [s.detach().clone() for s in tr.scale_opt.param_groups[0]["params"]]))
    np.testing.assert_allclose(results[0][0], results[1][0], rtol=1e-5)
    for a, b in zip(results[0][1], results[1][1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5)


def test_batch_ste_only_tensors_and_errors(dev):
    import learned_quantization_amd as lq
    m = _model(dev, "cifar", "channelwise", mode="cl", value=1e-7)
    batch = lq.FakeQuantBatch(m)
    outs = batch.quantize_all()
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    for e, o in zip(batch.entries, outs):
        assert torch.equal(o, lq.fq_forward(e.param.data, e.nested.scale.data))
        assert e.nested.scale.grad is None                     # STE-only: no scale gradient from the op
        assert torch.equal(e.param.grad, torch.ones_like(o))
    m.convs[0].kernel.data = m.convs[0].kernel.data.clone()    # re-allocation must be detected, not silently stale
    with pytest.raises(RuntimeError, match="re-allocated"):
        batch.quantize_all()


@pytest.mark.parametrize("kind", ["maxbin", "difference", "inverse"])
@pytest.mark.parametrize("orient", ["rowwise", "columnwise", "channelwise", "scalar"])
def test_injected_penalty_grads_equal_autograd_of_the_loss_object(dev, kind, orient, tmp_path):
    """lq_batch_penalty_grads == autograd through SCCE*.compute_total_loss (which itself is checked against the oracle)."""
    import learned_quantization_amd as lq
    gamma = 0.37
    m = _model(dev, "cifar", orient, mode="cl", value=gamma)
    layers = lq.custom_layers_of(m)
    cls = {"maxbin": lq.SCCEMaxBin, "difference": lq.SCCEDifference, "inverse": lq.SCCEInverse}[kind]
    loss_obj = cls(layers, gamma, str(tmp_path))
    y = torch.tensor([1, 0, 3], device=dev)
    p = torch.softmax(torch.randn(3, 10, device=dev), dim=1)
    loss_obj.compute_total_loss(y, p).mean().backward()
    want = {}
    for l in layers:
        want[l.name] = (None if l.kernel.grad is None else l.kernel.grad.clone(), None if l.b.grad is None else l.b.grad.clone(),
                        l.nested_q_k_layer.scale.grad.clone(), l.nested_q_b_layer.scale.grad.clone())
        l.kernel.grad = None
        l.b.grad = None
        l.nested_q_k_layer.scale.grad = None
        l.nested_q_b_layer.scale.grad = None
    batch = lq.FakeQuantBatch(m)
    g = torch.Generator(device=dev).manual_seed(3)
    seeds = {}
    for l in layers:                       # pretend the task loss already left gradients: the injection must ADD to them
        l.kernel.grad = torch.randn(l.kernel.shape, device=dev, generator=g) * 1e-3
        l.b.grad = torch.randn(l.b.shape, device=dev, generator=g) * 1e-3
        seeds[l.name] = (l.kernel.grad.clone(), l.b.grad.clone())
    batch.inject_penalty_grads(kind, gamma)
    # float64 oracle of gamma * penalty, within 1e-5 * sum|terms| (tests/_bounds.py); the injection ADDS to P.grad, which
    # costs one float32 rounding of the sum: 2^-23 * |seed + injected|
    l64 = []
    for l in layers:
        k, ks = l.kernel.detach().cpu().numpy(), l.nested_q_k_layer.scale.detach().cpu().numpy()
        b, bs = l.b.detach().cpu().numpy(), l.nested_q_b_layer.scale.detach().cpu().numpy()
        l64.append((k, ks, O.group_descriptor(k.shape, ks.shape), b, bs, O.group_descriptor(b.shape, bs.shape)))
    g64 = O64.penalty_grads(kind, l64, gamma)
    u = 2.0 ** -23
    for l, e in zip(layers, g64):
        wk, wb, wsk, wsb = want[l.name]
        sk, sb = seeds[l.name]
        if kind != "inverse":
            for got, seed, w_hip, w64, nm in ((l.kernel.grad, sk, wk, e["dK"], "dK"), (l.b.grad, sb, wb, e["db"], "db")):
                got, seed, w_hip = got.cpu().numpy().astype(np.float64), seed.cpu().numpy().astype(np.float64), w_hip.cpu().numpy().astype(np.float64)
                round_off = u * np.abs(seed + w_hip).reshape(-1)
                # the batch kernel and the single-tensor kernel run the same device code
                assert np.all(np.abs(got - (seed + w_hip)).reshape(-1) <= round_off), f"{l.name} {nm}: batch != seed + single-tensor op"
                if kind == "difference":      # MaxBin dP depends on the float32 tie split: checked against the f32 oracle elsewhere
                    assert_within_terms(w_hip, w64, e[nm + "_abs"], f"{l.name} {nm}")
        else:
            assert torch.equal(l.kernel.grad, sk) and torch.equal(l.b.grad, sb)
        # batch finalize and single-tensor finalize merge the same float64 partials in different block shapes
        assert_within_terms(l.nested_q_k_layer.scale.grad.cpu().numpy(), wsk.cpu().numpy(), e["dsK_abs"], f"{l.name} ds_k: batch vs single-tensor op", rel=1e-6)
        assert_within_terms(l.nested_q_b_layer.scale.grad.cpu().numpy(), wsb.cpu().numpy(), e["dsb_abs"], f"{l.name} ds_b: batch vs single-tensor op", rel=1e-6)
        assert_within_terms(l.nested_q_k_layer.scale.grad.cpu().numpy(), e["dsK"], e["dsK_abs"], f"{l.name} ds_k")
        assert_within_terms(l.nested_q_b_layer.scale.grad.cpu().numpy(), e["dsb"], e["dsb_abs"], f"{l.name} ds_b")


def test_batched_mode_refuses_silent_gradient_accumulation(dev):
    import learned_quantization_amd as lq
    m = _model(dev, "mnist", "rowwise")
    batch = lq.FakeQuantBatch(m)
    outs = batch.quantize_all()
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    outs = batch.quantize_all()
    with pytest.raises(RuntimeError, match="not supported in batched mode"):
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])


# LDS-tile path (lq_conv_tile.hpp): <= 9 taps, co % 4 == 0 -- full tiles, edge tiles along ci (3, 48, 100, 33, 300: the last
# c-tile is partial) and along co (40, 20, 36, 8, 12), several channel blocks per tile (1x1, 2x2, 1x3), even tap counts;
# element-wise path: 7x7 and 5x3 (more than 9 taps), co = 10 (co % 4 != 0)
CONV_SHAPES = [(3, 3, 3, 32), (3, 3, 64, 128), (7, 7, 3, 64), (1, 1, 64, 128), (3, 3, 256, 256), (1, 1, 512, 2048), (5, 3, 6, 10),
               (3, 3, 48, 40), (2, 2, 100, 20), (1, 3, 33, 36), (3, 1, 16, 8), (1, 1, 300, 12), (3, 3, 160, 96),
               # two-stage K2 (lq_conv_tile.hpp): odd numbers of channel blocks, partial stages, two taps, one channel in the last wave
               (1, 1, 96, 64), (1, 1, 160, 32), (1, 2, 70, 16), (1, 1, 40, 8), (3, 3, 5, 4), (2, 3, 29, 68), (1, 1, 2048, 64)]


@pytest.mark.parametrize("orient", ["rowwise", "columnwise", "channelwise", "scalar"])
def test_oihw_companion_ops_are_bit_identical_to_permuting(dev, orient):
    """lq_fq_forward_oihw / lq_fq_scale_grad_oihw (conv kernels: HWIO parameter, OIHW consumer; NQ-L:321, 338-350): the OIHW
    output is the HWIO output permuted, ds is what lq_fq_scale_grad gives on the un-permuted gradient, dP is that gradient --
    all bit for bit, so handing MIOpen the companion changes no number, only saves the transposition launches."""
    import learned_quantization_amd as lq
    from learned_quantization_amd import ops
    rng = np.random.default_rng(17)
    for shape in CONV_SHAPES:
        k = torch.tensor(rng.normal(0, 0.05, size=shape).astype(np.float32), device=dev)
        s = torch.tensor(rng.uniform(1e-3, 1e-2, size=O.scale_shape(shape, orient)).astype(np.float32), device=dev)
        out, out_oihw = ops.fq_forward_oihw(k, s)
        ref = lq.fq_forward(k, s)
        assert torch.equal(out, ref), f"{shape} {orient}: HWIO output"
        assert out_oihw.is_contiguous() and torch.equal(out_oihw, ref.permute(3, 2, 0, 1)), f"{shape} {orient}: OIHW companion"
        view, only = ops.fq_forward_oihw(k, s, hwio_out=False)          # the companion alone (the autograd op's form)
        assert torch.equal(only, out_oihw) and torch.equal(view, ref), f"{shape} {orient}: companion-only forward"
        _, out_o = O.fq_forward(k.cpu().numpy(), s.cpu().numpy())
        np.testing.assert_array_equal(out_oihw.cpu().numpy(), np.transpose(out_o, (3, 2, 0, 1)))
        dy_oihw = torch.tensor((rng.normal(0, 1, size=out_oihw.shape) * 10.0 ** rng.uniform(-8, -2, size=out_oihw.shape)).astype(np.float32), device=dev)
        for lam in (1e-10, 2e-2):
            ds, dP = ops.fq_scale_grad_oihw(k, s, dy_oihw, lam)
            dy_hwio = dy_oihw.permute(2, 3, 1, 0).contiguous()
            assert torch.equal(dP, dy_hwio), f"{shape} {orient}: dP"
            assert torch.equal(ds, lq.fq_scale_grad(k, s, dy_hwio, lam)), f"{shape} {orient} lam={lam}: ds"


@pytest.mark.parametrize("storage", ["hwio", "oihw"])
def test_conv_layer_hands_miopen_the_oihw_companion(dev, storage):
    """CustomConv2DLayer.call (NQ-L:338-350) == explicit fake-quant + permute + conv2d: through the companion path for a kernel
    stored HWIO, and with no transposition anywhere for a kernel stored OIHW (the gradient then has the parameter's strides)."""
    import learned_quantization_amd as lq
    import torch.nn.functional as F
    lq.reset_layer_names()
    layer = lq.CustomConv2DLayer(seed=0, penalty_threshold=1e-3, orientation="channelwise", initializer=lq.RandomNormal(seed=3),
                                 filters=16, kernel_size=(3, 3), strides=(1, 1), padding="same", name="c", regularizer=None,
                                 input_shape=8, device=dev, kernel_storage=storage)
    assert layer.kernel.is_contiguous() == (storage == "hwio") and layer.kernel.permute(3, 2, 0, 1).is_contiguous() == (storage == "oihw")
    with torch.no_grad():
        layer.nested_q_k_layer.scale.uniform_(1e-3, 1e-2)
    x = torch.randn(4, 8, 12, 12, device=dev)
    y = layer(x)
    qk = lq.fq_forward(layer.kernel.data, layer.nested_q_k_layer.scale.data)
    qb = lq.fq_forward(layer.b.data, layer.nested_q_b_layer.scale.data)
    y_ref = F.conv2d(x, qk.permute(3, 2, 0, 1).contiguous(), None, 1, 1) + qb.view(1, -1, 1, 1)
    # two executions of the same convolution: equal up to MIOpen's own run-to-run noise (profiles/r04/rehearsal_diag/)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.cpu().numpy(), rtol=1e-4, atol=1e-6)
    y.square().mean().backward()
    assert layer.kernel.grad is not None and layer.kernel.grad.shape == layer.kernel.shape
    assert layer.kernel.grad.stride() == layer.kernel.stride()
    assert layer.nested_q_k_layer.scale.grad is not None and bool((layer.nested_q_k_layer.scale.grad <= 0).all())


@pytest.mark.parametrize("hwio_out", [True, False])
def test_batch_emits_oihw_companions_and_reads_oihw_gradients(dev, hwio_out):
    """FakeQuantBatch on a conv model: one forward launch emits every OIHW companion, one scale-gradient launch gathers every
    OIHW weight gradient and writes dP in HWIO order -- bit-identical to the single-tensor ops on the permuted tensors.
    ``hwio_out=False`` (the trainer's form): the HWIO output is not materialised where the LDS tile writes the companion; the
    batch hands out the permuted view of the companion in its place."""
    import learned_quantization_amd as lq
    m = _model(dev, "cifar", "channelwise", kernel_storage="hwio")
    batch = lq.FakeQuantBatch(m, hwio_out=hwio_out)
    assert hwio_out or sum(e.out is None for e in batch.entries) == 6, "every 3x3 kernel of the CIFAR CNN takes the tile path"
    batch.quantize_all()
    layers = lq.custom_layers_of(m)
    for l in layers:                    # against the single-tensor forward
        assert torch.equal(l._q_pre[0], lq.fq_forward(l.kernel.data, l.nested_q_k_layer.scale.data))
        assert l._q_pre[0].is_contiguous() == hwio_out
    g = torch.Generator(device=dev).manual_seed(9)
    outs, dys = [], []
    for l in layers:
        qk_hwio, qb, qk_oihw = l._q_pre
        assert qk_oihw is not None and torch.equal(qk_oihw, qk_hwio.permute(3, 2, 0, 1)) and qk_oihw.is_contiguous()
        outs += [qk_oihw, qb]
        dys += [torch.randn(qk_oihw.shape, device=dev, generator=g) * 1e-3, torch.randn(qb.shape, device=dev, generator=g) * 1e-3]
    torch.autograd.backward(outs, dys)
    for l, d_k, d_b in zip(layers, dys[0::2], dys[1::2]):
        d_hwio = d_k.permute(2, 3, 1, 0).contiguous()
        assert torch.equal(l.kernel.grad, d_hwio), f"{l.name}: dP must be the un-permuted dy"
        assert torch.equal(l.nested_q_k_layer.scale.grad,
                           lq.fq_scale_grad(l.kernel.data, l.nested_q_k_layer.scale.data, d_hwio, l.nested_q_k_layer.penalty_threshold))
        assert torch.equal(l.b.grad, d_b)
        assert torch.equal(l.nested_q_b_layer.scale.grad,
                           lq.fq_scale_grad(l.b.data, l.nested_q_b_layer.scale.data, d_b, l.nested_q_b_layer.penalty_threshold))


@pytest.mark.parametrize("config,orient,storage", [("cifar", "channelwise", "oihw"), ("cifar", "rowwise", "hwio"), ("mnist", "columnwise", None),
                                                   ("imagenette", "channelwise", "oihw"), ("imagenette", "scalar", "hwio")])
def test_fused_scale_update_equals_scale_grad_then_adam(dev, config, orient, storage):
    """lq_batch_scale_grad_step (the finalize applies Adam + MinValueConstraint to the scale whose gradient it has just emitted,
    custom_layers.py:116, 158) == lq_batch_scale_grad followed by lq_batch_scale_adam: ds, moments and scales bit for bit, over
    three steps, for either kernel storage (the HWIO one through the OIHW companions and the LDS tiles)."""
    import learned_quantization_amd as lq
    res = {}
    for fused in (False, True):
        m = _model(dev, config, orient, value=1e-4, kernel_storage=storage)
        batch = lq.FakeQuantBatch(m, hwio_out=False)
        opt = lq.BatchedScaleAdam(batch, fused=fused)
        g = torch.Generator(device=dev).manual_seed(21)
        hist = []
        for step in range(3):
            opt.zero_grad()
            batch.quantize_all()
            outs, dys = [], []
            for l in lq.custom_layers_of(m):
                qk, qb, qo = l._q_pre
                for o in ((qo if qo is not None else qk), qb):
                    if o is not None:
                        outs.append(o)
                        dys.append(torch.randn(tuple(o.shape), device=dev, generator=g) * 1e-3)
            torch.autograd.backward(outs, dys)
            ds = [e.nested.scale.grad.clone() for e in batch.entries]
            opt.step()
            hist.append((ds, [e.nested.scale.detach().clone() for e in batch.entries], [e.m.clone() for e in batch.entries],
                         [e.v.clone() for e in batch.entries]))
        res[fused] = hist
    for step, (a, b) in enumerate(zip(res[False], res[True])):
        for what, xa, xb in zip(("ds", "scale", "m", "v"), a, b):
            for i, (ta, tb) in enumerate(zip(xa, xb)):
                assert torch.equal(ta, tb), f"step {step} tensor {i}: {what}"
    assert any(float((s1 - s0).abs().max()) > 0 for s0, s1 in zip(res[True][0][1], res[True][2][1])), "the scales moved"


def test_fused_scale_update_is_refused_where_something_sits_between_gradient_and_update(dev):
    import learned_quantization_amd as lq
    m = _model(dev, "cifar", "channelwise", mode="cl", value=1e-7)          # STE-only layers: ds comes from the loss term
    with pytest.raises(ValueError, match="nested-quantization"):
        lq.BatchedScaleAdam(lq.FakeQuantBatch(m), fused=True)
    m = _model(dev, "mnist", "rowwise")
    b = lq.FakeQuantBatch(m)
    opt = lq.BatchedScaleAdam(b, fused=True)
    with pytest.raises(RuntimeError, match="without a backward"):
        opt.step()
    # ... and a SECOND scale-gradient pass before step() is refused: the finalize of the first has already applied Adam to the scales
    # (gradient accumulation or a retried backward would update the scales twice and the weights once)
    outs = b.quantize_all()
    torch.autograd.backward(outs, [torch.ones_like(o) * 1e-3 for o in outs])
    scales = [e.nested.scale.detach().clone() for e in b.entries]
    opt.zero_grad()
    for p in m.parameters():
        p.grad = None
    outs = b.quantize_all()
    with pytest.raises(RuntimeError, match="already been applied"):
        torch.autograd.backward(outs, [torch.ones_like(o) * 1e-3 for o in outs])
    assert all(torch.equal(s0, e.nested.scale.detach()) for s0, e in zip(scales, b.entries)), "the refused pass changed nothing"
    opt.step()                                    # acknowledges the first pass; the next step is accepted again
    opt.zero_grad()
    for p in m.parameters():
        p.grad = None
    outs = b.quantize_all()
    torch.autograd.backward(outs, [torch.ones_like(o) * 1e-3 for o in outs])
    opt.step()


@pytest.mark.parametrize("config,orient,storage", [("cifar", "channelwise", "oihw"), ("cifar", "rowwise", "hwio"), ("mnist", "columnwise", None),
                                                   ("imagenette", "channelwise", "oihw")])
def test_leaf_mode_equals_the_autograd_node(dev, config, orient, storage):
    """FakeQuantBatch(autograd=False) -- fake-quantised tensors as leaves, finish_backward() after the backward pass (the trainer's
    form) -- gives the outputs, dP and ds of the one-node autograd form bit for bit; where a parameter already holds a gradient
    (a regulariser's, a data-parallel bucket view) dP is ADDED to it, as AccumulateGrad would."""
    import learned_quantization_amd as lq
    res = {}
    for autograd in (True, False):
        m = _model(dev, config, orient, value=1e-4, kernel_storage=storage)
        batch = lq.FakeQuantBatch(m, hwio_out=False, autograd=autograd)
        g = torch.Generator(device=dev).manual_seed(33)
        outs = batch.quantize_all()
        consumed, dys = [], []
        for l in lq.custom_layers_of(m):                       # what the layers consume: the companion where there is one
            w, qb = l.quantized_parameters()
            for o in (w, qb):
                if o is not None:
                    consumed.append(o)
                    dys.append(torch.randn(tuple(o.shape), device=dev, generator=g) * 1e-3)
        pre = {}
        if config == "imagenette":                             # a gradient that is already there (every third parameter)
            for k, e in enumerate(batch.entries):
                if k % 3 == 0:
                    e.param.grad = torch.full_like(e.param.data, 0.25)
                    pre[k] = True
        torch.autograd.backward(consumed, dys)
        if not autograd:
            batch.finish_backward()
        res[autograd] = ([o.detach().clone() for o in outs], [e.param.grad.detach().clone() for e in batch.entries],
                         [e.nested.scale.grad.detach().clone() for e in batch.entries])
    for what, a, b in zip(("out", "dP", "ds"), res[True], res[False]):
        for i, (ta, tb) in enumerate(zip(a, b)):
            assert torch.equal(ta, tb), f"tensor {i}: {what}"
    assert any(float(t.abs().max()) > 0 for t in res[False][2])


def test_leaf_mode_finish_backward_belongs_to_leaf_mode(dev):
    import learned_quantization_amd as lq
    b = lq.FakeQuantBatch(_model(dev, "mnist", "rowwise"))
    with pytest.raises(RuntimeError, match="autograd=False"):
        b.finish_backward()
    b = lq.FakeQuantBatch(_model(dev, "mnist", "rowwise"), autograd=False)
    outs = b.quantize_all()
    torch.autograd.backward(outs, [torch.ones_like(o) * 1e-3 for o in outs])
    b.finish_backward()
    with pytest.raises(RuntimeError, match="one backward pass per forward"):       # a second call would add dP once more
        b.finish_backward()


@pytest.mark.parametrize("orient", ["channelwise", "rowwise", "columnwise", "scalar"])
@pytest.mark.parametrize("seed", [0, 1])
def test_batch_of_odd_conv_shapes_equals_single_tensor_ops(dev, orient, seed):
    """The batch's own traversal forms (round 4: fragment column tiles with cut groups, partial last tiles and row blocks shorter than a
    stage; the periodic float4 stream for narrow column matrices; 2048 / 4096-element units) on conv kernels the models do not have --
    2x2 ... 5x3 taps, odd channel counts, 3 ... 700 output channels, stored OIHW -- against the single-tensor ops bit for bit
    (lambda < 4e-4: exact vote sums) and, for the small ones, against the oracle."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(100 + seed)
    taps = [(3, 3), (1, 1), (2, 2), (1, 3), (3, 1), (2, 3), (3, 5), (5, 3), (1, 2), (4, 4), (5, 5), (7, 7)]
    lq.reset_layer_names()
    layers = []
    for k in range(22):
        kh, kw = taps[int(rng.integers(len(taps)))]
        ci = int(rng.choice([3, 4, 7, 12, 20, 28, 33, 40, 64, 100, 128, 172]))
        co = int(rng.choice([3, 8, 17, 32, 60, 64, 130, 256, 700]))
        if kh * kw * ci * co > 3_000_000:
            co = 64
        l = lq.CustomConv2DLayer(seed=k, penalty_threshold=float(rng.choice([1e-11, 1e-8, 1e-4])), orientation=orient,
                                 initializer=lq.RandomNormal(seed=1000 * seed + k), filters=co, kernel_size=(kh, kw), strides=(1, 1),
                                 padding="same", name="c", regularizer=None, input_shape=ci, device=dev, kernel_storage="oihw")
        with torch.no_grad():
            l.nested_q_k_layer.scale.copy_(torch.tensor(rng.uniform(1e-3, 1e-2, size=tuple(l.nested_q_k_layer.scale.shape)).astype(np.float32)))
            l.nested_q_b_layer.scale.fill_(float(rng.uniform(1e-3, 1e-2)))
        layers.append(l)
    batch = lq.FakeQuantBatch(layers, hwio_out=False)
    outs = batch.quantize_all()
    g = torch.Generator(device=dev).manual_seed(7 + seed)
    dys = []
    for o in outs:      # magnitudes over ten decades: votes on both sides of every threshold in use
        mag = torch.pow(10.0, torch.empty(tuple(o.shape), device=dev).uniform_(-13.0, -3.0, generator=g))
        dys.append(torch.randn(tuple(o.shape), device=dev, generator=g) * mag)
    torch.autograd.backward(outs, dys)
    for e, o, d in zip(batch.entries, outs, dys):
        tag = f"{tuple(e.param.shape)} {orient} desc {e.desc}"
        assert torch.equal(o, lq.fq_forward(e.param.data, e.nested.scale.data)), f"{tag}: forward"
        ds_ref = lq.fq_scale_grad(e.param.data, e.nested.scale.data, d, e.nested.penalty_threshold)
        assert torch.equal(e.nested.scale.grad, ds_ref), f"{tag}: scale gradient"
        assert torch.equal(e.param.grad, d), f"{tag}: dP must be dy"
        if e.param.numel() <= 100000:
            P, s = e.param.detach().cpu().numpy(), e.nested.scale.detach().cpu().numpy()
            _, out_o = O.fq_forward(P, s)
            np.testing.assert_array_equal(o.detach().cpu().numpy(), out_o, err_msg=tag)
            _, ds_o = O.nq_backward(P, s, e.nested.penalty_threshold, d.cpu().numpy())
            np.testing.assert_allclose(e.nested.scale.grad.cpu().numpy(), ds_o, rtol=1e-5, atol=0, err_msg=tag)
