"""GPU tests of the host surface: autograd ops, layers, loss terms, scale optimizer -- against the oracle."""
import numpy as np
import pytest
import torch

from oracle import lq_oracle as O
from oracle import lq_oracle_f64 as O64
from oracle import lq_oracle_torch as OT

from _bounds import assert_within_terms

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_autograd_nq_op(dev):
    import learned_quantization_amd as lq
    rng = np.random.default_rng(0)
    P = rng.normal(0, 0.05, size=(3, 3, 8, 16)).astype(np.float32)
    s = rng.uniform(1e-3, 1e-2, size=(1, 1, 8, 1)).astype(np.float32)
    w = rng.normal(0, 1e-3, size=P.shape).astype(np.float32)
    Pt = torch.tensor(P, device=dev, requires_grad=True)
    st = torch.tensor(s, device=dev, requires_grad=True)
    out = lq.my_custom_gradient(Pt, st, 1e-3)
    (out * torch.tensor(w, device=dev)).sum().backward()
    _, ds_o = O.nq_backward(P, s, 1e-3, w)
    assert torch.equal(Pt.grad.cpu(), torch.tensor(w))                 # dP == dy bit for bit (STE)
    np.testing.assert_allclose(st.grad.cpu().numpy(), ds_o, rtol=RTOL)
    # STE-only variant: ds == 0
    Pt.grad = None
    st.grad = None
    out = lq.my_custom_gradient(Pt, st)
    (out * torch.tensor(w, device=dev)).sum().backward()
    assert torch.equal(Pt.grad.cpu(), torch.tensor(w))
    assert torch.count_nonzero(st.grad) == 0 and st.grad.shape == st.shape


def test_dense_layer_forward_backward(dev):
    import learned_quantization_amd as lq
    lq.reset_layer_names()
    layer = lq.CustomDenseLayer(seed=42, units=10, penalty_threshold=1e-3, orientation="rowwise",
                                initializer=lq.RandomNormal(seed=42), name="custom_dense_layer",
                                regularizer=None, input_shape=(None, 20), device=dev)
    assert layer.name == "custom_dense_layer" and tuple(layer.W.shape) == (20, 10)
    assert tuple(layer.nested_q_w_layer.scale.shape) == (20, 1) and tuple(layer.nested_q_b_layer.scale.shape) == (1,)
    with torch.no_grad():
        layer.nested_q_w_layer.scale.fill_(0.01)
        layer.nested_q_b_layer.scale.fill_(0.02)
    x = torch.randn(7, 20, device=dev)
    y = layer(x)
    W, b = layer.W.detach().cpu().numpy(), layer.b.detach().cpu().numpy()
    _, qW = O.fq_forward(W, np.full((20, 1), 0.01, np.float32))
    _, qb = O.fq_forward(b, np.full((1,), 0.02, np.float32))
    np.testing.assert_allclose(y.detach().cpu().numpy(), x.cpu().numpy() @ qW + qb, rtol=1e-4, atol=1e-5)
    y.square().sum().backward()
    dW = layer.W.grad.cpu().numpy()
    _, ds_o = O.nq_backward(W, np.full((20, 1), 0.01, np.float32), 1e-3, dW)
    np.testing.assert_allclose(layer.nested_q_w_layer.scale.grad.cpu().numpy(), ds_o, rtol=RTOL)
    _, dsb_o = O.nq_backward(b, np.full((1,), 0.02, np.float32), 1e-3, layer.b.grad.cpu().numpy())
    np.testing.assert_allclose(layer.nested_q_b_layer.scale.grad.cpu().numpy(), dsb_o, rtol=RTOL)


@pytest.mark.parametrize("padding,strides", [("same", (1, 1)), ("valid", (1, 1)), ("same", (2, 2))])
def test_conv_layer_matches_reference_layout(dev, padding, strides):
    import learned_quantization_amd as lq
    lq.reset_layer_names()
    layer = lq.CustomConv2DLayer(seed=1, penalty_threshold=1e-11, orientation="channelwise",
                                 initializer=lq.RandomNormal(seed=1), filters=6, kernel_size=(3, 3), strides=strides,
                                 padding=padding, name="x", regularizer=lq.l2(1e-4), input_shape=4, device=dev)
    assert layer.name == "custom_conv2d_layer" and layer.padding == padding.upper()
    assert tuple(layer.kernel.shape) == (3, 3, 4, 6) and tuple(layer.nested_q_k_layer.scale.shape) == (1, 1, 4, 1)
    with torch.no_grad():
        layer.nested_q_k_layer.scale.fill_(0.004)
        layer.nested_q_b_layer.scale.fill_(0.003)
    x = torch.randn(2, 4, 9, 9, device=dev)
    y = layer(x)
    k, b = layer.kernel.detach().cpu().numpy(), layer.b.detach().cpu().numpy()
    _, qk = O.fq_forward(k, np.full((1, 1, 4, 1), 0.004, np.float32))
    _, qb = O.fq_forward(b, np.full((1,), 0.003, np.float32))
    # TF-semantics reference conv on CPU (explicit SAME padding)
    xt = x.cpu()
    wt = torch.tensor(qk).permute(3, 2, 0, 1)
    if padding == "same":
        import math
        pads = []
        for n, kk, ss in ((9, 3, strides[0]), (9, 3, strides[1])):
            tot = max((math.ceil(n / ss) - 1) * ss + kk - n, 0)
            pads.append((tot // 2, tot - tot // 2))
        xt = torch.nn.functional.pad(xt, (pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    ref = torch.nn.functional.conv2d(xt, wt, None, strides) + torch.tensor(qb).view(1, -1, 1, 1)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    y.sum().backward()
    _, ds_o = O.nq_backward(k, np.full((1, 1, 4, 1), 0.004, np.float32), 1e-11, layer.kernel.grad.cpu().numpy())
    np.testing.assert_allclose(layer.nested_q_k_layer.scale.grad.cpu().numpy(), ds_o, rtol=RTOL)
    assert layer.regularization_loss() is not None


def _mk_layers(dev, orient="channelwise"):
    import learned_quantization_amd as lq
    from learned_quantization_amd.custom_loss_terms import custom_layers as CL
    lq.reset_layer_names()
    layers = []
    for ci, co in ((3, 8), (8, 16)):
        l = CL.CustomConv2DLayer(seed=0, penalty_rate=1e-7, orientation=orient, initializer=lq.RandomNormal(seed=ci),
                                 filters=co, kernel_size=(3, 3), strides=(1, 1), padding="same", name="c",
                                 regularizer=None, input_shape=ci, device=dev)
        with torch.no_grad():
            l.nested_q_k_layer.scale.uniform_(1e-3, 1e-2)
            l.nested_q_b_layer.scale.uniform_(1e-3, 1e-2)
        layers.append(l)
    return layers


@pytest.mark.parametrize("orient", ["rowwise", "columnwise", "channelwise", "scalar"])
@pytest.mark.parametrize("kind", ["maxbin", "difference", "inverse"])
def test_loss_terms_value_and_gradients(dev, kind, orient):
    import learned_quantization_amd as lq
    layers = _mk_layers(dev, orient)
    cls = {"maxbin": lq.SCCEMaxBin, "difference": lq.SCCEDifference, "inverse": lq.SCCEInverse}[kind]
    import tempfile
    loss_obj = cls(layers, penalty_rate=0.5, log_dir=tempfile.mkdtemp())
    pen = getattr(loss_obj, f"compute_{kind}_penalty")()
    np_layers = [(l.kernel.detach().cpu().numpy(), l.nested_q_k_layer.scale.detach().cpu().numpy(),
                  l.b.detach().cpu().numpy(), l.nested_q_b_layer.scale.detach().cpu().numpy()) for l in layers]
    oracle_fn = {"maxbin": O.maxbin_penalty, "difference": O.difference_penalty, "inverse": O.inverse_penalty}[kind]
    assert float(pen) == pytest.approx(float(oracle_fn(np_layers)), rel=RTOL)
    # total loss + gradients against torch-CPU autograd of the op-for-op restatement
    y_true = torch.tensor([1, 0, 3], device=dev)
    y_pred = torch.softmax(torch.randn(3, 5, device=dev), dim=1)
    total = loss_obj.compute_total_loss(y_true, y_pred)
    assert total.shape == (3,)
    np.testing.assert_allclose(total.detach().cpu().numpy(),
                               O.total_loss(y_true.cpu().numpy(), y_pred.cpu().numpy(), 0.5, oracle_fn(np_layers)),
                               rtol=RTOL)
    total.mean().backward()
    # gradients against the float64 oracle within 1e-5 * sum|terms| (tests/_bounds.py); CL-F:75-116, 161-195, 240-275
    l64 = [(k, ks, O.group_descriptor(k.shape, ks.shape), b, bs, O.group_descriptor(b.shape, bs.shape)) for k, ks, b, bs in np_layers]
    val64 = O64.penalty(kind, l64)
    assert_within_terms(float(pen), val64, abs(val64), f"{kind} {orient} penalty value")          # every term is >= 0
    g64 = O64.penalty_grads(kind, l64, 0.5)
    tl = [tuple(torch.tensor(a, requires_grad=True) for a in l) for l in np_layers]
    fn_t = {"maxbin": OT.maxbin_penalty, "difference": OT.difference_penalty, "inverse": OT.inverse_penalty}[kind]
    (0.5 * fn_t(tl)).backward()              # float32 torch-CPU autograd: decides the MaxBin tie split like TF's float32 autodiff
    for l, e, (k, ks, b, bs) in zip(layers, g64, tl):
        assert_within_terms(l.nested_q_k_layer.scale.grad.cpu().numpy(), e["dsK"], e["dsK_abs"], f"{kind} {orient} ds_k")
        assert_within_terms(l.nested_q_b_layer.scale.grad.cpu().numpy(), e["dsb"], e["dsb_abs"], f"{kind} {orient} ds_b")
        if kind == "difference":
            assert_within_terms(l.kernel.grad.cpu().numpy(), e["dK"], e["dK_abs"], f"{kind} {orient} dK")
            assert_within_terms(l.b.grad.cpu().numpy(), e["db"], e["db_abs"], f"{kind} {orient} db")
        elif kind == "maxbin":
            np.testing.assert_allclose(l.kernel.grad.cpu().numpy(), k.grad.numpy(), rtol=1e-5, atol=0, err_msg=f"{kind} {orient} dK")
            np.testing.assert_allclose(l.b.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=0, err_msg=f"{kind} {orient} db")
        else:
            assert l.kernel.grad is None and l.b.grad is None


def test_scale_adam_matches_keras_restatement(dev):
    import learned_quantization_amd as lq
    rng = np.random.default_rng(0)
    s0 = np.full(64, O.SCALE_INIT, np.float32)
    p = torch.nn.Parameter(torch.tensor(s0, device=dev))
    p.lq_constraint = lq.MinValueConstraint(lq.SCALE_INIT)
    opt = lq.ScaleAdam([p], lr=1e-4)
    s, m, v = s0.copy(), np.zeros(64, np.float32), np.zeros(64, np.float32)
    for step in range(1, 6):
        g = (rng.normal(0, 1.0, size=64) * (10.0 ** rng.integers(-6, 1))).astype(np.float32)
        p.grad = torch.tensor(g, device=dev)
        opt.step()
        s, m, v = O.keras_adam_step(s, g, m, v, step, lr=1e-4, min_value=O.SCALE_MIN)
        np.testing.assert_allclose(p.detach().cpu().numpy(), s, rtol=2e-6, atol=0)
        assert float(p.min()) >= O.SCALE_MIN
    # bare constraint
    c = lq.MinValueConstraint(0.5)
    w = torch.tensor([0.1, 0.5, 0.9, float("nan")], device=dev)
    got = c(w).cpu().numpy()
    np.testing.assert_array_equal(got[:3], np.array([0.5, 0.5, 0.9], np.float32))
    assert np.isnan(got[3]) and c.get_config() == {"min_value": 0.5}


@pytest.mark.parametrize("shape,orient,axis", [
    ((3, 3, 8, 16), "channelwise", 1), ((784, 128), "rowwise", 1), ((784, 128), "columnwise", 1), ((784, 128), "scalar", 0),
    ((3, 3, 64, 128), "channelwise", 1), ((3, 3, 64, 128), "rowwise", 2), ((7, 7, 3, 64), "columnwise", 1),
    ((1, 1, 256, 1024), "channelwise", 1), ((3, 3, 512, 512), "channelwise", 1), ((3, 3, 512, 512), "scalar", 3),
    ((128, 10), "rowwise", 1), ((10,), "scalar", 0), ((5, 70000), "rowwise", 1), ((70000, 5), "columnwise", 0)])
def test_callback_statistic_absmax_over_axis(dev, shape, orient, axis):
    """np.max(np.abs(floor(k/s)), axis=1) of custom_callbacks.py:98-99 (and any other axis) -- exact: integer atomics on the
    bit pattern of |q|.  Covers the LDS-table path, the direct path (post > table), the one-output-per-wave path (post == 1)."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(0)
    k = rng.normal(0, 0.05, size=shape).astype(np.float32)
    s = rng.uniform(1e-3, 1e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
    got = lq.q_absmax_over_axis(torch.tensor(k, device=dev), torch.tensor(s, device=dev), axis=axis).cpu().numpy()
    want = np.max(np.abs(O.quantized_integers(k, s)), axis=axis)     # custom_callbacks.py:98
    np.testing.assert_array_equal(got, want)
    if len(shape) > 1:       # NaN propagates like np.max; all-zero slices give 0
        k2 = k.copy()
        k2[0] = 0.0
        k2.reshape(-1)[-1] = np.nan
        got = lq.q_absmax_over_axis(torch.tensor(k2, device=dev), torch.tensor(s, device=dev), axis=axis).cpu().numpy()
        with np.errstate(all="ignore"):
            want = np.max(np.abs(O.quantized_integers(k2, s)), axis=axis)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("shape,orient,smag", [((3, 3, 8, 16), "channelwise", 1e-2), ((784, 128), "rowwise", 1.1920929e-05),
                                               ((50,), "scalar", 1e-3), ((64, 3, 31, 31), "columnwise", 0.7)])
def test_q_unique_matches_numpy_unique(dev, shape, orient, smag):
    import learned_quantization_amd as lq
    rng = np.random.default_rng(4)
    P = rng.normal(0, 0.05, size=shape).astype(np.float32)
    if len(shape) == 4 and shape[0] == 64:
        P = (P * 2000).astype(np.float32)
    s = (rng.uniform(0.5, 2.0, size=O.scale_shape(shape, orient)) * smag).astype(np.float32)
    vals, counts = lq.q_unique(torch.tensor(P, device=dev), torch.tensor(s, device=dev))
    u, c = np.unique(O.quantized_integers(P, s), return_counts=True)      # custom_callbacks.py:92
    np.testing.assert_array_equal(vals.cpu().numpy(), u.astype(np.int32))
    np.testing.assert_array_equal(counts.cpu().numpy(), c)
    assert int(counts.sum()) == P.size


def test_keras_adam_multi_tensor_matches_restatement(dev):
    """lq_adam_set_step == oracle.keras_adam_step for a set of odd-sized tensors; a parameter without gradient is skipped;
    torch mode == torch.optim.Adam."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(2)
    shapes = [(5,), (17, 3), (3, 3, 8, 16), (1,), (1025,), (2048, 7)]
    ps = [torch.nn.Parameter(torch.tensor(rng.normal(0, 0.05, size=s).astype(np.float32), device=dev)) for s in shapes]
    ref = [p.detach().cpu().numpy().copy() for p in ps]
    ms = [np.zeros_like(r) for r in ref]
    vs = [np.zeros_like(r) for r in ref]
    opt = lq.KerasAdam(ps, lr=1e-3)
    for step in range(1, 5):
        for i, p in enumerate(ps):
            if i == 3 and step == 2:
                p.grad = None                      # skipped this step
                continue
            g = (rng.normal(0, 1.0, size=shapes[i]) * 10.0 ** rng.integers(-4, 1)).astype(np.float32)
            p.grad = torch.tensor(g, device=dev)
            # Keras applies the global iteration count to every variable
            ref[i], ms[i], vs[i] = O.keras_adam_step(ref[i], g, ms[i], vs[i], step, lr=1e-3)
        opt.step()
        for i, p in enumerate(ps):
            np.testing.assert_allclose(p.detach().cpu().numpy(), ref[i], rtol=3e-6, atol=1e-9, err_msg=f"step {step} tensor {i}")
    # torch arithmetic
    pa = [torch.nn.Parameter(torch.tensor(r, device=dev)) for r in ref]
    pb = [torch.nn.Parameter(torch.tensor(r, device=dev)) for r in ref]
    oa, ob = lq.KerasAdam(pa, lr=1e-3, eps=1e-8, mode="torch"), torch.optim.Adam(pb, lr=1e-3, eps=1e-8)
    for step in range(3):
        for a, b in zip(pa, pb):
            g = torch.tensor(rng.normal(0, 1e-2, size=tuple(a.shape)).astype(np.float32), device=dev)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("sval", [1.0 - 2.0 ** -24, 1.0, 1.0 + 2.0 ** -23])
@pytest.mark.parametrize("orientation", ["scalar", "rowwise"])
def test_difference_sign_is_the_sign_of_the_float32_residual(dev, sval, orientation):
    """custom_loss_functions.py:172-176 runs in float32: ``sign(P - P/s)`` is the sign of the ROUNDED residual.  With ``s``
    within an ulp of 1 the rounded quotient IS ``P`` for every element whose half-ulp exceeds ``|P/s - P|`` (all of them at
    s = 1; the denormals at 1 -+ ulp): the reference routes gradient 0 there, a float64 sign would route +-c (1 - 1/s).
    The float64 oracle takes that sign from float32 since commit b10f190; this case pins the decision with the float32 oracle
    (oracle/lq_oracle.py::difference_term_grads, the op-for-op restatement): element for element, exactly 0 where the float32
    residual is 0, the float32 oracle's value elsewhere."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(int(sval * 2 ** 24) & 0xffff)
    rows, cols = 48, 2048
    mant = rng.uniform(1.0, 2.0, size=(rows, cols))
    expo = rng.integers(-100, 100, size=(rows, cols))
    P = (np.ldexp(mant, expo) * rng.choice([-1.0, 1.0], size=(rows, cols))).astype(np.float32)
    P[:, :64] = (rng.integers(1, 5000, size=(rows, 64)) * 2.0 ** -149).astype(np.float32) * rng.choice([-1.0, 1.0], size=(rows, 64))   # denormals
    P[:, 64:80] = 0.0
    P[:, 80:96] = np.float32(2.0 ** -126) * rng.choice([-1.0, 1.0], size=(rows, 16))       # smallest normals
    if orientation == "scalar":
        s = np.array([sval], np.float32)
    else:       # row-wise: the three scales of this test and two ordinary ones side by side
        s = rng.choice(np.array([1.0 - 2.0 ** -24, 1.0, 1.0 + 2.0 ** -23, 0.75, 3.0], np.float32), size=(rows, 1)).astype(np.float32)
        s[0, 0] = sval
    c = 0.7
    Pt = torch.tensor(P, device=dev, requires_grad=True)
    st = torch.tensor(s, device=dev, requires_grad=True)
    (lq.difference_term(Pt, st) * c).backward()
    dp = Pt.grad.cpu().numpy()
    dp_o, ds_o = O.difference_term_grads(P, s, c)
    full = np.broadcast_to(s if s.ndim == 2 else s.reshape(1, 1), P.shape)
    resid = (P - (P / full).astype(np.float32)).astype(np.float32)
    zero = resid == 0
    if orientation == "scalar":
        assert zero[:, :80].all()                  # denormals and zeros: the rounded quotient is P itself
    f64_would_differ = zero & (P != 0) & (full != 1.0)
    if sval != 1.0:
        assert f64_would_differ.any(), "the case must contain elements where a float64 sign disagrees"
    assert (dp[zero] == 0).all(), f"{int((dp[zero] != 0).sum())} elements with a zero float32 residual received a gradient"
    np.testing.assert_array_equal(np.sign(dp[~zero]), np.sign(dp_o[~zero]))
    np.testing.assert_allclose(dp[~zero], dp_o[~zero], rtol=1e-6, atol=0)
    # ds: the same float32 signs enter the per-group sums (bounded as everywhere, against the float64 oracle that shares them)
    desc = O.group_descriptor(P.shape, s.shape)
    _, ds64, ds_abs = O64.difference_term_grads(P, s, c, *desc)
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, "difference ds near s = 1")
