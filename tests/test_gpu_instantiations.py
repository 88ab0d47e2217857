"""Every kernel instantiation the library ships is launched by a parity case.

`tools/kernel_coverage.py` compares the kernel names of a rocprofv3 trace of the GPU suite with the device stubs of
liblq_hip.so; the first such run (profiles/r03/kernel_coverage_before.txt) found 171 of 340 instantiations that no test
launched.  Those that no descriptor can reach were removed from the dispatch (csrc/lq_kernels.hip: kStreamOp, the constexpr
branches of launch_stream2_impl); the others are reached here, each case against the oracle like every other parity test
(q, out, max|q| bit-exact; mean and ds within 1e-5: custom_layers.py:55-60, 62-118).

What decides the instantiation (lq_kernels.hip: make_plan, launch_stream2_impl, launch_traverse):
  * NT: nontemporal accesses from 64 MiB per tensor (16 777 216 elements) -- most cases below are the round-2 streaming shapes
    of test_gpu_parity.py::STREAM2_SHAPES again, at 17 M elements instead of 4-5 M;
  * row length / column count: team width and float4 per lane of k_row_tiny / k_row_win, flat or periodic or tiled columns;
  * outer == 1 or not: direct emit vs group-major partials (the GM parameter of k_row_tiny).
"""
import numpy as np
import pytest
import torch

import ctypes

from _bounds import stable_seed
from conftest import as_f32p
from oracle import lq_oracle as O
from test_gpu_parity import _t, dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

NT = 16_777_216          # elements from which a tensor is streamed with nontemporal accesses (kNtBytes / 4)


def _rows(n, L):
    return -(-n // L)


# (shape, orientation, lambda, what it reaches)
CASES = [
    # ---- k_row_tiny<OP, NT, 2, LG, GM>: rows of 8..64 elements on the 16-byte grid; LG = log2(lanes per row), GM 0 = outer == 1
    ((_rows(NT, 8) + 3, 8), "rowwise", 1e-10, "tiny rows LG 1, nt, direct emit"),
    ((2, _rows(NT, 16) + 1, 8), "columnwise", 3e-2, "tiny rows LG 1, nt, outer > 1"),
    ((_rows(NT, 16) + 3, 16), "rowwise", 3e-2, "tiny rows LG 2, nt"),
    ((2, _rows(NT, 32) + 1, 16), "columnwise", 1e-10, "tiny rows LG 2, nt, outer > 1"),
    ((_rows(NT, 32) + 3, 32), "rowwise", 1e-10, "tiny rows LG 3, nt"),
    ((2, _rows(NT, 64) + 1, 32), "columnwise", 3e-2, "tiny rows LG 3, nt, outer > 1"),
    ((_rows(NT, 64) + 3, 64), "rowwise", 3e-2, "tiny rows LG 4, nt"),
    ((2, _rows(NT, 128) + 1, 64), "columnwise", 1e-10, "tiny rows LG 4, nt, outer > 1"),
    ((2, 300001, 8), "columnwise", 1e-10, "tiny rows LG 1, outer > 1, default policy"),
    ((2, 150001, 16), "columnwise", 3e-2, "tiny rows LG 2, outer > 1, default policy"),
    ((2, 40001, 64), "columnwise", 1e-10, "tiny rows LG 4, outer > 1, default policy"),
    # ---- k_row_win<OP, NT, LG, V, U>: aligned float4 windows, teams of 2..64 lanes, 1..5 float4 per lane
    ((_rows(NT, 5) + 1, 5), "rowwise", 1e-10, "row windows LG 1 (rows of 5), nt"),
    ((_rows(NT, 10) + 1, 10), "rowwise", 3e-2, "row windows LG 2 (rows of 10), nt"),
    ((_rows(NT, 26) + 1, 26), "rowwise", 1e-10, "row windows LG 3 (rows of 26), nt"),
    ((_rows(NT, 58) + 1, 58), "rowwise", 3e-2, "row windows LG 4 (rows of 58), nt"),
    ((_rows(NT, 122) + 1, 122), "rowwise", 1e-10, "row windows LG 5 (rows of 122), nt"),
    ((_rows(NT, 250) + 1, 250), "rowwise", 3e-2, "row windows 64 lanes x 1 (rows of 250), nt"),
    ((_rows(NT, 500) + 1, 500), "rowwise", 1e-10, "row windows 64 lanes x 2 (rows of 500, on the grid), nt"),
    ((3, _rows(NT, 3 * 750) + 1, 750), "columnwise", 3e-2, "row windows 64 lanes x 3 (rows of 750), nt, outer > 1"),
    ((_rows(NT, 1023) + 1, 1023), "rowwise", 1e-10, "row windows 64 lanes x 5 (rows of 1023), nt; K1 straddling flat stream, nt"),
    ((420001, 10), "rowwise", 1e-10, "row windows LG 2, default policy"),
    ((162001, 26), "rowwise", 3e-2, "row windows LG 3, default policy"),
    ((34501, 122), "rowwise", 1e-10, "row windows LG 5, default policy"),
    # round 3: teams of 16 / 32 lanes with two or three float4 per lane for rows of 65..340 elements (K2 and K4 choose differently)
    ((_rows(NT, 68) + 1, 68), "rowwise", 1e-10, "rows of 68, nt: 16 lanes x 2 (K2 and K4)"),
    ((_rows(NT, 100) + 1, 100), "rowwise", 3e-2, "rows of 100, nt: K2 16 lanes x 2, K4 32 lanes x 1"),
    ((2, _rows(NT, 2 * 132) + 1, 132), "columnwise", 1e-10, "rows of 132, nt, outer > 1: K2 16 lanes x 3, K4 32 lanes x 2"),
    ((_rows(NT, 200) + 1, 200), "rowwise", 3e-2, "rows of 200, nt: 32 lanes x 2 (K2 and K4)"),
    ((_rows(NT, 300) + 1, 300), "rowwise", 1e-10, "rows of 300, nt: K2 32 lanes x 3, K4 64 lanes x 2"),
    ((62001, 68), "rowwise", 3e-2, "rows of 68, default policy: 16 lanes x 2"),
    ((42101, 100), "rowwise", 1e-10, "rows of 100, default policy: K2 16 lanes x 2"),
    ((32001, 131), "rowwise", 3e-2, "rows of 131 (off the grid), default policy: K2 16 lanes x 3, K4 32 lanes x 2"),
    ((2, 10551, 200), "columnwise", 1e-10, "rows of 200, default policy, outer > 1: 32 lanes x 2"),
    ((14101, 300), "rowwise", 3e-2, "rows of 300, default policy: K2 32 lanes x 3"),
    # rows of 1153..1945 elements as one wave per row (two 1024-chunks otherwise): 5..8 float4 per lane
    ((_rows(NT, 1200) + 1, 1200), "rowwise", 1e-10, "one wave per row, 5 float4 per lane, nt"),
    ((_rows(NT, 1500) + 1, 1500), "rowwise", 3e-2, "one wave per row, 6 float4 per lane, nt"),
    ((_rows(NT, 1601) + 1, 1601), "rowwise", 1e-10, "one wave per row, 7 float4 per lane (off the grid), nt"),
    ((2, _rows(NT, 2 * 1801) + 1, 1801), "columnwise", 3e-2, "one wave per row, 8 float4 per lane, nt, outer > 1"),
    # ---- k_row_seg<OP, 1, 2>: short rows off the grid whose team would be poorly filled
    ((_rows(NT, 17) + 1, 17), "rowwise", 1e-10, "row segments (rows of 17), nt"),
    # ---- columns
    ((_rows(NT, 3) + 1, 3), "columnwise", 1e-10, "C = 3 (NHWC), nt: gathered-scale flat K1 (mode 10), periodic K2 / K4 with two float4"),
    ((_rows(NT, 9) + 1, 3, 3), "columnwise", 3e-2, "C = 9 with inner = 3, nt: flat K1 mode 11"),
    ((_rows(NT, 48) + 1, 48), "columnwise", 1e-10, "C = 48, nt: periodic K4 with one float4 per stream"),
    ((_rows(NT, 64) + 1, 64), "columnwise", 3e-2, "C = 64, nt: one-shot flat column kernels (K1, K2 with four float4, K4)"),
    ((_rows(NT, 8) + 1, 8), "columnwise", 1e-10, "C = 8, nt: one-shot flat column kernels"),
    ((_rows(NT, 1024) + 1, 1024), "columnwise", 1e-10, "C = 1024, nt: pipelined column tile, whole lines per row"),
    ((_rows(NT, 1000) + 1, 1000), "columnwise", 3e-2, "C = 1000, nt: pipelined column tile with XCD-aware block order (K2)"),
    # ---- row stream, the forms the BENCH tensor does not take
    ((3400, 5000), "rowwise", 1e-10, "rows of 5000, nt: 1024-element chunks fill the rows better (256-thread units)"),
    ((3400, 5001), "rowwise", 1e-10, "rows of 5001, nt: the same off the 16-byte grid (TAIL instantiation)"),
    ((1100, 4096), "rowwise", 3e-2, "rows of 4096 below the nontemporal size: 512-thread units, default policy"),
    ((4200, 4100), "rowwise", 3e-2, "rows of 4100, nt, lambda >= 4e-4: two float4 per thread with a folded tail (K4)"),
    # ---- k_finalize_cols<OP>: the column form of the finalize (>= 131072 partials, at most 64 row blocks; lq_kernels.hip)
    ((4096, 4096), "columnwise", 1e-10, "column finalize of a wide column-wise matrix, K2 and K4"),
    ((2, 2048, 4096, 1), "channelwise", 3e-2, "column finalize, groups of one column in two layers"),
]


def _check_case_c(c_oracle, Pt, st, dyt, lam, name):
    """K1 (with the integer view), K2 + K3 (with their parts) and K4 of device tensors against the scalar C restatement
    (oracle/lq_oracle.c: the NumPy oracle takes 18 s for 17 M elements, the C loop 2.5 s; tests/test_oracle.py holds the two to
    each other).  Bars as everywhere: q, out, max|q| bit-exact, mean and ds within 1e-5."""
    import learned_quantization_amd as lq
    P, s, dy = (np.ascontiguousarray(t.cpu().numpy()) for t in (Pt, st, dyt))
    desc = O.group_descriptor(P.shape, s.shape)
    n, G = P.size, desc[1]
    out_c, q_c = np.empty(n, np.float32), np.empty(n, np.float32)
    c_oracle.lqo_fq_forward(as_f32p(P.reshape(-1)), as_f32p(s.reshape(-1)), as_f32p(out_c), as_f32p(q_c), *desc)
    ds_c, maxq_c, mean_c = (np.empty(G, np.float32) for _ in range(3))
    below = np.empty(G, np.int64)
    c_oracle.lqo_nq_scale_grad(as_f32p(P.reshape(-1)), as_f32p(s.reshape(-1)), as_f32p(dy.reshape(-1)), np.float32(lam), as_f32p(ds_c),
                               as_f32p(maxq_c), as_f32p(mean_c), below.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), *desc)
    out, q = lq.fq_forward(Pt, st, q_dtype=torch.float32)
    assert np.array_equal(q.cpu().numpy().reshape(-1), q_c), f"{name}: q"
    assert np.array_equal(out.cpu().numpy().reshape(-1), out_c), f"{name}: out"
    del out, q
    ds, parts = lq.fq_scale_grad(Pt, st, dyt, lam, return_parts=True)
    parts = parts.cpu().numpy()
    np.testing.assert_array_equal(parts[0], maxq_c, err_msg=f"{name}: max|q|")
    np.testing.assert_allclose(parts[1], mean_c, rtol=1e-5, atol=1e-30, equal_nan=True, err_msg=f"{name}: mean")
    np.testing.assert_allclose(ds.cpu().numpy().reshape(-1), ds_c, rtol=1e-5, atol=1e-30, equal_nan=True, err_msg=f"{name}: ds")
    out2, ds2 = lq.fq_fwd_bwd_fused(Pt, st, dyt, lam)
    assert np.array_equal(out2.cpu().numpy().reshape(-1), out_c), f"{name}: fused out"
    np.testing.assert_allclose(ds2.cpu().numpy().reshape(-1), ds_c, rtol=1e-5, atol=1e-30, equal_nan=True, err_msg=f"{name}: fused ds")


@pytest.mark.parametrize("shape,orient,lam,what", CASES, ids=[c[3].split(":")[0].split(",")[0] + f" #{i}" for i, c in enumerate(CASES)])
def test_instantiation_parity(shape, orient, lam, what, dev, c_oracle):  # noqa: F811
    # inputs drawn on the device (seeded; 17 M normals and powers take 10 s in NumPy): weights-like P, gradients over 7 decades
    g = torch.Generator(device=dev).manual_seed(stable_seed(shape, orient) % (2 ** 31))
    P = torch.randn(shape, device=dev, generator=g) * 0.05
    dy = torch.randn(shape, device=dev, generator=g) * torch.pow(10.0, torch.rand(shape, device=dev, generator=g) * 7.0 - 9.0)
    sshape = O.scale_shape(shape, orient)
    s = torch.rand(sshape, device=dev, generator=g) * 0.029 + 1e-3
    _check_case_c(c_oracle, P, s, dy, lam, f"{shape} {orient} lam={lam} ({what})")


def test_penalty_terms_through_the_column_finalize(dev):  # noqa: F811
    """MaxBin forward (k_finalize_cols<OP_MAXBIN_FWD>) and Difference backward (<OP_DIFF_BWD>) on a 1024 x 2048 column-wise
    matrix: 64 row blocks x 2048 columns of partials -- 131072, where the column form starts -- read as whole rows
    (custom_loss_functions.py:90-110, 172-195)."""
    import learned_quantization_amd as lq
    from oracle import lq_oracle_f64 as O64
    from _bounds import assert_within_terms
    rng = np.random.default_rng(8)
    shape = (1024, 2048)
    Pn = rng.normal(0, 0.05, size=shape).astype(np.float32)
    sn = rng.uniform(1e-3, 3e-2, size=(1, 2048)).astype(np.float32)
    desc = O.group_descriptor(shape, sn.shape)
    Pt = _t(Pn, dev).requires_grad_(True)
    st = _t(sn, dev).requires_grad_(True)
    mb = lq.maxbin_term(Pt, st)
    assert_within_terms(float(mb), O64.maxbin_term(Pn, sn, *desc), O64.term_abs("maxbin", Pn, sn, *desc), "maxbin value")
    (mb * 0.3).backward()
    _, ds64, ds_abs = O64.maxbin_term_grads(Pn, sn, 0.3, *desc)
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, "maxbin ds")
    assert int(torch.count_nonzero(Pt.grad)) >= 2048, "one maximum per column at least"
    Pt.grad = None
    st.grad = None
    df = lq.difference_term(Pt, st)
    assert_within_terms(float(df), O64.difference_term(Pn, sn, *desc), O64.term_abs("difference", Pn, sn, *desc), "difference value")
    (df * 0.7).backward()
    dp64, ds64, ds_abs, dp_abs = O64.difference_term_grads(Pn, sn, 0.7, *desc, with_dP_abs=True)
    assert_within_terms(Pt.grad.cpu().numpy(), dp64, dp_abs, "difference dP")
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, "difference ds")


def _misaligned(a, dev):
    """The array on the device at a base that is 4 bytes off the 16-byte grid (scalar forms of the traversals)."""
    base = torch.empty(a.size + 1, dtype=torch.float32, device=dev)
    v = base[1:].view(a.shape)
    v.copy_(torch.from_numpy(a))
    assert v.data_ptr() % 16 == 4 and v.is_contiguous()
    return v


def test_weight_sized_operations_on_long_misaligned_rows(dev):  # noqa: F811
    """Penalty terms (custom_loss_functions.py:90-110, 172-176), the integer view (custom_callbacks.py:85-87) and their
    gradients on rows of >= 1024 elements whose base is off the 16-byte grid: the scalar row-stream form of every operation."""
    import learned_quantization_amd as lq
    from oracle import lq_oracle_f64 as O64
    from _bounds import assert_within_terms
    rng = np.random.default_rng(5)
    shape = (5, 1031)
    Pn = rng.normal(0, 0.05, size=shape).astype(np.float32)
    sn = rng.uniform(1e-3, 3e-2, size=(5, 1)).astype(np.float32)
    desc = O.group_descriptor(shape, sn.shape)
    P = _misaligned(Pn, dev)
    s = _t(sn, dev)
    q = lq.quantized_integers(P, s, torch.int32).cpu().numpy()
    np.testing.assert_array_equal(q, O.quantized_integers(Pn, sn).astype(np.int32))
    Pt = P.detach().requires_grad_(True)           # same (misaligned) storage
    st = s.detach().clone().requires_grad_(True)
    mb = lq.maxbin_term(Pt, st)
    assert_within_terms(float(mb), O64.maxbin_term(Pn, sn, *desc), O64.term_abs("maxbin", Pn, sn, *desc), "maxbin value")
    df = lq.difference_term(Pt, st)
    assert_within_terms(float(df), O64.difference_term(Pn, sn, *desc), O64.term_abs("difference", Pn, sn, *desc), "difference value")
    (mb * 0.3).backward()
    dp32, _ = O.maxbin_term_grads(Pn, sn, 0.3)
    _, ds64, ds_abs = O64.maxbin_term_grads(Pn, sn, 0.3, *desc)
    np.testing.assert_allclose(Pt.grad.cpu().numpy(), dp32, rtol=1e-5, atol=0)
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, "maxbin ds")
    Pt.grad = None
    st.grad = None
    (df * 0.7).backward()
    dp64, ds64, ds_abs, dp_abs = O64.difference_term_grads(Pn, sn, 0.7, *desc, with_dP_abs=True)
    assert_within_terms(Pt.grad.cpu().numpy(), dp64, dp_abs, "difference dP")
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, "difference ds")


@pytest.mark.parametrize("shape,orient,misalign", [
    ((7, 7, 3, 66), "scalar", False),       # 9702 elements in one row, L % 4 == 2: float4 row stream with scalar head / tail
    ((7, 7, 3, 64), "scalar", True),        # the same kernel class off the 16-byte grid: scalar row stream
    ((64, 64, 3, 8), "channelwise", False),  # 4096 taps x 3 x 8: column mode (C = 24) of the element-wise companion
])
def test_elementwise_oihw_companion_forms(shape, orient, misalign, dev):  # noqa: F811
    """lq_fq_forward_oihw / lq_fq_scale_grad_oihw on conv kernels the LDS tile does not take (more than 9 taps), in the
    row-stream and column forms of the element-wise companion (custom_layers.py:321, 338-350)."""
    import learned_quantization_amd as lq
    from learned_quantization_amd import ops
    rng = np.random.default_rng(stable_seed(shape, orient))
    kn = rng.normal(0, 0.05, size=shape).astype(np.float32)
    sn = rng.uniform(1e-3, 1e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
    k = _misaligned(kn, dev) if misalign else _t(kn, dev)
    s = _t(sn, dev)
    out, out_oihw = ops.fq_forward_oihw(k, s)
    _, out_o = O.fq_forward(kn, sn)
    np.testing.assert_array_equal(out.cpu().numpy(), out_o)
    np.testing.assert_array_equal(out_oihw.cpu().numpy(), np.transpose(out_o, (3, 2, 0, 1)))
    dyn = (rng.normal(0, 1, size=shape) * 10.0 ** rng.uniform(-8, -2, size=shape)).astype(np.float32)
    dy_oihw = _t(np.ascontiguousarray(np.transpose(dyn, (3, 2, 0, 1))), dev)
    for lam in (1e-10, 2e-2):
        ds, dP = ops.fq_scale_grad_oihw(k, s, dy_oihw, lam)
        _, ds_o = O.nq_backward(kn, sn, lam, dyn)
        np.testing.assert_array_equal(dP.cpu().numpy(), dyn)
        np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=1e-5, atol=1e-30)
