"""Seeded randomized GPU-vs-oracle sweep: shapes, ranks, orientations, scale magnitudes, special values,
misaligned views -- every op of the C ABI.  Bit-exact on integers / out / max; rtol 1e-5 on the scale gradient (every
vote term has one sign); penalty values and gradients against the FLOAT64 oracle within 1e-5 * sum|terms|
(tests/_bounds.py: the condition-number form, the right yardstick where signed terms cancel)."""
import numpy as np
import pytest
import torch

from oracle import lq_oracle as O
from oracle import lq_oracle_f64 as O64

import os

from _bounds import assert_within_terms

pytestmark = pytest.mark.gpu
RTOL = 1e-5
# tools/soak_fuzz.py re-runs these sweeps with other seeds / more cases in one process
SEED_SHIFT = int(os.environ.get("LQ_FUZZ_SEED", "0"))
ITER_SCALE = int(os.environ.get("LQ_FUZZ_SCALE", "1"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand_shape(rng):
    rank = rng.choice([1, 2, 2, 4, 4, 3])
    if rank == 1:
        return (int(rng.integers(1, 3000)),)
    if rank == 2:
        return (int(rng.integers(1, 400)), int(rng.integers(1, 400)))
    if rank == 3:
        return (int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 1500)))
    return (int(rng.integers(1, 8)), int(rng.integers(1, 8)), int(rng.integers(1, 70)), int(rng.integers(1, 70)))


def _rand_case(rng):
    shape = _rand_shape(rng)
    orient = "scalar" if len(shape) == 1 else str(rng.choice(["rowwise", "columnwise", "channelwise", "scalar"]))
    if orient == "channelwise" and len(shape) < 3:
        orient = "rowwise"
    mag = float(10.0 ** rng.uniform(-3, 2))
    P = (rng.normal(0, mag, size=shape)).astype(np.float32)
    smag = float(10.0 ** rng.uniform(-6, 1)) * mag
    s = rng.uniform(0.5 * smag, 2 * smag, size=O.scale_shape(shape, orient)).astype(np.float32)
    dy = (rng.normal(0, 10.0 ** rng.uniform(-6, 0), size=shape)).astype(np.float32)
    lam = float(rng.choice([0.0, 1e-11, 1e-6, 1e-3, 3e-2, 0.4, 2.0]))
    kind = rng.integers(0, 6)
    flat = P.reshape(-1)
    if kind == 1 and flat.size > 4:
        flat[rng.integers(0, flat.size, size=max(1, flat.size // 10))] = 0.0          # exact zeros (out == 0 branch)
    elif kind == 2 and flat.size > 4:
        flat[rng.integers(0, flat.size, size=3)] = np.float32(1e-42)                    # denormals
        flat[rng.integers(0, flat.size)] = np.float32(-0.0)
    elif kind == 3:
        s = (s * np.float32(2.0 ** rng.integers(-30, 30))).astype(np.float32)          # far-off scales
    elif kind == 4 and flat.size > 4:
        s.reshape(-1)[0] = np.float32(np.nextafter(np.float32(2.0), np.float32(1.0)))   # all-ones mantissa -> IEEE path
    return P, s, dy, lam, orient


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_fuzz_forward_backward(dev):
    import learned_quantization_amd as lq
    rng = np.random.default_rng(20240229 + SEED_SHIFT)
    for it in range(160 * ITER_SCALE):
        P, s, dy, lam, orient = _rand_case(rng)
        tag = f"case {it}: shape={P.shape} orient={orient} lam={lam}"
        with np.errstate(all="ignore"):
            q_o, out_o = O.fq_forward(P, s)
            _, ds_o, im = O.nq_backward(P, s, lam, dy, return_intermediates=True)
        out, q = lq.fq_forward(_t(P, dev), _t(s, dev), q_dtype=torch.float32)
        np.testing.assert_array_equal(q.cpu().numpy(), q_o, err_msg=tag)
        np.testing.assert_array_equal(out.cpu().numpy(), out_o, err_msg=tag)
        ds, parts = lq.fq_scale_grad(_t(P, dev), _t(s, dev), _t(dy, dev), lam, return_parts=True)
        np.testing.assert_array_equal(parts[0].cpu().numpy(), np.asarray(im["maxvalue"], np.float32).reshape(-1), err_msg=tag)
        np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=RTOL, atol=1e-30, equal_nan=True, err_msg=tag)
        out2, ds2 = lq.fq_fwd_bwd_fused(_t(P, dev), _t(s, dev), _t(dy, dev), lam)
        np.testing.assert_array_equal(out2.cpu().numpy(), out_o, err_msg=tag)
        np.testing.assert_allclose(ds2.cpu().numpy(), ds.cpu().numpy(), rtol=2e-6, atol=1e-30, equal_nan=True, err_msg=tag)


def test_fuzz_misaligned_views(dev):
    """Storage offsets of 1..3 floats break 16-byte alignment: the scalar kernels must give identical results."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(7 + SEED_SHIFT)
    for it in range(24 * ITER_SCALE):
        rows, cols = int(rng.integers(1, 6)), int(rng.integers(1024, 6000))
        off = int(rng.integers(1, 4))
        base_p = torch.from_numpy(rng.normal(0, 0.05, size=rows * cols + off).astype(np.float32)).to(dev)
        base_d = torch.from_numpy(rng.normal(0, 1e-3, size=rows * cols + off).astype(np.float32)).to(dev)
        P, dy = base_p[off:].view(rows, cols), base_d[off:].view(rows, cols)
        assert P.data_ptr() % 16 != 0 and P.is_contiguous()
        s = torch.from_numpy(rng.uniform(1e-3, 1e-2, size=(rows, 1)).astype(np.float32)).to(dev)
        q_o, out_o = O.fq_forward(P.cpu().numpy(), s.cpu().numpy())
        _, ds_o = O.nq_backward(P.cpu().numpy(), s.cpu().numpy(), 1e-3, dy.cpu().numpy())
        np.testing.assert_array_equal(lq.fq_forward(P, s).cpu().numpy(), out_o)
        np.testing.assert_allclose(lq.fq_scale_grad(P, s, dy, 1e-3).cpu().numpy(), ds_o, rtol=RTOL)
        out2, ds2 = lq.fq_fwd_bwd_fused(P, s, dy, 1e-3)
        np.testing.assert_array_equal(out2.cpu().numpy(), out_o)


def test_periodic_columns_misaligned_and_penalties(dev):
    """C <= 64 at streaming size: the float4 grid-stride variant, its scalar fallback on the same geometry (bases off the
    16-byte grid), and the penalty passes, all against the oracle."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(11)
    for rows, C, inner in ((1400001, 3, 1), (350000, 12, 4), (66000, 64, 1), (350001, 3, 1)):   # the last: below 4 M, scalar form
        n = rows * C
        G = C // inner
        for off in (0, 1):
            base_p = torch.from_numpy(rng.normal(0, 0.05, size=n + off).astype(np.float32)).to(dev)
            base_d = torch.from_numpy(rng.normal(0, 1e-3, size=n + off).astype(np.float32)).to(dev)
            P, dy = base_p[off:].view(rows, G, inner), base_d[off:].view(rows, G, inner)
            assert (P.data_ptr() % 16 != 0) == bool(off)
            s = torch.from_numpy(rng.uniform(1e-3, 1e-2, size=(1, G, 1)).astype(np.float32)).to(dev)
            Pn, sn, dn = P.cpu().numpy(), s.cpu().numpy(), dy.cpu().numpy()
            q_o, out_o = O.fq_forward(Pn, sn)
            tag = f"rows={rows} C={C} inner={inner} off={off}"
            out, q = lq.fq_forward(P, s, q_dtype=torch.int32)
            np.testing.assert_array_equal(out.cpu().numpy(), out_o, err_msg=tag)
            np.testing.assert_array_equal(q.cpu().numpy(), q_o.astype(np.int32), err_msg=tag)
            for lam in (1e-10, 2e-2):
                _, ds_o = O.nq_backward(Pn, sn, lam, dn)
                ds = lq.fq_scale_grad(P, s, dy, lam)
                np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=RTOL, err_msg=tag)
                out2, ds2 = lq.fq_fwd_bwd_fused(P, s, dy, lam)
                np.testing.assert_array_equal(out2.cpu().numpy(), out_o, err_msg=tag)
                np.testing.assert_allclose(ds2.cpu().numpy(), ds.cpu().numpy(), rtol=2e-6, atol=1e-30, equal_nan=True, err_msg=tag)
        # penalty terms on the aligned tensor, against the float64 oracle within 1e-5 * sum|terms|
        _check_penalty_terms(lq, P, s, Pn, sn, dev, tag)


def _check_penalty_terms(lq, P_dev, s_dev, Pn, sn, dev, tag):
    """MaxBin / Difference / Inverse term of one tensor: value, dP and ds against oracle/lq_oracle_f64.py.
    custom_loss_functions.py:90-110 (MaxBin), :172-176 (Difference), :252-256 (Inverse)."""
    desc = O.group_descriptor(Pn.shape, sn.shape)
    Pt = P_dev.detach().clone().requires_grad_(True)
    st = s_dev.detach().clone().requires_grad_(True)
    mb = lq.maxbin_term(Pt, st)
    assert_within_terms(float(mb), O64.maxbin_term(Pn, sn, *desc), O64.term_abs("maxbin", Pn, sn, *desc), f"{tag}: maxbin value")
    df = lq.difference_term(Pt, st)
    assert_within_terms(float(df), O64.difference_term(Pn, sn, *desc), O64.term_abs("difference", Pn, sn, *desc), f"{tag}: difference value")
    iv = lq.inverse_term(st)
    assert_within_terms(float(iv), O64.inverse_term(sn), O64.term_abs("inverse", Pn, sn, *desc), f"{tag}: inverse value")
    (mb * 0.3).backward()
    dp64, ds64, ds_abs = O64.maxbin_term_grads(Pn, sn, 0.3, *desc)
    # dP of MaxBin: which elements tie for the maximum is decided by the float32 quotients (as in the reference), so the
    # element-wise gradient is compared with the float32 oracle; ds = -c/G * max/s does not depend on the tie split
    dp32, _ = O.maxbin_term_grads(Pn, sn, 0.3)
    np.testing.assert_allclose(Pt.grad.cpu().numpy(), dp32, rtol=1e-5, atol=0, err_msg=f"{tag}: maxbin dP")
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, f"{tag}: maxbin ds")
    Pt.grad = None
    st.grad = None
    (df * 0.7).backward()
    dp64, ds64, ds_abs, dp_abs = O64.difference_term_grads(Pn, sn, 0.7, *desc, with_dP_abs=True)
    assert_within_terms(Pt.grad.cpu().numpy(), dp64, dp_abs, f"{tag}: difference dP")     # g - g/s: two terms, cancel near s = 1
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, f"{tag}: difference ds")
    st.grad = None
    (iv * 1.3).backward()
    ds64, ds_abs = O64.inverse_term_grads(sn, 1.3)
    assert_within_terms(st.grad.cpu().numpy(), ds64, ds_abs, f"{tag}: inverse ds")


def test_fuzz_penalty_terms(dev):
    import learned_quantization_amd as lq
    rng = np.random.default_rng(99 + SEED_SHIFT)
    for it in range(60 * ITER_SCALE):
        P, s, _, _, orient = _rand_case(rng)
        s = np.abs(s) + np.float32(1e-12)
        with np.errstate(all="ignore"):
            _check_penalty_terms(lq, torch.tensor(P, device=dev), torch.tensor(s, device=dev), P, s, dev,
                                 f"case {it}: shape={P.shape} orient={orient}")


def test_inf_and_huge_values_take_the_ieee_path(dev):
    import learned_quantization_amd as lq
    P = np.array([[np.inf, -np.inf, 3e38, -3e38, 1e-45, -1e-45, 0.0, -0.0] * 160], np.float32)    # one row of 1280
    for sv in (0.5, 1e-3, 7.0, 3e-20, 3e20):
        s = np.array([[sv]], np.float32)
        with np.errstate(all="ignore"):
            q_o, out_o = O.fq_forward(P, s)
        out, q = lq.fq_forward(_t(P, dev), _t(s, dev), q_dtype=torch.float32)
        np.testing.assert_array_equal(q.cpu().numpy(), q_o)
        np.testing.assert_array_equal(out.cpu().numpy(), out_o)
        # sign of zero is preserved exactly (bit pattern), like IEEE division
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), out_o.view(np.uint32))


def test_pathological_scales(dev):
    """Zero, negative, denormal, infinite and NaN scales: outside the constraint of the reference (s >= 100*eps), but the op
    must still be the same three IEEE operations as the oracle -- including where NaN / Inf appear."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(5)
    P = rng.normal(0, 1.0, size=(6, 2048)).astype(np.float32)
    P[0, :8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3e38]
    dy = rng.normal(0, 1e-3, size=P.shape).astype(np.float32)
    s = np.array([[0.0], [-0.25], [1e-41], [np.inf], [np.nan], [3e38]], np.float32)
    with np.errstate(all="ignore"):
        q_o, out_o = O.fq_forward(P, s)
        _, ds_o, im = O.nq_backward(P, s, 1e-3, dy, return_intermediates=True)
    out, q = lq.fq_forward(_t(P, dev), _t(s, dev), q_dtype=torch.float32)
    np.testing.assert_array_equal(q.cpu().numpy(), q_o)                      # NaN == NaN positionally in assert_array_equal
    np.testing.assert_array_equal(out.cpu().numpy(), out_o)
    ds = lq.fq_scale_grad(_t(P, dev), _t(s, dev), _t(dy, dev), 1e-3).cpu().numpy()
    assert np.array_equal(np.isnan(ds), np.isnan(ds_o))
    fin = ~np.isnan(ds_o)
    np.testing.assert_allclose(ds[fin], ds_o[fin], rtol=RTOL)
    # the same rows through the small-row and column traversals
    P2 = np.ascontiguousarray(P[:, :48])
    dy2 = np.ascontiguousarray(dy[:, :48])
    with np.errstate(all="ignore"):
        _, out2_o = O.fq_forward(P2, s)
        _, out3_o = O.fq_forward(np.ascontiguousarray(P2.T), s.reshape(1, 6))
    np.testing.assert_array_equal(lq.fq_forward(_t(P2, dev), _t(s, dev)).cpu().numpy(), out2_o)
    np.testing.assert_array_equal(lq.fq_forward(_t(np.ascontiguousarray(P2.T), dev), _t(s.reshape(1, 6), dev)).cpu().numpy(), out3_o)


def test_negative_and_nan_threshold(dev):
    """penalty_threshold outside its meaningful range still follows the reference's comparisons: lambda < 0 -> every ratio
    is 'above' -> mean = -|tanh(lambda)|; lambda = NaN -> every comparison is false -> NaN."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(8)
    P = rng.normal(0, 0.05, size=(5, 3000)).astype(np.float32)
    dy = rng.normal(0, 1e-3, size=P.shape).astype(np.float32)
    s = rng.uniform(1e-3, 1e-2, size=(5, 1)).astype(np.float32)
    for lam in (-1e-3, -0.5):
        with np.errstate(all="ignore"):
            _, ds_o = O.nq_backward(P, s, lam, dy)
        ds = lq.fq_scale_grad(_t(P, dev), _t(s, dev), _t(dy, dev), lam).cpu().numpy()
        np.testing.assert_allclose(ds, ds_o, rtol=RTOL)
    with np.errstate(all="ignore"):
        _, ds_o = O.nq_backward(P, s, float("nan"), dy)
    ds = lq.fq_scale_grad(_t(P, dev), _t(s, dev), _t(dy, dev), float("nan")).cpu().numpy()
    assert np.all(np.isnan(ds_o)) and np.all(np.isnan(ds))


def test_int32_view_saturates(dev):
    import learned_quantization_amd as lq
    P = np.array([1e30, -1e30, 2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0, np.nan, np.inf, -np.inf, 5.9], np.float32)
    s = np.array([1.0], np.float32)
    q = lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int32).cpu().numpy()
    i32 = np.iinfo(np.int32)
    np.testing.assert_array_equal(q, np.array([i32.max, i32.min, 2147483520, i32.max, i32.min, i32.min, 0, i32.max, i32.min, 5], np.int64).astype(np.int32))


def _rand_streaming_case(rng):
    """Random descriptor of 4.2-6 M elements that lands in one of the streaming-size forms (csrc/lq_stream2.hpp): column tile
    (any C % 4 == 0 > 64, ragged row counts), periodic columns (C <= 64, any C), tiny rows (L in 8..64, L % 4 == 0), rows of
    68..1020 (flat forward), ragged long rows (TAIL instantiation of the row stream), rows of 5..1023 with L % 4 != 0, short rows in 32+ layers."""
    kind = int(rng.integers(0, 7))
    n = int(rng.integers(4_200_000, 6_000_000))
    if kind == 6:                                   # short rows in many layers: column mode with 16 <= inner < 200
        inner = int(rng.integers(17, 200))
        outer = int(rng.integers(32, 80))
        G = max(32, (n // (outer * inner)) // 32 * 32)        # G * inner: whole 128-byte lines
        return (outer, G, inner), "columnwise"
    if kind == 5:                                   # rows of 5..1023 elements off the 16-byte grid: straddling flat forward + row windows
        L = int(rng.integers(5, 65)) if rng.integers(0, 2) else int(rng.integers(5, 1024))      # half of them short rows
        L += 1 if L % 4 == 0 else 0
        if rng.integers(0, 2) or L < 16:
            return (n // L, L), "rowwise"
        outer = int(rng.integers(2, 6))
        return (outer, n // (L * outer), L), "columnwise"
    if kind == 0:                                   # column tile
        inner = int(rng.choice([1, 1, 2, 4, 6, 8, 12]))
        G = int(rng.integers(17, 700)) * 4 // (4 if inner % 4 == 0 else 1)
        G = max(G, 68 // inner + 1)
        while (G * inner) % 4 != 0 and rng.integers(0, 2):     # half of the cases keep C % 4 != 0 (dword-aligned float4 tile)
            G += 1
        outer = max(n // (G * inner), 2)
        return (outer, G, inner), "columnwise"
    if kind == 1:                                   # periodic columns
        C = int(rng.integers(2, 65))
        inner = int(rng.choice([d for d in (1, 2, 3, 4, 8) if C % d == 0 and C // d >= 2] or [1]))
        return (max(n // C, 2), C // inner, inner), "columnwise"
    if kind == 2:                                   # tiny rows
        L = int(rng.choice([8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64]))
        if rng.integers(0, 2):
            return (n // L, L), "rowwise"
        outer = int(rng.integers(2, 6))
        return (outer, n // (L * outer), L), "columnwise"
    if kind == 3:                                   # rows of 68..1020 elements
        L = int(rng.integers(17, 256)) * 4
        return (n // L, L), "rowwise"
    L = int(rng.integers(1025, 9000))               # long ragged rows
    return (n // L, L), "rowwise"


def test_fuzz_streaming_size_forms(dev):
    """Seeded random descriptors at streaming size through K1, K2+K3 and K4 against the oracle (bit-exact q / out / max|q|,
    ds within 1e-5): the round-2 kernels with random ragged extents, scale magnitudes and thresholds."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(424242 + SEED_SHIFT)
    for it in range(10 * ITER_SCALE):
        shape, orient = _rand_streaming_case(rng)
        mag = float(10.0 ** rng.uniform(-2, 1))
        P = rng.normal(0, mag, size=shape).astype(np.float32)
        s = (rng.uniform(0.5, 2.0, size=O.scale_shape(shape, orient)) * mag * float(10.0 ** rng.uniform(-4, 0))).astype(np.float32)
        dy = (rng.normal(0, 1, size=shape) * 10.0 ** rng.uniform(-9, -1, size=shape)).astype(np.float32)
        lam = float(rng.choice([0.0, 1e-11, 1e-6, 1e-3, 3e-2, 0.4]))
        tag = f"case {it}: shape={shape} orient={orient} lam={lam}"
        with np.errstate(all="ignore"):
            q_o, out_o = O.fq_forward(P, s)
            _, ds_o, im = O.nq_backward(P, s, lam, dy, return_intermediates=True)
        Pt, st, dt = _t(P, dev), _t(s, dev), _t(dy, dev)
        out, q = lq.fq_forward(Pt, st, q_dtype=torch.float32)
        np.testing.assert_array_equal(q.cpu().numpy(), q_o, err_msg=tag)
        np.testing.assert_array_equal(out.cpu().numpy(), out_o, err_msg=tag)
        ds, parts = lq.fq_scale_grad(Pt, st, dt, lam, return_parts=True)
        np.testing.assert_array_equal(parts[0].cpu().numpy(), np.asarray(im["maxvalue"], np.float32).reshape(-1), err_msg=tag)
        np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=RTOL, atol=1e-30, equal_nan=True, err_msg=tag)
        out2, ds2 = lq.fq_fwd_bwd_fused(Pt, st, dt, lam)
        np.testing.assert_array_equal(out2.cpu().numpy(), out_o, err_msg=tag)
        np.testing.assert_allclose(ds2.cpu().numpy(), ds_o, rtol=RTOL, atol=1e-30, equal_nan=True, err_msg=tag)
