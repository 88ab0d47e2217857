"""CPU tests that PIN THE ORACLE (no GPU, no product code under test).

The reference has no tests and cannot run here, so the oracle is pinned by:
  1. the hand-derived KAT of SURVEY.md 8(c)            (tests/golden/kat_survey.json)
  2. an independent float64 restatement of the thesis   (oracle/lq_oracle_f64.py)
  3. an independent scalar C restatement                (oracle/lq_oracle.c)
  4. the op-for-op torch-CPU restatement (bench baseline) agreeing with the NumPy one
  5. the source-implied invariants of SURVEY.md section 4
  6. regression against the committed golden fixtures.
"""
import hashlib

import numpy as np

from _bounds import stable_seed
import pytest
import torch

from oracle import lq_oracle as O
from oracle import lq_oracle_f64 as O64
from oracle import lq_oracle_torch as OT
from conftest import as_f32p

RTOL = 1e-5   # tolerance north_star states for float quantities


# ------------------------------------------------------------------ 1. hand-derived KAT
def test_kat_forward_and_ratio(kat):
    P, s, dy = (np.array(kat[k], np.float32) for k in ("P", "s", "dy"))
    q, out = O.fq_forward(P, s)
    np.testing.assert_array_equal(q, np.array(kat["q"], np.float32))
    np.testing.assert_array_equal(out, np.array(kat["out"], np.float32))
    _, _, im = O.nq_backward(P, s, 0.5, dy, return_intermediates=True)
    np.testing.assert_allclose(im["ratio"], np.array(kat["ratio"], np.float32), rtol=1e-6)


@pytest.mark.parametrize("lam", ["0.5", "0.05"])
def test_kat_scale_gradient(kat, lam):
    P, s, dy = (np.array(kat[k], np.float32) for k in ("P", "s", "dy"))
    dP, ds, im = O.nq_backward(P, s, float(lam), dy, return_intermediates=True)
    assert dP is not None and np.array_equal(dP, dy)
    exp = kat[f"lambda_{lam}"]
    np.testing.assert_allclose(ds, np.array(exp["ds"], np.float32), rtol=1e-6)
    if "sg" in exp:
        np.testing.assert_allclose(im["sg"], np.array(exp["sg"], np.float32), rtol=1e-6, atol=1e-9)


def test_kat_penalties(kat):
    P, s = np.array(kat["P"], np.float32), np.array(kat["s"], np.float32)
    b, sb = np.array(kat["bias"]["b"], np.float32), np.array(kat["bias"]["s_b"], np.float32)
    layers = [(P, s, b, sb)]
    assert float(O.maxbin_penalty(layers)) == pytest.approx(kat["penalties"]["maxbin"], rel=1e-6)
    assert float(O.difference_penalty(layers)) == pytest.approx(kat["penalties"]["difference"], rel=1e-6)
    assert float(O.inverse_penalty(layers)) == pytest.approx(kat["penalties"]["inverse"], rel=1e-6)


# ------------------------------------------------------------------ scale shapes (custom_layers.py:147-197)
def test_scale_shapes_match_reference_tables():
    # thesis chapter4.tex:87-103, 204-223 / SURVEY 8a row a8
    assert O.scale_shape((784, 128), "rowwise") == (784, 1)
    assert O.scale_shape((784, 128), "columnwise") == (1, 128)
    assert O.scale_shape((784, 128), "channelwise") == (1, 1)
    assert O.scale_shape((784, 128), "scalar") == (1,)
    assert O.scale_shape((3, 3, 64, 128), "rowwise") == (3, 1, 1, 1)
    assert O.scale_shape((3, 3, 64, 128), "columnwise") == (1, 3, 1, 1)
    assert O.scale_shape((3, 3, 64, 128), "channelwise") == (1, 1, 64, 1)
    assert O.scale_shape((128,), "scalar") == (1,)
    with pytest.raises(ValueError, match="Invalid scaler application"):
        O.scale_shape((3, 3), "diagonal")


def test_group_descriptor_table():
    # SURVEY 8b orientation -> (outer, G, inner), TF layouts
    assert O.group_descriptor((784, 128), (784, 1)) == (1, 784, 128)
    assert O.group_descriptor((784, 128), (1, 128)) == (784, 128, 1)
    assert O.group_descriptor((784, 128), (1, 1)) == (1, 1, 784 * 128)
    assert O.group_descriptor((784, 128), (1,)) == (1, 1, 784 * 128)
    assert O.group_descriptor((3, 3, 64, 128), (3, 1, 1, 1)) == (1, 3, 3 * 64 * 128)
    assert O.group_descriptor((3, 3, 64, 128), (1, 3, 1, 1)) == (3, 3, 64 * 128)
    assert O.group_descriptor((3, 3, 64, 128), (1, 1, 64, 1)) == (9, 64, 128)
    assert O.group_descriptor((256, 3, 224, 224), (1, 3, 1, 1)) == (256, 3, 50176)


# ------------------------------------------------------------------ 2./3. independent restatements
def _random_case(rng, shape, orient, lam):
    P = rng.normal(0, 0.05, size=shape).astype(np.float32)
    dy = rng.normal(0, 1e-3, size=shape).astype(np.float32)
    s = rng.uniform(1e-3, 3e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
    return P, s, dy, lam


CASES = [((37, 20), o) for o in O.ORIENTATIONS] + [((3, 3, 5, 7), o) for o in O.ORIENTATIONS] + [((11,), "scalar")]


@pytest.mark.parametrize("shape,orient", CASES)
@pytest.mark.parametrize("lam", [0.0, 1e-10, 5e-2, 0.7])
def test_oracle_vs_f64_and_c(shape, orient, lam, c_oracle):
    rng = np.random.default_rng(stable_seed(shape, orient, lam))
    P, s, dy, lam = _random_case(rng, shape, orient, lam)
    desc = O.group_descriptor(P.shape, s.shape)
    q, out = O.fq_forward(P, s)
    _, ds, im = O.nq_backward(P, s, lam, dy, return_intermediates=True)

    # float64 thesis math: integers agree except where the fp32 quotient sits within rounding of an integer
    q64, _ = O64.forward(P, s, *desc)
    t64 = P.astype(np.float64).reshape(-1) / np.broadcast_to(s, P.shape).astype(np.float64).reshape(-1)
    safe = np.abs(t64 - np.round(t64)) > 1e-4 * np.maximum(1.0, np.abs(t64))
    assert safe.mean() > 0.95
    np.testing.assert_array_equal(q.reshape(-1)[safe], q64[safe])
    ds64 = O64.scale_grad(P, s, lam, dy, *desc)
    if safe.all():
        np.testing.assert_allclose(ds.reshape(-1), ds64, rtol=2e-5, atol=1e-12)

    # scalar C restatement: integers / out / max bit-exact, ds within tolerance
    n, G = P.size, desc[1]
    Pc, sc, dyc = (np.ascontiguousarray(a.reshape(-1)) for a in (P, s, dy))
    out_c, q_c = np.empty(n, np.float32), np.empty(n, np.float32)
    c_oracle.lqo_fq_forward(as_f32p(Pc), as_f32p(sc), as_f32p(out_c), as_f32p(q_c), *desc)
    np.testing.assert_array_equal(q_c, q.reshape(-1))
    np.testing.assert_array_equal(out_c, out.reshape(-1))
    ds_c, maxq_c, mean_c = (np.empty(G, np.float32) for _ in range(3))
    below = np.empty(G, np.int64)
    import ctypes
    c_oracle.lqo_nq_scale_grad(as_f32p(Pc), as_f32p(sc), as_f32p(dyc), np.float32(lam), as_f32p(ds_c), as_f32p(maxq_c),
                               as_f32p(mean_c), below.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), *desc)
    np.testing.assert_array_equal(maxq_c, np.asarray(im["maxvalue"], np.float32).reshape(-1))
    np.testing.assert_allclose(ds_c, ds.reshape(-1), rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("shape,orient", CASES)
def test_oracle_vs_torch_restatement(shape, orient):
    rng = np.random.default_rng(7)
    for lam in (0.0, 1e-10, 5e-2):
        P, s, dy, lam = _random_case(rng, shape, orient, lam)
        q, out = O.fq_forward(P, s)
        _, ds = O.nq_backward(P, s, lam, dy)
        qt, outt = OT.fq_forward(torch.from_numpy(P), torch.from_numpy(s))
        np.testing.assert_array_equal(qt.numpy(), q)
        np.testing.assert_array_equal(outt.numpy(), out)
        o2, dp, dst = OT.nq_forward_backward(torch.from_numpy(P), torch.from_numpy(s), lam, torch.from_numpy(dy))
        np.testing.assert_array_equal(dp.numpy(), dy)
        np.testing.assert_allclose(dst.numpy(), ds, rtol=RTOL, atol=1e-12)


def test_penalties_three_way(c_oracle):
    rng = np.random.default_rng(3)
    layers, layers64 = [], []
    for shape, orient in [((3, 3, 5, 7), "channelwise"), ((3, 3, 7, 4), "rowwise"), ((20, 9), "columnwise"), ((9, 4), "scalar")]:
        K = rng.normal(0, 0.05, size=shape).astype(np.float32)
        sK = rng.uniform(1e-3, 3e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
        b = rng.normal(0, 0.05, size=(shape[-1],)).astype(np.float32)
        sb = rng.uniform(1e-3, 3e-2, size=(1,)).astype(np.float32)
        layers.append((K, sK, b, sb))
        layers64.append((K, sK, O.group_descriptor(K.shape, sK.shape), b, sb, O.group_descriptor(b.shape, sb.shape)))
    tl = [tuple(torch.from_numpy(a) for a in l) for l in layers]
    for kind, fn, fnt in (("maxbin", O.maxbin_penalty, OT.maxbin_penalty),
                          ("difference", O.difference_penalty, OT.difference_penalty),
                          ("inverse", O.inverse_penalty, OT.inverse_penalty)):
        v = float(fn(layers))
        assert v == pytest.approx(O64.penalty(kind, layers64), rel=RTOL)
        assert v == pytest.approx(float(fnt(tl)), rel=RTOL)
    # per-tensor terms against C
    K, sK, b, sb = layers[0]
    d = O.group_descriptor(K.shape, sK.shape)
    Kc, sc = np.ascontiguousarray(K.reshape(-1)), np.ascontiguousarray(sK.reshape(-1))
    assert c_oracle.lqo_maxbin_term(as_f32p(Kc), as_f32p(sc), *d) == pytest.approx(O64.maxbin_term(K, sK, *d), rel=RTOL)
    assert c_oracle.lqo_difference_term(as_f32p(Kc), as_f32p(sc), *d) == pytest.approx(O64.difference_term(K, sK, *d), rel=RTOL)
    assert c_oracle.lqo_inverse_term(as_f32p(sc), sc.size) == pytest.approx(O64.inverse_term(sK), rel=RTOL)


def test_penalty_gradients_match_torch_autograd():
    """The analytic fp32 gradients (what the HIP backward kernels implement) against autograd of the
    op-for-op restatement -- torch.amax splits ties evenly like tf.reduce_max."""
    rng = np.random.default_rng(11)
    for shape, orient in [((3, 3, 5, 7), "channelwise"), ((20, 9), "rowwise"), ((20, 9), "columnwise"), ((13,), "scalar"), ((6, 5), "channelwise")]:
        P = rng.normal(0, 0.05, size=shape).astype(np.float32)
        P.reshape(-1)[3] = P.reshape(-1)[0]           # force a tie candidate
        s = rng.uniform(1e-3, 3e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
        c = 0.37
        for name, fwd, grads in (("maxbin", lambda p, q: torch.mean(OT._maxbin(p, q)), O.maxbin_term_grads),
                                 ("difference", lambda p, q: torch.mean(torch.abs(p - p / q)), O.difference_term_grads)):
            pt = torch.from_numpy(P.copy()).requires_grad_(True)
            st = torch.from_numpy(s.copy()).requires_grad_(True)
            (fwd(pt, st) * c).backward()
            dp, ds = grads(P, s, c)
            np.testing.assert_allclose(dp, pt.grad.numpy(), rtol=1e-5, atol=1e-9, err_msg=f"{name} dP {shape} {orient}")
            np.testing.assert_allclose(ds, st.grad.numpy(), rtol=1e-4, atol=1e-7, err_msg=f"{name} ds {shape} {orient}")
        st = torch.from_numpy(s.copy()).requires_grad_(True)
        (torch.mean(1.0 / st) * c).backward()
        np.testing.assert_allclose(O.inverse_term_grads(s, c), st.grad.numpy(), rtol=1e-5)


# ------------------------------------------------------------------ 5. invariants
def test_invariants():
    rng = np.random.default_rng(5)
    P, s, dy, _ = _random_case(rng, (3, 3, 6, 8), "channelwise", 0)
    q, out = O.fq_forward(P, s)
    assert np.array_equal(q, np.round(q))                                  # q integral
    assert np.array_equal(out, q * s)                                       # out == q*s exactly
    dP, ds0 = O.nq_backward(P, s, 0.0, dy)
    assert np.array_equal(dP, dy)                                           # dP == dy bit for bit
    assert np.all(ds0 == 0)                                                 # lambda = 0 -> (-)0.0; the sign of zero depends on the summation
    for lam in (1e-10, 1e-3, 0.3):
        _, ds = O.nq_backward(P, s, lam, dy)
        assert np.all(ds <= 0)                                              # ds <= 0 always
    dP, dsz = O.ste_backward(P, s, dy)
    assert np.array_equal(dP, dy) and np.all(dsz == 0) and dsz.shape == s.shape
    assert np.all(O.min_value_constraint(np.array([0.0, 1e-6, 1.0], np.float32)) >= O.SCALE_MIN)
    # permutation inside a group leaves max / all exactly and the mean within tolerance
    perm = rng.permutation(3 * 3 * 8)
    Pp = P.transpose(2, 0, 1, 3).reshape(6, -1)[:, perm].reshape(6, 3, 3, 8).transpose(1, 2, 0, 3)
    dyp = dy.transpose(2, 0, 1, 3).reshape(6, -1)[:, perm].reshape(6, 3, 3, 8).transpose(1, 2, 0, 3)
    _, ds_a = O.nq_backward(P, s, 1e-3, dy)
    _, ds_b = O.nq_backward(np.ascontiguousarray(Pp), s, 1e-3, np.ascontiguousarray(dyp))
    np.testing.assert_allclose(ds_a, ds_b, rtol=1e-6)


def test_correctly_rounded_quotient_matters():
    """SURVEY section 7 hard part 1: floor(P * (1/s)) and floor(exact quotient) both differ from
    floor(RN(P/s)) -- the oracle must be (and is) the fp32-RN one."""
    rng = np.random.default_rng(0)
    P = rng.normal(0, 0.05, size=1 << 20).astype(np.float32)
    s = np.float32(O.SCALE_INIT)
    q = O.quantized_integers(P, np.array([s]))
    q_recip = np.floor(P * (np.float32(1.0) / s))
    q_exact = np.floor(P.astype(np.float64) / np.float64(s))
    assert np.count_nonzero(q != q_recip) > 0
    assert np.array_equal(q, np.floor((P / s).astype(np.float32)))
    assert np.count_nonzero(q != q_exact) >= 0   # informational; differs at 1e-4-like scales


def test_export_int8_wraps():
    P = np.array([0.0, 1.0, 127.0, 128.0, 255.0, 256.0, -1.0, -128.0, -129.0, 300.7], np.float32)
    got = O.export_int8(P, np.array([1.0], np.float32))
    np.testing.assert_array_equal(got, np.array([0, 1, 127, -128, -1, 0, -1, -128, 127, 44], np.int8))


def test_keras_adam_step_known_value():
    # first Adam step moves by ~lr regardless of gradient scale; projection keeps s >= 100*eps
    s = np.full(4, O.SCALE_INIT, np.float32)
    g = np.array([-1.0, -1e-3, 1.0, 0.0], np.float32)
    s1, m, v = O.keras_adam_step(s, g, np.zeros(4, np.float32), np.zeros(4, np.float32), 1, lr=1e-4,
                                 min_value=O.SCALE_MIN)
    np.testing.assert_allclose(s1[0], O.SCALE_INIT + 1e-4, rtol=1e-4)
    assert s1[2] == O.SCALE_MIN and s1[3] == O.SCALE_INIT
    assert np.all(s1 >= O.SCALE_MIN)


# ------------------------------------------------------------------ 6. golden regression
def test_golden_cases_regression(golden_cases):
    assert len(golden_cases) >= 30
    for name, c in golden_cases.items():
        q, out = O.fq_forward(c["P"], c["s"])
        np.testing.assert_array_equal(q, c["q"], err_msg=name)
        np.testing.assert_array_equal(out, c["out"], err_msg=name)
        _, ds, im = O.nq_backward(c["P"], c["s"], float(c["lam"]), c["dy"], return_intermediates=True)
        np.testing.assert_array_equal(np.asarray(im["maxvalue"], np.float32).reshape(-1), c["maxq"].reshape(-1), err_msg=name)
        np.testing.assert_allclose(ds, c["ds"], rtol=1e-6, atol=0, equal_nan=True, err_msg=name)


def test_mnist_real_weights_integers(mnist_weights, mnist_expected):
    meta, exp = mnist_expected
    for name in ("W1", "b1", "W2", "b2"):
        P = mnist_weights[name]
        q = O.quantized_integers(P, np.array([O.SCALE_INIT], np.float32)).astype(np.int32)
        np.testing.assert_array_equal(q, exp[f"{name}_q_init"].astype(np.int32))
        for key, m in meta.items():
            if not key.startswith(name + "@") or "absmax" in key:
                continue
            sval = np.float32(float(key.split("@")[1]))
            q = O.quantized_integers(P, np.array([sval], np.float32)).astype(np.int32)
            assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == m["sha256"], key
            assert int(q.min()) == m["min"] and int(q.max()) == m["max"]
    # real weights at the init scale: |q| up to ~1.9e4 -- NOT int8-safe
    assert np.abs(exp["W1_q_init"].astype(np.int32)).max() > 127


def test_c_oracle_under_asan_ubsan(tmp_path):
    """The C restatement on ragged shapes under AddressSanitizer + UBSan (CPU build; GPU sanitizers are unavailable)."""
    import os
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           "-o", exe, os.path.join(root, "tests", "tools", "oracle_sanitize.c"),
                           os.path.join(root, "oracle", "lq_oracle.c"), "-lm"])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert res.returncode == 0 and "oracle_sanitize: ok" in res.stdout, res.stdout + res.stderr
