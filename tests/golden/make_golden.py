"""Generates tests/golden/golden_cases.npz and tests/golden/mnist_expected.npz.

    python tests/golden/make_golden.py

The reference cannot run here (TensorFlow 2.11 not installed, no network) and ships no
golden vectors, so these fixtures are produced by the float32 restatement oracle/lq_oracle.py,
after tests/test_oracle.py has pinned that restatement against (a) the hand-derived KAT
kat_survey.json, (b) the float64 thesis-math restatement and (c) the scalar C restatement.
They are REGRESSION pins of the oracle and transport vectors to the GPU box (where
/root/reference does not exist); they are not reference outputs.

Cases: every (rank, orientation) pair incl. the degenerate ones of SURVEY 8a (channelwise on
rank 2 -> all-ones scale), edge cases (zeros, out == 0, lambda = 0, all-above, mostly-below,
NaN dy, negative zero), and the real MNIST baseline weights at several scales.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import lq_oracle as O  # noqa: E402


def case(rng, shape, orientation, lam, s_lo, s_hi, p_std=0.05, dy_std=1e-3, tweak=None):
    P = rng.normal(0.0, p_std, size=shape).astype(np.float32)
    dy = rng.normal(0.0, dy_std, size=shape).astype(np.float32)
    sshape = O.scale_shape(shape, orientation)
    s = rng.uniform(s_lo, s_hi, size=sshape).astype(np.float32)
    if tweak:
        P, dy, s = tweak(P, dy, s)
    q, out = O.fq_forward(P, s)
    _, ds, im = O.nq_backward(P, s, lam, dy, return_intermediates=True)
    return dict(P=P, s=s, dy=dy, lam=np.float32(lam), q=q, out=out, ds=ds,
                maxq=im["maxvalue"], mean=im["mean"])


def main():
    rng = np.random.default_rng(42)
    cases = {}
    shapes = {"dense": (37, 20), "conv": (3, 3, 5, 7), "bias": (11,)}
    i = 0
    for name, shape in shapes.items():
        for orient in O.ORIENTATIONS:
            if len(shape) == 1 and orient != "scalar":
                continue   # bias scale is always scalar (custom_layers.py:225-227)
            for lam in (0.0, 1e-10, 5e-2):
                cases[f"c{i:02d}_{name}_{orient}_lam{lam:g}"] = case(rng, shape, orient, lam, 1e-3, 3e-2)
                i += 1
    # init-scale case: s = 100*eps everywhere, |q| ~ 2e4
    cases["init_scale_conv_channelwise"] = case(
        rng, (3, 3, 8, 16), "channelwise", 1e-11, 1, 1,
        tweak=lambda P, dy, s: (P, dy, np.full_like(s, O.SCALE_INIT)))

    def zeros(P, dy, s):
        P[...] = 0.0
        return P, dy, s
    cases["edge_all_zero_param"] = case(rng, (6, 9), "rowwise", 1e-3, 1e-2, 2e-2, tweak=zeros)

    def out_zero(P, dy, s):
        P = np.abs(P) * 1e-3       # 0 <= P < s  -> q = 0, out = 0 -> eps substitution (:63)
        return P.astype(np.float32), dy, s
    cases["edge_out_zero"] = case(rng, (6, 9), "columnwise", 1e-3, 1e-2, 2e-2, tweak=out_zero)

    def neg_zero(P, dy, s):
        P[0, :] = -0.0
        P[1, :] = -1e-9            # q = -1 -> out = -s (not zero)
        return P, dy, s
    cases["edge_negative_zero"] = case(rng, (4, 8), "rowwise", 1e-2, 1e-2, 2e-2, tweak=neg_zero)

    def all_above(P, dy, s):
        return P, (np.sign(dy) * (np.abs(dy) + 10.0)).astype(np.float32), s
    cases["edge_all_above"] = case(rng, (5, 12), "rowwise", 1e-3, 1e-2, 2e-2, tweak=all_above)

    def mostly_below(P, dy, s):
        return P, (dy * 1e-6).astype(np.float32), s
    cases["edge_mostly_below_big_lambda"] = case(rng, (5, 12), "columnwise", 0.7, 1e-2, 2e-2, tweak=mostly_below)

    def nan_dy(P, dy, s):
        dy[1, 3] = np.nan
        return P, dy, s
    cases["edge_nan_dy"] = case(rng, (4, 8), "rowwise", 1e-3, 1e-2, 2e-2, tweak=nan_dy)

    # activation-like: U[0,255) data, per-channel scales 0.5/1/2, tiny spatial extent
    def act(P, dy, s):
        P = rng.uniform(0, 255, size=P.shape).astype(np.float32)
        s = np.array([0.5, 1.0, 2.0], np.float32).reshape(1, 3, 1, 1)
        return P, dy, s
    c = case(rng, (4, 3, 8, 8), "columnwise", 1e-3, 1, 1, tweak=act)
    cases["bench_like_nchw_per_channel"] = c

    flat = {}
    for k, v in cases.items():
        for kk, vv in v.items():
            flat[f"{k}/{kk}"] = vv
    np.savez_compressed(os.path.join(HERE, "golden_cases.npz"), **flat)
    print("golden_cases.npz:", len(cases), "cases")

    # ---- real weights -------------------------------------------------------------------
    w = np.load(os.path.join(HERE, "mnist_baseline_weights.npz"))
    exp = {}
    meta = {}
    for name in ("W1", "b1", "W2", "b2"):
        P = w[name]
        for sval in (float(O.SCALE_INIT), 1e-3, 0.0123):
            q = O.quantized_integers(P, np.array([sval], np.float32)).astype(np.int32)
            key = f"{name}@{sval:.9g}"
            meta[key] = dict(sum=int(q.astype(np.int64).sum()), abssum=int(np.abs(q.astype(np.int64)).sum()),
                             min=int(q.min()), max=int(q.max()),
                             sha256=hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest())
            if sval == float(O.SCALE_INIT):
                exp[f"{name}_q_init"] = q.astype(np.int16) if np.abs(q).max() < 32768 else q
        if P.ndim == 2:
            for orient in ("rowwise", "columnwise"):
                sshape = O.scale_shape(P.shape, orient)
                s = (np.abs(P).max(axis=1 if orient == "rowwise" else 0).reshape(sshape) / 12.0 + 1e-6).astype(np.float32)
                q = O.quantized_integers(P, s).astype(np.int32)
                meta[f"{name}@{orient}_absmax12"] = dict(
                    sum=int(q.astype(np.int64).sum()), abssum=int(np.abs(q.astype(np.int64)).sum()),
                    min=int(q.min()), max=int(q.max()),
                    sha256=hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest())
    np.savez_compressed(os.path.join(HERE, "mnist_expected.npz"), **exp)
    with open(os.path.join(HERE, "mnist_expected.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("mnist_expected:", len(meta), "entries")


if __name__ == "__main__":
    main()
