"""Float64 restatement of the THESIS math  --  TEST INFRASTRUCTURE ONLY.

Independent cross-check of oracle/lq_oracle.py.  Written from the formulas of
/root/reference/thesis/chapters/chapter3.tex, not from the Python source:

  forward              P_q = floor(P / s),  P_r = P_q * s                 chapter3.tex:30-37
  ratio                r = |grad| / |P_r|   with P_r := eps where P_r == 0 chapter3.tex:104-118
                       (the thesis prints max(eps,|P_r|); the code replaces only exact
                        zeros -- custom_layers.py:63 -- and SURVEY 8(c) says code wins)
  vote                 g = -tanh(lambda)            if all r >= lambda in the group
                           -tanh(lambda - r)*1(r<lambda)  otherwise           chapter3.tex:160-167
  range                m = max |P_q| over the group                          chapter3.tex:130-138
  scale gradient       grad_s = mean_group(g) * m                            chapter3.tex:84-88,169-171
  MaxBin / Inverse / Difference penalties                                    chapter3.tex:242-337

Structure differs on purpose from lq_oracle.py: everything is expressed through
a flat group index  gid(i) = (i // inner) % G  (the C-ABI descriptor) instead
of axis reductions, so a bug in either the axis logic or the descriptor logic
shows up as a disagreement between the two.
"""
from __future__ import annotations

import numpy as np

EPS_F32 = float(np.finfo(np.float32).eps)


def group_ids(outer, G, inner):
    i = np.arange(outer * G * inner, dtype=np.int64)
    return (i // inner) % G


def forward(P, s, outer, G, inner):
    P = np.asarray(P, np.float64).reshape(-1)
    s = np.asarray(s, np.float64).reshape(-1)
    gid = group_ids(outer, G, inner)
    q = np.floor(P / s[gid])
    return q, q * s[gid]


def scale_grad(P, s, lam, dy, outer, G, inner):
    """Returns ds[G] in float64."""
    P = np.asarray(P, np.float64).reshape(-1)
    s = np.asarray(s, np.float64).reshape(-1)
    dy = np.asarray(dy, np.float64).reshape(-1)
    lam = float(np.float32(lam))
    gid = group_ids(outer, G, inner)
    q, pr = forward(P, s, outer, G, inner)
    pr_nz = np.where(pr == 0.0, EPS_F32, pr)
    r = np.abs(dy) / np.abs(pr_nz)
    ds = np.zeros(G, np.float64)
    for g in range(G):
        sel = gid == g
        rg = r[sel]
        m = np.max(np.abs(q[sel]))
        if np.all(rg >= lam):
            vote = -abs(np.tanh(lam))
        else:
            below = ~(rg >= lam)
            vote = np.sum(-np.abs(np.tanh(lam - rg[below]))) / rg.size
        ds[g] = vote * m
    return ds


def maxbin_term(P, s, outer, G, inner):
    """mean over groups of max |P|/s_k."""
    P = np.asarray(P, np.float64).reshape(-1)
    s = np.asarray(s, np.float64).reshape(-1)
    gid = group_ids(outer, G, inner)
    t = np.abs(P) / s[gid]
    return float(np.mean([np.max(t[gid == g]) for g in range(G)]))


def difference_term(P, s, outer, G, inner):
    P = np.asarray(P, np.float64).reshape(-1)
    s = np.asarray(s, np.float64).reshape(-1)
    gid = group_ids(outer, G, inner)
    return float(np.mean(np.abs(P - P / s[gid])))


def inverse_term(s):
    s = np.asarray(s, np.float64).reshape(-1)
    return float(np.mean(1.0 / np.where(s == 0.0, EPS_F32, s)))


def penalty(kind, layers):
    """layers: list of (K, sK, descK, b, sb, descb); desc = (outer, G, inner)."""
    num = 0.0
    den = 0.0
    for K, sK, dK, b, sb, db in layers:
        nK, nb = float(np.size(K)), float(np.size(b))
        if kind == "maxbin":
            tK, tb = maxbin_term(K, sK, *dK), maxbin_term(b, sb, *db)
        elif kind == "difference":
            tK, tb = difference_term(K, sK, *dK), difference_term(b, sb, *db)
        elif kind == "inverse":
            tK, tb = inverse_term(sK), inverse_term(sb)
        else:
            raise ValueError(kind)
        num += nK * tK + nb * tb
        den += nK + nb
    return num / den


# --------------------------------------------------------------------------- #
#  Float64 gradients of the penalty terms, with the condition-number yardstick.
#  A float32 implementation that sums n signed terms t_i cannot be held to a bound relative to |sum t_i| (the
#  terms cancel); the right yardstick is sum |t_i|.  Every function returns, next to the float64 gradient, the
#  float64 sum of the absolute values of the terms that gradient is a sum of (for single-term quantities that is
#  |gradient| itself).  tests/_bounds.py asserts |got - f64| <= 1e-5 * sum|t_i|.
#  Formulas: chapter3.tex:242-282 (MaxBin), :289-311 (Inverse), :317-337 (Difference); d|x| = sign(x);
#  reduce_max routes its gradient evenly to the tied maxima (TensorFlow math_grad._MinOrMaxGrad).
# --------------------------------------------------------------------------- #
def maxbin_term_grads(P, s, c, outer, G, inner):
    """d(c * mean_g max_i |P_i|/s_g): returns (dP, ds, ds_abs)."""
    P = np.asarray(P, np.float64).reshape(-1)
    s = np.asarray(s, np.float64).reshape(-1)
    gid = group_ids(outer, G, inner)
    t = np.abs(P) / s[gid]
    m = np.full(G, -np.inf)
    np.maximum.at(m, gid, t)
    ind = t == m[gid]
    cnt = np.bincount(gid, weights=ind.astype(np.float64), minlength=G)
    g = ind / cnt[gid] * (float(c) / G)                         # gradient with respect to t
    dP = g / s[gid] * np.sign(P)
    terms = -g * t / s[gid]
    ds = np.bincount(gid, weights=terms, minlength=G)
    ds_abs = np.bincount(gid, weights=np.abs(terms), minlength=G)
    return dP, ds, ds_abs


def difference_term_grads(P, s, c, outer, G, inner, with_dP_abs=False):
    """d(c * mean |P - P/s|): returns (dP, ds, ds_abs) [, dP_abs].  dP_i = g_i - g_i/s is itself a sum of TWO signed terms
    (the gradient through P and the one through P/s, which autodiff adds): they cancel when s is near 1, so its yardstick
    is |g_i| + |g_i/s|, not |dP_i|."""
    P32 = np.asarray(P, np.float32).reshape(-1)
    s32 = np.asarray(s, np.float32).reshape(-1)
    P = P32.astype(np.float64)
    s = s32.astype(np.float64)
    gid = group_ids(outer, G, inner)
    pq = P / s[gid]
    # tf.sign of the float32 residual (custom_loss_functions.py:172-176 runs in float32): WHICH sign an element gets -- and
    # whether it is exactly zero, which happens for s within an ulp of 1, where fl(P / s) == P -- is the reference's float32
    # decision, as the MaxBin tie split is; the magnitudes around it are float64.  (Seen in a soak: 3 of 1580 elements with a
    # zero float32 residual, gradient 0 in float32 and in the kernel, 1.8e-7 with a float64 sign.)
    with np.errstate(all="ignore"):
        sgn = np.sign(P32 - P32 / s32[gid]).astype(np.float64)
    g = sgn * (float(c) / P.size)
    dP = g - g / s[gid]
    terms = g * pq / s[gid]
    ds = np.bincount(gid, weights=terms, minlength=G)
    ds_abs = np.bincount(gid, weights=np.abs(terms), minlength=G)
    if with_dP_abs:
        return dP, ds, ds_abs, np.abs(g) + np.abs(g / s[gid])
    return dP, ds, ds_abs


def inverse_term_grads(s, c):
    """d(c * mean 1/where(s == 0, eps, s)): returns (ds, ds_abs); no gradient reaches an exactly zero scale."""
    s = np.asarray(s, np.float64).reshape(-1)
    s_nz = np.where(s == 0.0, EPS_F32, s)
    ds = np.where(s == 0.0, 0.0, -(float(c) / s.size) / (s_nz * s_nz))
    return ds, np.abs(ds)


def term_abs(kind, P, s, outer, G, inner):
    """sum of |terms| of the float64 VALUE of one tensor term (every term is non-negative: it equals the value)."""
    if kind == "maxbin":
        return abs(maxbin_term(P, s, outer, G, inner))
    if kind == "difference":
        return abs(difference_term(P, s, outer, G, inner))
    return abs(inverse_term(s))


def penalty_grads(kind, layers, rate):
    """Gradients of ``rate * penalty(kind, layers)`` for every tensor: list of dicts (one per layer) with the float64
    arrays dK, dsK, dsK_abs, db, dsb, dsb_abs (dK/db are None for the inverse penalty, which reads only scales).
    The coefficient of tensor i's term is rate * numel_i / sum(numel)  (custom_loss_functions.py:110-116)."""
    den = sum(float(np.size(K)) + float(np.size(b)) for K, sK, dK, b, sb, db in layers)
    out = []
    for K, sK, dK, b, sb, db in layers:
        cK, cb = rate * float(np.size(K)) / den, rate * float(np.size(b)) / den
        e = {}
        if kind == "maxbin":
            e["dK"], e["dsK"], e["dsK_abs"] = maxbin_term_grads(K, sK, cK, *dK)
            e["db"], e["dsb"], e["dsb_abs"] = maxbin_term_grads(b, sb, cb, *db)
        elif kind == "difference":
            e["dK"], e["dsK"], e["dsK_abs"], e["dK_abs"] = difference_term_grads(K, sK, cK, *dK, with_dP_abs=True)
            e["db"], e["dsb"], e["dsb_abs"], e["db_abs"] = difference_term_grads(b, sb, cb, *db, with_dP_abs=True)
        elif kind == "inverse":
            e["dK"] = e["db"] = None
            e["dsK"], e["dsK_abs"] = inverse_term_grads(sK, cK)
            e["dsb"], e["dsb_abs"] = inverse_term_grads(sb, cb)
        else:
            raise ValueError(kind)
        out.append(e)
    return out
