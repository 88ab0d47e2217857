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
