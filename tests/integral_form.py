"""A THIRD, differently shaped restatement of the LoCoHD score (test infrastructure, CPU, NumPy only).

oracle/locohd_oracle.c follows the reference's two-pointer merge loop line by line (/root/reference/src/locohd.rs:97-223)
and the HIP kernels were written by the same hand from the same reading: a common-mode misreading of a branch the
reference's own known answers do not pin (category weights, the non-default statistical distances, from_coords) would
pass every oracle-vs-HIP test.  This module computes the same quantity from its mathematical DEFINITION instead
(/root/reference/loco_hd/loco_hd.pyi:216-231, README "mathematical background"):

    S = integral_0^inf w(r) H(r) dr = sum_j [F(u_{j+1}) - F(u_j)] * H(A(u_j), B(u_j)),   u_0 = 0 < u_1 < ... , u_{J+1} = inf

with u_j the DISTINCT distances occurring in either environment, A(r) / B(r) the weighted category counts of the points at
distance <= r (closed ball, anchors included), H the statistical distance of the two normalised count vectors and F the CDF
of the weight function.  No merge loop, no event-by-event state: the counts come from one-hot prefix sums looked up with
`searchsorted`, everything after the environment membership test is np.longdouble (80-bit on x86), and every formula is
written from the published definition of the distribution / divergence rather than from the Rust code's operation order.
The two shapes agree exactly where ties make the reference's zero-width intervals vanish (finite H).
"""
from __future__ import annotations

import numpy as np

LD = np.longdouble


# ---- weight-function CDFs (/root/reference/src/locohd/weight_function/cdfs.rs, from their closed forms) ----------------
def cdf(name: str, params, x):
    x = np.asarray(x, dtype=LD)
    p = [LD(v) for v in params]
    inf = np.isinf(x)
    xf = np.where(inf, LD(0), x)
    if name == "hyper_exp":
        k = len(p) // 2
        a, b = np.asarray(p[:k], dtype=LD), np.asarray(p[k:], dtype=LD)
        out = LD(1) - (a[None, :] * np.exp(-b[None, :] * xf[..., None])).sum(-1) / a.sum()
    elif name == "dagum":
        A, B, P = p
        with np.errstate(divide="ignore", over="ignore"):  # x = 0: 0^-A = inf, (1 + inf)^-P = 0 (IEEE pow, like powf)
            out = (LD(1) + np.power(xf / B, -A)) ** (-P)
    elif name == "uniform":
        lo, hi = p
        out = np.clip((xf - lo) / (hi - lo), LD(0), LD(1))
    elif name == "kumaraswamy":
        lo, hi, A, B = p
        z = np.clip((xf - lo) / (hi - lo), LD(0), LD(1))
        out = LD(1) - (LD(1) - z ** A) ** B
    else:
        raise ValueError(name)
    return np.where(inf, LD(1), out)


# ---- statistical distances (/root/reference/src/locohd/pmf/statistical_distances.rs:4-78, from their definitions) --------
def distance(name: str, params, P, Q):
    """P, Q: [..., C] longdouble probability vectors; returns [...]"""
    if name == "Hellinger":
        e = LD(params[0])
        return (np.abs(P ** (1 / e) - Q ** (1 / e)) ** e).sum(-1) ** (1 / e) / LD(2) ** (1 / e)
    if name == "Kolmogorov-Smirnov":
        return np.abs(P - Q).max(-1)
    if name == "Kullback-Leibler":
        eps = LD(params[0])
        return (P * (np.log(P + eps) - np.log(Q + eps))).sum(-1)
    if name == "Renyi":
        alpha, eps = float(params[0]), LD(params[1])
        if alpha == 1.0:
            return distance("Kullback-Leibler", [eps], P, Q)
        ratio = (P + eps) / (Q + eps)
        if np.isinf(alpha):
            return np.log(ratio.max(-1))
        if alpha == 0.0:
            with np.errstate(divide="ignore"):
                return -np.log(np.where(P > 0, Q, LD(0)).sum(-1))
        a = LD(params[0])
        return np.log((P * ratio ** (a - 1)).sum(-1)) / (a - 1)
    raise ValueError(name)


# ---- the score of one anchor pair from two UNSORTED environments ----------------------------------------------------------
def score(cat_a, dist_a, cat_b, dist_b, n_categories, wf, sd=("Hellinger", [2.0]), category_weights=None):
    """cat_x: category index per point, dist_x: distance per point (any order; the anchor is the point at distance 0 that the
    caller put first or anywhere -- the definition does not single it out)."""
    w = np.ones(n_categories, dtype=LD) if category_weights is None else np.asarray(category_weights, dtype=LD)
    sides = []
    for cat, dist in ((cat_a, dist_a), (cat_b, dist_b)):
        dist = np.asarray(dist, dtype=np.float64)
        order = np.argsort(dist, kind="stable")
        d = dist[order]
        onehot = np.zeros((len(d) + 1, n_categories), dtype=LD)
        onehot[np.arange(1, len(d) + 1), np.asarray(cat)[order]] = 1
        sides.append((d, np.cumsum(onehot, axis=0) * w[None, :]))  # prefix[k] = weighted counts of the k nearest points
    (da, pa), (db, pb) = sides
    u = np.unique(np.concatenate([[0.0], da, db]))  # distinct breakpoints, ascending; u_0 = 0
    A = pa[np.searchsorted(da, u, side="right")]    # counts at distance <= u_j
    B = pb[np.searchsorted(db, u, side="right")]
    H = distance(sd[0], sd[1], A / A.sum(-1, keepdims=True), B / B.sum(-1, keepdims=True))
    F = cdf(wf[0], wf[1], np.concatenate([u, [np.inf]]))
    return float((np.diff(F) * H).sum())


# ---- drivers: the environments of from_primitives / from_coords by brute force --------------------------------------------
def _sqdist(p, xyz):
    d = xyz - p[None, :]
    return d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]  # float64, the reference's summation order (utils.rs:1-8)


def from_primitives(cat_a, xyz_a, tag_a, cat_b, xyz_b, tag_b, pairs, thr, n_categories, wf, sd=("Hellinger", [2.0]),
                    category_weights=None, accept_same=True):
    """Environment of anchor i = every point p with |p - anchor|^2 < thr^2 that IS the anchor or satisfies the tag rule
    (accept_same: (tag == anchor tag) == accept_same), /root/reference/src/locohd.rs:514-542."""
    out = []
    xyz_a, xyz_b = np.asarray(xyz_a, dtype=np.float64), np.asarray(xyz_b, dtype=np.float64)
    thr2 = np.float64(thr) * np.float64(thr) if np.isfinite(thr) else np.inf

    def env(i, cat, xyz, tag):
        d2 = _sqdist(xyz[i], xyz)
        keep = d2 < thr2
        if tag is not None:
            rule = (np.asarray(tag) == tag[i]) == bool(accept_same)
            rule[i] = True
            keep &= rule
        return np.asarray(cat)[keep], np.sqrt(d2[keep])

    for i, j in pairs:
        ca, da = env(int(i), cat_a, xyz_a, tag_a)
        cb, db = env(int(j), cat_b, xyz_b, tag_b)
        out.append(score(ca, da, cb, db, n_categories, wf, sd, category_weights))
    return np.asarray(out)


def from_coords(cat_a, xyz_a, cat_b, xyz_b, n_categories, wf, sd=("Hellinger", [2.0]), category_weights=None):
    """/root/reference/src/locohd.rs:463-476: every point is an anchor, the environment is the whole structure."""
    xyz_a, xyz_b = np.asarray(xyz_a, dtype=np.float64), np.asarray(xyz_b, dtype=np.float64)
    return np.asarray([score(cat_a, np.sqrt(_sqdist(xyz_a[i], xyz_a)), cat_b, np.sqrt(_sqdist(xyz_b[i], xyz_b)), n_categories, wf, sd,
                             category_weights) for i in range(len(xyz_a))])
