"""Stand-in for cvxopt 1.2.7 (pinned by the reference's marl_cav.yml:14, not installable here: no network,
no wheel in /opt/wheelhouse, no conda package -- probed again in round 2).

`solvers.qp` has two modes (`solvers.mode`, default from $MM_REF_QP, else "exact"):

  "exact"   the EXACT KKT solution of the reference's 3-variable shield QP (P = diag(1, 1, 1e18), q = 0,
            rows `a*d - s <= h_k`, `d <= hi`, `-d <= -lo`) in closed form;
  "coneqp"  coneqp.py: a restatement of cvxopt's own interior-point algorithm for this problem class
            (Mehrotra predictor-corrector, NT scaling, chol2 KKT solver, default tolerances), returning
            the ITERATE cvxopt's algorithm stops at, with its status and iteration count.

Fixtures say which mode made them (meta["qp_solver"]).  "coneqp" follows the published algorithm; the
last bits of the real binary (BLAS/LAPACK summation order) are not reproducible without it -> the raw
solver output stays "parity unpinned against the cvxopt binary", pinned against this restatement.
"""
import os

import numpy as np

from . import coneqp as _coneqp


class matrix(object):
    def __init__(self, arr, tc="d"):
        self.a = np.array(arr, dtype=float)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)


def exact_kkt(G, h):
    """Closed-form minimiser for the reference's G structure (asserted)."""
    # rows 0 (and 3 when present): a*d + 0*e - s <= h_k ; row 1: d <= hi ; row 2: -d <= -lo
    a = G[0, 0]
    hc = h[0]
    if G.shape[0] == 4:
        assert G[3, 0] == a and G[3, 2] == -1.0 and G[3, 1] == 0.0
        hc = min(hc, h[3])
    assert G[0, 2] == -1.0 and G[0, 1] == 0.0
    assert tuple(G[1]) == (1.0, 0.0, 0.0) and tuple(G[2]) == (-1.0, 0.0, 0.0)
    hi, lo = h[1], -h[2]
    if a > 0:
        d = min(0.0, hc / a)
    elif a < 0:
        d = max(0.0, hc / a)
    else:
        d = 0.0
    d = min(max(d, lo), hi)
    s = max(0.0, a * d - hc)
    return np.array([d, 0.0, s])


def ipm(P, q, G, h):
    """coneqp.py on numpy inputs -> (x[3], status str, iterations)."""
    P = np.asarray(P, dtype=float)
    assert np.count_nonzero(P - np.diag(np.diagonal(P))) == 0, "the restatement takes a diagonal P"
    r = _coneqp.coneqp(np.diagonal(P).tolist(), np.asarray(q, dtype=float).ravel().tolist(),
                       np.asarray(G, dtype=float).tolist(), np.asarray(h, dtype=float).ravel().tolist())
    return np.array(r["x"], dtype=float), r["status"], r["iterations"]


class _Solvers(object):
    options = {}
    mode = os.environ.get("MM_REF_QP", "exact")
    log = None  # optional list; when set, every QP is appended as (G, h, x_returned, x_other, status, iterations):
    #             x_other is the solution of the mode that was NOT returned, status / iterations are the IPM's

    @staticmethod
    def qp(P, q, G, h, A=None, b=None):
        assert A is None and b is None
        Gn = np.asarray(G.a, dtype=float)
        hn = np.asarray(h.a, dtype=float).ravel()
        mode = solvers.mode
        x_ex = exact_kkt(Gn, hn)
        if mode == "coneqp" or solvers.log is not None:
            x_ip, status, iters = ipm(P.a, q.a, Gn, hn)
        if mode == "coneqp":
            x, other = x_ip, x_ex
        else:
            x, other = x_ex, (x_ip if solvers.log is not None else None)
            if solvers.log is None:
                status, iters = "optimal", 0
        if solvers.log is not None:
            solvers.log.append((Gn.copy(), hn.copy(), x.copy(), other.copy(), status, iters))
        return {"x": x, "status": status if mode == "coneqp" else "optimal", "iterations": iters}


solvers = _Solvers()
