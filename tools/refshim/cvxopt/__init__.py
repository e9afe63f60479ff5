"""Stand-in for cvxopt 1.2.7 (pinned by the reference's marl_cav.yml, not installable here).

`solvers.qp` returns the EXACT KKT solution of the reference's 3-variable shield QP
(P = diag(1, 1, 1e18), q = 0, rows `a*d - s <= h_k`, `d <= hi`, `-d <= -lo`) in closed form.
Consequence, stated wherever fixtures made with this module are used: shield goldens are
"reference state/G/h assembly + exact-KKT solve"; the raw interior-point iterate of the real
cvxopt (abstol 1e-7 / reltol 1e-6) is NOT reproduced -> "parity unpinned" for the solver output.
"""
import numpy as np


class matrix(object):
    def __init__(self, arr, tc="d"):
        self.a = np.array(arr, dtype=float)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)


class _Solvers(object):
    options = {}
    log = None  # optional list; when set, every (G, h, x) triple is appended

    @staticmethod
    def qp(P, q, G, h, A=None, b=None):
        G = np.asarray(G.a, dtype=float)
        h = np.asarray(h.a, dtype=float).ravel()
        # rows 0 (and 3 when present): a*d + 0*e - s <= h_k ; row 1: d <= hi ; row 2: -d <= -lo
        a = G[0, 0]
        hc = h[0]
        if G.shape[0] == 4:
            assert G[3, 0] == a and G[3, 2] == -1.0
            hc = min(hc, h[3])
        assert G[0, 2] == -1.0 and G[1, 0] == 1.0 and G[2, 0] == -1.0
        hi, lo = h[1], -h[2]
        if a > 0:
            d = min(0.0, hc / a)
        elif a < 0:
            d = max(0.0, hc / a)
        else:
            d = 0.0
        d = min(max(d, lo), hi)
        s = max(0.0, a * d - hc)
        x = np.array([d, 0.0, s])
        if solvers.log is not None:
            solvers.log.append((G.copy(), h.copy(), x.copy()))
        return {"x": x, "status": "optimal"}


solvers = _Solvers()
