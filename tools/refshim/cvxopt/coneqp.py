"""A restatement of cvxopt 1.2.x `solvers.coneqp` for the case the reference uses it in
(highway_env/vehicle/safety/cbf.py:128-135: `solvers.qp(P, q, G, h)` with dense P, q, G, h, no
equality constraints, default options except show_progress) -- i.e.

    minimize (1/2) x'Px + q'x   subject to   G x + s = h,  s >= 0        ('l' cone only)

solved by the primal-dual path-following method of the cvxopt documentation ("Cone Programming",
algorithm of coneqp; L. Vandenberghe, "The CVXOPT linear and quadratic cone program solvers", 2010):
Mehrotra predictor-corrector, Nesterov-Todd scaling (for the 'l' cone: W = diag(sqrt(s./z))),
step factor STEP = 0.99, sigma = (1 - step + (ds'dz/gap) step^2)^3 clipped to [0, 1]... in the order
cvxopt's own Python source performs it, default tolerances abstol 1e-7 / reltol 1e-6 / feastol 1e-7,
maxiters 100, no iterative refinement (cvxopt's default for problems without 'q'/'s' cones), and the
'chol2' KKT solver cvxopt picks for such problems: Cholesky of S = P + G' W^-1 W^-T G.

cvxopt itself is NOT installable here (no network; pinned 1.2.7 by the reference's marl_cav.yml:14),
so this file was written from the published algorithm, not checked against the binary.  What is NOT
reproduced is the last-bit behaviour of the BLAS/LAPACK cvxopt links against (summation order and
FMA use inside dgemv / dsyrk / dpotrf / dnrm2 are implementation-defined); here every sum runs in
ascending index order with one rounding per operation, `x ** 3` is x*x*x and 1/d is a division --
all of which moves an iterate by a few ulp, far inside the 1e-7 tolerances that define the answer.
The same operation order is restated in C (oracle/mm_oracle.c: qp_ipm) and in HIP
(marl-mass_amd/csrc/mm_device.h: qp_ipm), so the three agree bit for bit.

Pure Python floats on purpose (IEEE double, one rounding per operation, no numpy reassociation).
"""
import math

MAXITERS = 100
ABSTOL = 1e-7
RELTOL = 1e-6
FEASTOL = 1e-7
STEP = 0.99


def _dot(a, b):
    t = 0.0
    for x, y in zip(a, b):
        t += x * y
    return t


def _nrm2(a):
    return math.sqrt(_dot(a, a))


class _Chol2(object):
    """misc.kkt_chol2 for dense G, no equality constraints: factor(W) -> solve(x, z)."""

    def __init__(self, Pd, G):
        self.Pd, self.G = Pd, G
        self.n, self.m = len(Pd), len(G)

    def factor(self, di):
        n, m, G = self.n, self.m, self.G
        Gs = [[di[k] * G[k][j] for j in range(n)] for k in range(m)]  # Gs = W^-1 G
        S = [[0.0] * n for _ in range(n)]
        for i in range(n):  # syrk: lower triangle of Gs' Gs, then S += P
            for j in range(i + 1):
                t = 0.0
                for k in range(m):
                    t += Gs[k][i] * Gs[k][j]
                S[i][j] = t
            S[i][i] = S[i][i] + self.Pd[i]
        L = [[0.0] * n for _ in range(n)]  # potrf (lower): left-looking, divisions
        for j in range(n):
            t = S[j][j]
            for k in range(j):
                t -= L[j][k] * L[j][k]
            if not (t > 0.0):
                raise ArithmeticError("S is not positive definite")
            L[j][j] = math.sqrt(t)
            for i in range(j + 1, n):
                t = S[i][j]
                for k in range(j):
                    t -= L[i][k] * L[j][k]
                L[i][j] = t / L[j][j]
        self.Gs, self.L, self.di = Gs, L, di

    def solve(self, x, z):
        """[P G'; G -W'W] [ux; W^-1 uz] = [bx; bz]: x, z hold bx, bz on entry and ux, uz on exit."""
        n, m, Gs, L, di = self.n, self.m, self.Gs, self.L, self.di
        for k in range(m):  # z := W^-T bz
            z[k] = z[k] * di[k]
        for j in range(n):  # x := x + Gs' z  (gemv 'T', beta = 1)
            t = 0.0
            for k in range(m):
                t += Gs[k][j] * z[k]
            x[j] = x[j] + t
        for j in range(n):  # trsv: L x = x
            x[j] = x[j] / L[j][j]
            for i in range(j + 1, n):
                x[i] = x[i] - x[j] * L[i][j]
        for j in range(n - 1, -1, -1):  # trsv 'T': L' x = x
            t = x[j]
            for i in range(n - 1, j, -1):
                t -= L[i][j] * x[i]
            x[j] = t / L[j][j]
        for k in range(m):  # z := Gs x - z  (gemv 'N', beta = -1)
            t = -z[k]
            for j in range(n):
                t += x[j] * Gs[k][j]
            z[k] = t


def coneqp(Pd, q, G, h):
    """Pd: diagonal of P (the reference's P is diagonal, cbf.py:40-44); q: n; G: m x n rows; h: m.
    Returns dict(x, s, z, status, iterations, gap)."""
    n, m = len(q), len(h)
    Pd = [float(v) for v in Pd]
    q = [float(v) for v in q]
    G = [[float(v) for v in row] for row in G]
    h = [float(v) for v in h]
    kkt = _Chol2(Pd, G)

    resx0 = max(1.0, _nrm2(q))
    resz0 = max(1.0, _nrm2(h))

    # initial point: [P G'; G -I] [x; z] = [-q; h], s = -z, shifted into the cone
    kkt.factor([1.0] * m)
    x = [-v for v in q]
    z = list(h)
    kkt.solve(x, z)
    s = [-v for v in z]
    nrms = _nrm2(s)
    ts = max(-v for v in s)
    if ts >= -1e-8 * max(nrms, 1.0):
        a = 1.0 + ts
        s = [v + a for v in s]
    nrmz = _nrm2(z)
    tz = max(-v for v in z)
    if tz >= -1e-8 * max(nrmz, 1.0):
        a = 1.0 + tz
        z = [v + a for v in z]

    gap = _dot(s, z)
    d = di = lmbda = None
    for iters in range(MAXITERS + 1):
        # rx = P x + q + G' z ; f0 = (1/2)(x'(Px + q) + x'q)
        rx = [q[j] + Pd[j] * x[j] for j in range(n)]
        f0 = 0.5 * (_dot(x, rx) + _dot(x, q))
        for j in range(n):
            t = 0.0
            for k in range(m):
                t += G[k][j] * z[k]
            rx[j] = rx[j] + t
        resx = _nrm2(rx)
        # rz = s + G x - h
        rz = [0.0] * m
        for k in range(m):
            t = s[k] - h[k]
            for j in range(n):
                t += x[j] * G[k][j]
            rz[k] = t
        resz = _nrm2(rz)
        pcost = f0
        dcost = f0 + _dot(z, rz) - gap
        if pcost < 0.0:
            relgap = gap / -pcost
        elif dcost > 0.0:
            relgap = gap / dcost
        else:
            relgap = None
        pres = resz / resz0
        dres = resx / resx0
        if (pres <= FEASTOL and dres <= FEASTOL and
                (gap <= ABSTOL or (relgap is not None and relgap <= RELTOL))) or iters == MAXITERS:
            status = "unknown" if iters == MAXITERS else "optimal"
            return dict(x=x, s=s, z=z, status=status, iterations=iters, gap=gap)

        if iters == 0:  # misc.compute_scaling ('l' block)
            d = [math.sqrt(s[k] / z[k]) for k in range(m)]
            di = [1.0 / d[k] for k in range(m)]
            lmbda = [math.sqrt(s[k] * z[k]) for k in range(m)]
        lmbdasq = [v * v for v in lmbda]
        try:
            kkt.factor(di)
        except ArithmeticError:
            return dict(x=x, s=s, z=z, status="unknown", iterations=iters, gap=gap)

        def f4(bx, bz, bs):
            for k in range(m):  # s := lmbda o\ bs ; z := bz - W' s
                bs[k] = bs[k] / lmbda[k]
                bz[k] = bz[k] - d[k] * bs[k]
            kkt.solve(bx, bz)
            for k in range(m):  # us = lmbda o\ bs - uz
                bs[k] = bs[k] - bz[k]

        mu = gap / m
        sigma, eta = 0.0, 0.0
        ws3 = None
        for i in (0, 1):
            ds = [0.0] * m
            for k in range(m):
                t = 0.0
                if i == 1:
                    t = t - ws3[k]
                t = t - lmbdasq[k]
                ds[k] = t + sigma * mu
            dx = [(-1.0 + eta) * rx[j] for j in range(n)]
            dz = [(-1.0 + eta) * rz[k] for k in range(m)]
            f4(dx, dz, ds)
            dsdz = _dot(ds, dz)
            if i == 0:  # ds o dz for the Mehrotra correction
                ws3 = [ds[k] * dz[k] for k in range(m)]
            for k in range(m):  # scale2: into the lambda-scaled frame
                ds[k] = ds[k] / lmbda[k]
                dz[k] = dz[k] / lmbda[k]
            ts = max(-v for v in ds)
            tz = max(-v for v in dz)
            t = max(0.0, ts, tz)
            if t == 0:
                step = 1.0
            elif i == 0:
                step = min(1.0, 1.0 / t)
            else:
                step = min(1.0, STEP / t)
            if i == 0:
                sg = min(1.0, max(0.0, 1.0 - step + dsdz / gap * (step * step)))
                sigma = sg * sg * sg
                eta = 0.0

        for j in range(n):
            x[j] = x[j] + step * dx[j]
        for k in range(m):  # updated iterates in the current scaling, then misc.update_scaling
            ds[k] = (step * ds[k] + 1.0) * lmbda[k]
            dz[k] = (step * dz[k] + 1.0) * lmbda[k]
            ds[k] = math.sqrt(ds[k])
            dz[k] = math.sqrt(dz[k])
            d[k] = d[k] * ds[k] / dz[k]
            di[k] = 1.0 / d[k]
            lmbda[k] = ds[k] * dz[k]
            s[k] = d[k] * lmbda[k]
            z[k] = di[k] * lmbda[k]
        gap = _dot(lmbda, lmbda)
    raise AssertionError("unreachable")
