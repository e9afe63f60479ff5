/*
 * mm_qp.h -- the shield's QP through cvxopt's interior-point algorithm (MM_QP_IPM, the reference-faithful mode).
 *
 * Reference call: `solvers.qp(self.P, self.q, G, h)` (highway_env/vehicle/safety/cbf.py:128-135) ->
 * cvxopt 1.2.x coneqp with default options: Mehrotra predictor-corrector, Nesterov-Todd scaling, 'chol2'
 * KKT solver, abstol 1e-7 / reltol 1e-6 / feastol 1e-7, at most 100 iterations, no iterative refinement.
 * The problem is always
 *     minimize 1/2 (d^2 + e^2 + 1e18 s^2)    over u = (d, e, s)
 *     subject to  [a 0 -1] u <= h0,  d <= h1,  -d <= h2,  and with rows == 4:  [a 0 -1] u <= h3
 * (P, q: cbf.py:40-47; G, h: cbf.py:288-322 / 386-422).  This header is that algorithm specialised to this
 * sparsity pattern: the zero entries of G and P are dropped, every remaining operation is kept in the order
 * of the general dense algorithm (tools/refshim/cvxopt/coneqp.py on the reference side, qp_ipm in
 * oracle/mm_oracle.c), so the result is bit-identical to those for every input the reference produces
 * (tests/test_qp_ipm.py).  `e` decouples (it stays 0) and is not carried.
 *
 * Form: a resumable state machine -- mm_qp_start (initial point), then per iteration mm_qp_top (residuals and
 * cvxopt's stopping test) and mm_qp_bottom (scaling, KKT factorisation, predictor + corrector, step, update).
 * The HIP step kernels run ONE wave-wide loop over it in which every lane is at its own iteration of its own QP:
 * a lane whose QP stops posts its result and the lanes that were waiting for that decision (MASS: a follower reads
 * its leader's decided acceleration) start theirs in the same loop, so neither a slow QP of another env nor the
 * sweep order idles the wave longer than the dependency chain itself (marl-mass_amd/csrc/mm_kernels.hip).
 * mm_qp_ipm_cbf below is the plain start / top / bottom loop for one QP (oracle, qp_kernel, literal sweep).
 * No MFMA: the largest dense object is a 3x3 Cholesky factor with two structural zeros.
 *
 * Divisions by the same denominator inside an iteration (the four scaling entries lambda_k: six each; the Cholesky
 * pivots: five and four) go through MM_QP_RCP / MM_QP_DIVR: by default a plain IEEE division; the device build
 * (mm_device.h) defines them as ONE refined reciprocal per denominator and a 4-instruction correctly rounded
 * quotient per numerator -- the same quotient bits (tests/test_hip_parity.py::test_qp_division_forms), a third of the
 * instructions.
 *
 * The includer defines MMM_FN (`static inline`, or `__device__ __forceinline__`); needs sqrt, fmin, fmax.
 * Build with -ffp-contract=off: a*b+c must keep two roundings here as in Python.
 */
#ifndef MM_QP_H
#define MM_QP_H

#ifndef MMM_FN
#define MMM_FN static inline
#endif

#define MM_QP_MAXITERS 100
#define MM_QP_ABSTOL 1e-7
#define MM_QP_RELTOL 1e-6
#define MM_QP_FEASTOL 1e-7
#define MM_QP_STEP 0.99

/* a / b for several a with one b: R = MM_QP_RCP(b) once, then MM_QP_DIVR(a, b, R) == a / b bit for bit */
#ifndef MM_QP_RCP
#define MM_QP_RCP(b) (b)
#define MM_QP_DIVR(a, b, R) ((a) / (b))
#endif

typedef struct MMQpKkt {  /* misc.kkt_chol2 factor: S = P + Gs'Gs = L L', Gs = W^-1 G */
  double g0, g3;          /* Gs[0][0], Gs[3][0] = di * a  (Gs[k][2] = -di[k] for k = 0, 3; Gs[1][0] = di1, Gs[2][0] = -di2) */
  double l00, l20, l22;   /* L (L11 = 1, L10 = L21 = 0) */
  double r00, r22;        /* MM_QP_RCP of the pivots l00, l22 */
  double di[4];
} MMQpKkt;

typedef struct MMQpState {
  double a, h[4], resz0;        /* the problem; resz0 = max(1, |h|) */
  double x0, x2, gap;           /* primal iterate (d, slack) and s'z */
  double s[4], z[4];            /* slacks and multipliers */
  double d[4], di[4], lmbda[4]; /* Nesterov-Todd scaling W = diag(d), W^-1, lambda = W^-1 s = W z (valid from the first mm_qp_bottom on) */
  int m4, iters;
} MMQpState;
typedef struct MMQpRes { double rx0, rx2, rz[4]; } MMQpRes;  /* what mm_qp_top hands to mm_qp_bottom */

MMM_FN int mm_qp_factor(MMQpKkt *f, double a, int m4, const double *di) {
  f->di[0] = di[0]; f->di[1] = di[1]; f->di[2] = di[2]; f->di[3] = m4 ? di[3] : 0.0;
  f->g0 = di[0] * a;
  f->g3 = m4 ? di[3] * a : 0.0;
  double s00 = f->g0 * f->g0 + di[1] * di[1] + di[2] * di[2];
  if (m4) s00 = s00 + f->g3 * f->g3;
  s00 = s00 + 1.0;
  double s20 = 0.0 + (-di[0]) * f->g0;
  if (m4) s20 = s20 + (-di[3]) * f->g3;
  double s22 = di[0] * di[0];
  if (m4) s22 = s22 + di[3] * di[3];
  s22 = s22 + 1e18;
  if (!(s00 > 0.0)) return 0;
  f->l00 = sqrt(s00);
  f->r00 = MM_QP_RCP(f->l00);
  f->l20 = MM_QP_DIVR(s20, f->l00, f->r00);
  const double t = s22 - f->l20 * f->l20;
  if (!(t > 0.0)) return 0;
  f->l22 = sqrt(t);
  f->r22 = MM_QP_RCP(f->l22);
  return 1;
}
/* [P G'; G -W'W][ux; W^-1 uz] = [bx; bz]; (x0, x2, z) hold the right-hand side on entry, the solution on exit */
MMM_FN void mm_qp_solve(const MMQpKkt *f, int m4, double *x0, double *x2, double *z) {
  z[0] = z[0] * f->di[0]; z[1] = z[1] * f->di[1]; z[2] = z[2] * f->di[2];
  if (m4) z[3] = z[3] * f->di[3];
  double t = 0.0 + f->g0 * z[0];
  t = t + f->di[1] * z[1];
  t = t + (-f->di[2]) * z[2];
  if (m4) t = t + f->g3 * z[3];
  double a0 = *x0 + t;
  t = 0.0 + (-f->di[0]) * z[0];
  if (m4) t = t + (-f->di[3]) * z[3];
  double a2 = *x2 + t;
  a0 = MM_QP_DIVR(a0, f->l00, f->r00);          /* trsv: L x = x */
  a2 = a2 - a0 * f->l20;
  a2 = MM_QP_DIVR(a2, f->l22, f->r22);
  a2 = MM_QP_DIVR(a2, f->l22, f->r22);          /* trsv 'T': L' x = x */
  t = a0 - f->l20 * a2;
  a0 = MM_QP_DIVR(t, f->l00, f->r00);
  t = -z[0]; t = t + a0 * f->g0; t = t + a2 * (-f->di[0]); z[0] = t;
  t = -z[1]; t = t + a0 * f->di[1]; z[1] = t;
  t = -z[2]; t = t + a0 * (-f->di[2]); z[2] = t;
  if (m4) { t = -z[3]; t = t + a0 * f->g3; t = t + a2 * (-f->di[3]); z[3] = t; }
  *x0 = a0; *x2 = a2;
}
MMM_FN double mm_qp_dot(const double *p, const double *q, int m4) {
  double t = 0.0 + p[0] * q[0];
  t = t + p[1] * q[1];
  t = t + p[2] * q[2];
  if (m4) t = t + p[3] * q[3];
  return t;
}
MMM_FN double mm_qp_maxneg(const double *p, int m4) {  /* misc.max_step for the 'l' cone */
  double t = -p[0];
  if (-p[1] > t) t = -p[1];
  if (-p[2] > t) t = -p[2];
  if (m4 && -p[3] > t) t = -p[3];
  return t;
}

/* The initial point (coneqp: solve the KKT system with W = I, shift s and z into the cone).  Returns 0 when cvxopt would
 * raise ValueError("Rank(A) < p or Rank([P; A; G]) < n") -- only a NaN input does that here; q->x0 is then NaN. */
MMM_FN int mm_qp_start(MMQpState *q, double a, double h0, double h1, double h2, double h3, int rows) {
  const int m4 = rows == 4;
  q->a = a; q->m4 = m4; q->iters = 0;
  q->h[0] = h0; q->h[1] = h1; q->h[2] = h2; q->h[3] = m4 ? h3 : 0.0;
  q->resz0 = fmax(1.0, sqrt(mm_qp_dot(q->h, q->h, m4)));  /* resx0 = max(1, |q|) = 1 */
  MMQpKkt kkt;
  const double one[4] = {1.0, 1.0, 1.0, 1.0};
  if (!mm_qp_factor(&kkt, a, m4, one)) {
    q->x0 = a - a + (h0 - h0); q->x2 = q->x0; q->gap = q->x0;  /* NaN stays NaN */
    return 0;
  }
  double x0 = -0.0, x2 = -0.0, *s = q->s, *z = q->z;
  z[0] = q->h[0]; z[1] = q->h[1]; z[2] = q->h[2]; z[3] = q->h[3];
  mm_qp_solve(&kkt, m4, &x0, &x2, z);
  s[0] = -z[0]; s[1] = -z[1]; s[2] = -z[2]; s[3] = -z[3];
  double nrm = sqrt(mm_qp_dot(s, s, m4));
  const double ts = mm_qp_maxneg(s, m4);
  if (ts >= -1e-8 * fmax(nrm, 1.0)) { const double sh = 1.0 + ts; s[0] = s[0] + sh; s[1] = s[1] + sh; s[2] = s[2] + sh; s[3] = s[3] + sh; }
  nrm = sqrt(mm_qp_dot(z, z, m4));
  const double tz = mm_qp_maxneg(z, m4);
  if (tz >= -1e-8 * fmax(nrm, 1.0)) { const double sh = 1.0 + tz; z[0] = z[0] + sh; z[1] = z[1] + sh; z[2] = z[2] + sh; z[3] = z[3] + sh; }
  q->x0 = x0; q->x2 = x2;
  q->gap = mm_qp_dot(s, z, m4);
  for (int k = 0; k < 4; k++) { q->d[k] = 1.0; q->di[k] = 1.0; q->lmbda[k] = 1.0; }
  return 1;
}

/* Residuals of the current iterate and cvxopt's stopping test.  Returns 0: go on (call mm_qp_bottom with *r), 1: stop,
 * "optimal", 2: stop, "unknown" (iteration cap).  m4 is passed so that a caller with a compile-time value gets the
 * specialised code. */
MMM_FN int mm_qp_top(const MMQpState *q, int m4, MMQpRes *r) {
  const double a = q->a, x0 = q->x0, x2 = q->x2, gap = q->gap;
  const double *s = q->s, *z = q->z, *h = q->h;
  /* rx = P x + G' z ; f0 = 1/2 x'Px ; rz = s + G x - h */
  double rx0 = 0.0 + 1.0 * x0, rx2 = 0.0 + 1e18 * x2;
  const double f0 = 0.5 * (((0.0 + x0 * rx0) + x2 * rx2) + 0.0);
  double t = 0.0 + a * z[0];
  t = t + 1.0 * z[1];
  t = t + (-1.0) * z[2];
  if (m4) t = t + a * z[3];
  rx0 = rx0 + t;
  t = 0.0 + (-1.0) * z[0];
  if (m4) t = t + (-1.0) * z[3];
  rx2 = rx2 + t;
  const double resx = sqrt((0.0 + rx0 * rx0) + rx2 * rx2);
  double *rz = r->rz;
  rz[0] = ((s[0] - h[0]) + x0 * a) + x2 * (-1.0);
  rz[1] = (s[1] - h[1]) + x0 * 1.0;
  rz[2] = (s[2] - h[2]) + x0 * (-1.0);
  rz[3] = m4 ? ((s[3] - h[3]) + x0 * a) + x2 * (-1.0) : 0.0;
  r->rx0 = rx0; r->rx2 = rx2;
  const double resz = sqrt(mm_qp_dot(rz, rz, m4));
  const double pcost = f0, dcost = f0 + mm_qp_dot(z, rz, m4) - gap;
  int have_rel = 0;
  double relgap = 0.0;
  if (pcost < 0.0) { relgap = gap / -pcost; have_rel = 1; }
  else if (dcost > 0.0) { relgap = gap / dcost; have_rel = 1; }
  const double pres = resz / q->resz0, dres = resx / 1.0;
  const int met = pres <= MM_QP_FEASTOL && dres <= MM_QP_FEASTOL && (gap <= MM_QP_ABSTOL || (have_rel && relgap <= MM_QP_RELTOL));
  if (q->iters == MM_QP_MAXITERS) return 2;  /* coneqp: the cap wins over a test met in the same iteration */
  return met ? 1 : 0;
}

/* One interior-point iteration from the residuals of mm_qp_top.  Returns 0 when the KKT matrix is singular ("Terminated
 * (singular KKT matrix)": the iterate stands, status "unknown"), else 1. */
MMM_FN int mm_qp_bottom(MMQpState *q, int m4, const MMQpRes *r) {
  const double a = q->a, mm = m4 ? 4.0 : 3.0, gap = q->gap;
  double *s = q->s, *z = q->z, *d = q->d, *di = q->di, *lmbda = q->lmbda;
  const double *rz = r->rz;
  double lmbdasq[4], dz[4], ds[4], ws3[4] = {0.0, 0.0, 0.0, 0.0}, rl[4];
  MMQpKkt kkt;
  double t;
  if (q->iters == 0) {  /* misc.compute_scaling */
    for (int k = 0; k < 4; k++)
      if (k < 3 || m4) { d[k] = sqrt(s[k] / z[k]); di[k] = 1.0 / d[k]; lmbda[k] = sqrt(s[k] * z[k]); }
      else { d[k] = 1.0; di[k] = 1.0; lmbda[k] = 1.0; }
  }
  for (int k = 0; k < 4; k++) { lmbdasq[k] = lmbda[k] * lmbda[k]; rl[k] = MM_QP_RCP(lmbda[k]); }
  (void)rl;
  if (!mm_qp_factor(&kkt, a, m4, di)) return 0;
  const double mu = gap / mm;
  double sigma = 0.0, step = 1.0, dx0 = 0.0, dx2 = 0.0;
  for (int i = 0; i < 2; i++) {
    for (int k = 0; k < 4; k++) {
      t = 0.0;
      if (i == 1) t = t - ws3[k];
      t = t - lmbdasq[k];
      ds[k] = t + sigma * mu;
    }
    dx0 = -r->rx0; dx2 = -r->rx2;
    for (int k = 0; k < 4; k++) { dz[k] = -rz[k]; ds[k] = MM_QP_DIVR(ds[k], lmbda[k], rl[k]); dz[k] = dz[k] - d[k] * ds[k]; }
    mm_qp_solve(&kkt, m4, &dx0, &dx2, dz);
    for (int k = 0; k < 4; k++) ds[k] = ds[k] - dz[k];
    const double dsdz = mm_qp_dot(ds, dz, m4);
    if (i == 0) for (int k = 0; k < 4; k++) ws3[k] = ds[k] * dz[k];
    for (int k = 0; k < 4; k++) { ds[k] = MM_QP_DIVR(ds[k], lmbda[k], rl[k]); dz[k] = MM_QP_DIVR(dz[k], lmbda[k], rl[k]); }
    const double ts = mm_qp_maxneg(ds, m4), tz = mm_qp_maxneg(dz, m4);
    const double tm = fmax(0.0, fmax(ts, tz));
    if (tm == 0) step = 1.0;
    else if (i == 0) step = fmin(1.0, 1.0 / tm);
    else step = fmin(1.0, MM_QP_STEP / tm);
    if (i == 0) {
      const double sg = fmin(1.0, fmax(0.0, 1.0 - step + dsdz / gap * (step * step)));
      sigma = sg * sg * sg;
    }
  }
  q->x0 = q->x0 + step * dx0;
  q->x2 = q->x2 + step * dx2;
  for (int k = 0; k < 4; k++) {  /* updated iterates in the current scaling, then misc.update_scaling */
    if (k == 3 && !m4) continue;
    ds[k] = (step * ds[k] + 1.0) * lmbda[k];
    dz[k] = (step * dz[k] + 1.0) * lmbda[k];
    ds[k] = sqrt(ds[k]);
    dz[k] = sqrt(dz[k]);
    d[k] = d[k] * ds[k] / dz[k];
    di[k] = 1.0 / d[k];
    lmbda[k] = ds[k] * dz[k];
    s[k] = d[k] * lmbda[k];
    z[k] = di[k] * lmbda[k];
  }
  q->gap = mm_qp_dot(lmbda, lmbda, m4);
  q->iters = q->iters + 1;
  return 1;
}

/* One QP from start to stop.  Returns 1 ("optimal") or 0 ("unknown": iteration cap or singular KKT matrix).
 * d = u[0], s = u[2]. */
MMM_FN int mm_qp_ipm_cbf(double a, double h0, double h1, double h2, double h3, int rows, double *d_out, double *s_out,
                         int *iters_out) {
  MMQpState q;
  MMQpRes r;
  int status = 0;
  if (mm_qp_start(&q, a, h0, h1, h2, h3, rows)) {
    for (;;) {
      const int stop = mm_qp_top(&q, q.m4, &r);
      if (stop) { status = stop == 1; break; }
      if (!mm_qp_bottom(&q, q.m4, &r)) break;
    }
  }
  *d_out = q.x0; *s_out = q.x2; *iters_out = q.iters;
  return status;
}

#endif /* MM_QP_H */
