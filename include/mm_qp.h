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
 * Written with scalar fields and straight-line code on purpose (no arrays, no loops): on the device every value
 * must live in a register -- an array that is indexed by a loop variable or reached through a pointer ends up in
 * scratch memory, and a scratch round trip inside the iteration costs more than the iteration.  The fourth row
 * exists only with rows == 4 (MASS with constrain_adj, a few per cent of the QPs): everything that touches it sits
 * in `if (m4)` blocks, which a wave skips when none of its lanes has one.
 *
 * Divisions by the same denominator inside an iteration (the scaling entries lambda_k: six each; the Cholesky
 * pivots: five and four) go through MM_QP_RCP / MM_QP_DIVR, square roots through MM_QP_SQRT: by default plain IEEE
 * operations; a device build may define cheaper forms with the same result bits (marl-mass_amd/csrc/mm_device.h).
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
/* dres = sqrt(v) <= MM_QP_FEASTOL  <=>  v <= MM_QP_FEASTOL_SQ: the largest double whose correctly rounded square root is
 * <= 1e-7 (sqrt is monotone; found by stepping ulps around 1e-14, tools/gen_math_consts.py).  Saves the square root. */
#define MM_QP_FEASTOL_SQ 0x1.6849b86a12b9bp-47

/* a / b for several a with one b: R = MM_QP_RCP(b) once, then MM_QP_DIVR(a, b, R) == a / b bit for bit */
#ifndef MM_QP_RCP
#define MM_QP_RCP(b) (b)
#define MM_QP_DIVR(a, b, R) ((a) / (b))
#endif
#ifndef MM_QP_DIV
#define MM_QP_DIV(a, b) ((a) / (b))
#endif
#ifndef MM_QP_SQRT
#define MM_QP_SQRT(x) sqrt(x)
#endif

typedef struct MMQpKkt {  /* misc.kkt_chol2 factor: S = P + Gs'Gs = L L', Gs = W^-1 G */
  double g0, g3;          /* Gs[0][0], Gs[3][0] = di * a  (Gs[k][2] = -di[k] for k = 0, 3; Gs[1][0] = di1, Gs[2][0] = -di2) */
  double l00, l20, l22;   /* L (L11 = 1, L10 = L21 = 0) */
  double r00, r22;        /* MM_QP_RCP of the pivots l00, l22 */
  double di0, di1, di2, di3;
} MMQpKkt;

typedef struct MMQpState {
  double a, h0, h1, h2, h3, resz0;  /* the problem; resz0 = max(1, |h|) */
  double x0, x2, gap;               /* primal iterate (d, slack) and s'z */
  double s0, s1, s2, s3, z0, z1, z2, z3;          /* slacks and multipliers */
  double d0, d1, d2, d3, di0, di1, di2, di3;      /* Nesterov-Todd scaling W = diag(d) and W^-1 ... */
  double l0, l1, l2, l3;                          /* ... lambda = W^-1 s = W z (valid from the first mm_qp_bottom on) */
  double ax0, ax2;                                /* what the last iteration added to x0 / x2 (step * dx; mm_qp_frozen reads it) */
  int m4, iters;
} MMQpState;
typedef struct MMQpRes { double rx0, rx2, rz0, rz1, rz2, rz3; } MMQpRes;  /* what mm_qp_top hands to mm_qp_bottom */

MMM_FN int mm_qp_factor(MMQpKkt *f, double a, int m4, double di0, double di1, double di2, double di3) {
  f->di0 = di0; f->di1 = di1; f->di2 = di2; f->di3 = 0.0;
  f->g0 = di0 * a;
  f->g3 = 0.0;
  double s00 = f->g0 * f->g0 + di1 * di1 + di2 * di2;
  double s20 = 0.0 + (-di0) * f->g0;
  double s22 = di0 * di0;
  if (m4) {
    f->di3 = di3;
    f->g3 = di3 * a;
    s00 = s00 + f->g3 * f->g3;
    s20 = s20 + (-di3) * f->g3;
    s22 = s22 + di3 * di3;
  }
  s00 = s00 + 1.0;
  s22 = s22 + 1e18;
  if (!(s00 > 0.0)) return 0;
  f->l00 = MM_QP_SQRT(s00);
  f->r00 = MM_QP_RCP(f->l00);
  f->l20 = MM_QP_DIVR(s20, f->l00, f->r00);
  const double t = s22 - f->l20 * f->l20;
  if (!(t > 0.0)) return 0;
  f->l22 = MM_QP_SQRT(t);
  f->r22 = MM_QP_RCP(f->l22);
  return 1;
}
/* [P G'; G -W'W][ux; W^-1 uz] = [bx; bz]; (x0, x2, z) hold the right-hand side on entry, the solution on exit */
MMM_FN void mm_qp_solve(const MMQpKkt *f, int m4, double *x0, double *x2, double *z0, double *z1, double *z2, double *z3) {
  double w0 = *z0 * f->di0, w1 = *z1 * f->di1, w2 = *z2 * f->di2, w3 = 0.0;
  double t = 0.0 + f->g0 * w0;
  t = t + f->di1 * w1;
  t = t + (-f->di2) * w2;
  double u = 0.0 + (-f->di0) * w0;
  if (m4) {
    w3 = *z3 * f->di3;
    t = t + f->g3 * w3;
    u = u + (-f->di3) * w3;
  }
  double a0 = *x0 + t;
  double a2 = *x2 + u;
  a0 = MM_QP_DIVR(a0, f->l00, f->r00);          /* trsv: L x = x */
  a2 = a2 - a0 * f->l20;
  a2 = MM_QP_DIVR(a2, f->l22, f->r22);
  a2 = MM_QP_DIVR(a2, f->l22, f->r22);          /* trsv 'T': L' x = x */
  t = a0 - f->l20 * a2;
  a0 = MM_QP_DIVR(t, f->l00, f->r00);
  t = -w0; t = t + a0 * f->g0; t = t + a2 * (-f->di0); *z0 = t;
  t = -w1; t = t + a0 * f->di1; *z1 = t;
  t = -w2; t = t + a0 * (-f->di2); *z2 = t;
  if (m4) { t = -w3; t = t + a0 * f->g3; t = t + a2 * (-f->di3); *z3 = t; }
  *x0 = a0; *x2 = a2;
}
/* sum_k p_k q_k in index order, one rounding per operation */
MMM_FN double mm_qp_dot(double p0, double q0, double p1, double q1, double p2, double q2, double p3, double q3, int m4) {
  double t = 0.0 + p0 * q0;
  t = t + p1 * q1;
  t = t + p2 * q2;
  if (m4) t = t + p3 * q3;
  return t;
}
MMM_FN double mm_qp_maxneg(double p0, double p1, double p2, double p3, int m4) {  /* misc.max_step for the 'l' cone */
  double t = -p0;
  if (-p1 > t) t = -p1;
  if (-p2 > t) t = -p2;
  if (m4 && -p3 > t) t = -p3;
  return t;
}

/* The initial point (coneqp: solve the KKT system with W = I, shift s and z into the cone).  Returns 0 when cvxopt would
 * raise ValueError("Rank(A) < p or Rank([P; A; G]) < n") -- only a NaN input does that here; q->x0 is then NaN. */
MMM_FN int mm_qp_start(MMQpState *q, double a, double h0, double h1, double h2, double h3, int rows) {
  const int m4 = rows == 4;
  q->a = a; q->m4 = m4; q->iters = 0;
  q->ax0 = q->ax2 = INFINITY;
  q->h0 = h0; q->h1 = h1; q->h2 = h2; q->h3 = m4 ? h3 : 0.0;
  q->resz0 = fmax(1.0, sqrt(mm_qp_dot(q->h0, q->h0, q->h1, q->h1, q->h2, q->h2, q->h3, q->h3, m4)));  /* resx0 = max(1, |q|) = 1 */
  q->d0 = q->d1 = q->d2 = q->d3 = 1.0; q->di0 = q->di1 = q->di2 = q->di3 = 1.0; q->l0 = q->l1 = q->l2 = q->l3 = 1.0;
  MMQpKkt kkt;
  if (!mm_qp_factor(&kkt, a, m4, 1.0, 1.0, 1.0, 1.0)) {
    q->x0 = a - a + (h0 - h0); q->x2 = q->x0; q->gap = q->x0;  /* NaN stays NaN */
    q->s0 = q->s1 = q->s2 = q->s3 = q->z0 = q->z1 = q->z2 = q->z3 = q->x0;
    return 0;
  }
  double x0 = -0.0, x2 = -0.0;
  double z0 = q->h0, z1 = q->h1, z2 = q->h2, z3 = q->h3;
  mm_qp_solve(&kkt, m4, &x0, &x2, &z0, &z1, &z2, &z3);
  double s0 = -z0, s1 = -z1, s2 = -z2, s3 = -z3;
  /* "if ts >= -1e-8 * max(nrm, 1.0)": the right-hand side is <= -1e-8 whatever the norm is, so ts >= 0 (some s_k <= 0, the
   * common case) settles the test without the norm and its square root */
  const double ts = mm_qp_maxneg(s0, s1, s2, s3, m4);
  int shift = ts >= 0.0;
  if (!shift) shift = ts >= -1e-8 * fmax(sqrt(mm_qp_dot(s0, s0, s1, s1, s2, s2, s3, s3, m4)), 1.0);
  if (shift) { const double sh = 1.0 + ts; s0 = s0 + sh; s1 = s1 + sh; s2 = s2 + sh; s3 = s3 + sh; }
  const double tz = mm_qp_maxneg(z0, z1, z2, z3, m4);
  shift = tz >= 0.0;
  if (!shift) shift = tz >= -1e-8 * fmax(sqrt(mm_qp_dot(z0, z0, z1, z1, z2, z2, z3, z3, m4)), 1.0);
  if (shift) { const double sh = 1.0 + tz; z0 = z0 + sh; z1 = z1 + sh; z2 = z2 + sh; z3 = z3 + sh; }
  q->x0 = x0; q->x2 = x2;
  q->s0 = s0; q->s1 = s1; q->s2 = s2; q->s3 = s3; q->z0 = z0; q->z1 = z1; q->z2 = z2; q->z3 = z3;
  q->gap = mm_qp_dot(s0, z0, s1, z1, s2, z2, s3, z3, m4);
  return 1;
}

/*
 * Frozen-iterate certificate (MM_QP_CERTIFY builds only: the HIP kernels; the oracle and the reference-side restatement
 * always run the literal loop, so every HIP-vs-oracle parity test checks the certificate bit for bit).
 *
 * 0.4 % of the shield's QPs are infeasible without the slack (the CBF row cannot be met even at full braking): the solution is
 * d = -h2 (the lower bound) with slack s > 0, the multiplier of the CBF row is ~1e18 * s, and cvxopt's dual-residual test
 * ||P x + G'z|| <= 1e-7 can never pass on a cancellation of 2^47-sized terms -- coneqp runs to its 100-iteration cap and
 * returns status "unknown" (cbf.py:134-140 keeps the iterate).  From iteration ~28 on those 70+ iterations change nothing that
 * is returned: x0 sits EXACTLY on -h2, x2 is constant, and what still moves (s, z of the active rows shrinking / wiggling in
 * their last bits, the scaling) feeds back into x only through Newton steps that are absorbed by the rounding of x + step*dx.
 * mm_qp_frozen() recognises that state from quantities the iteration has anyway; a caller may then stop and report
 * (x0, x2, "unknown", 100 iterations).  It returns 1 only if ALL of the following hold, each chosen so that it keeps holding:
 *   (1) the last update of x0 and of x2 was absorbed with 20 bits to spare: |step * dx| <= 2^-75 |x| (half an ulp is
 *       >= 2^-54 |x|).  In this regime dx0 = -dz2 + O((|rx| + 1e18 a |dz0|) / di2^2) with dz_k = s_k (1 + o(1)) of the ACTIVE rows:
 *       it shrinks with s_k (x ~0.01 per iteration, never grows by more than the centring term, see (3)), and di2^2 >= 2^100.
 *   (2) row 2 (-d <= h2) is active and met with equality in floating point: x0 == -h2, its residual rz2 is exactly 0 and
 *       s2 <= 2^-70 |h2| (so fl(s2 - h2) = -h2 whatever s2 does next); row 1 is inactive (z1 <= 2^-60); each CBF row (0, and 3
 *       with rows == 4) is either active the same way (rz == 0, s tiny, z >= 2^20) or inactive (z <= 2^-60), at least one active.
 *       Exactly-zero residuals of the active rows mean nothing pulls x away: rz is a function of (h, x, a) alone once s is
 *       below half an ulp of h.
 *   (3) no underflow / overflow before the cap: s_active >= 2^-400 and d_active >= 2^-120 now; both lose <= 2^-6.7 / 2^-3.4
 *       per iteration for at most 92 more iterations (s' = s (1 - step) + step * sigma * mu / z with step <= 1, sigma <= 1).
 *   (4) the stopping test cannot pass before the cap.  Its dual part needs |rx0| <= 1e-7 with rx0 = fl(x0 + t),
 *       t = ((a z0 + z1) - z2) [+ a z3]: the multipliers of the active rows are >= 2^20, so t is a sum of multiples of
 *       g = ulp of the smallest of them (tiny terms -- inactive multipliers, <= 2^-60 -- are absorbed or add < 2^-59), i.e.
 *       t lies on the lattice g Z up to 2^-59, and |rx0| >= dist(x0, g Z) - 2^-58.  Required: dist(x0, (g/2) Z) > 2e-7 (g/2:
 *       the multipliers wiggle in their last bits and may cross a binade).  A QP that fails (4) (x0 within 2e-7 of the
 *       lattice: 6e-6 of them at the usual g = 2^-5) simply iterates on.
 * Verified bit for bit against the literal loop on every QP the reference assembled (tests/golden: 78 388, 364 capped, all
 * certified between iteration 23 and 51) and on the oracle in every rollout / soak test.
 */
#ifndef MM_QP_ILOGB
#define MM_QP_ILOGB(x) ilogb(x)
#define MM_QP_LDEXP(x, e) ldexp(x, e)
#define MM_QP_RINT(x) rint(x)
#endif
MMM_FN int mm_qp_row_state(double s, double z, double rz, double hmag) {  /* 1 active, 2 inactive, 0 neither */
  if (rz == 0.0 && s <= 0x1p-70 * hmag && s >= 0x1p-400 && z >= 0x1p+20) return 1;
  if (z <= 0x1p-60) return 2;
  return 0;
}
MMM_FN int mm_qp_frozen(const MMQpState *q, const MMQpRes *r) {
  if (q->iters < 16 || !(q->a > 0.0)) return 0;  /* (never met earlier: first at 17 in 1.5e8 capped QPs, tools/qp_certificate_fuzz.c; a QP that
                                                     converges is through by iteration 11 and should not pay for the conditions below) */
  if (!(fabs(q->ax0) <= 0x1p-75 * fabs(q->x0)) || !(fabs(q->ax2) <= 0x1p-75 * fabs(q->x2))) return 0;             /* (1) */
  if (q->x0 != -q->h2 || mm_qp_row_state(q->s2, q->z2, r->rz2, fabs(q->h2)) != 1 || !(q->z1 <= 0x1p-60)) return 0; /* (2) */
  const int r0 = mm_qp_row_state(q->s0, q->z0, r->rz0, fmax(fabs(q->h0), fabs(q->x2)));
  const int r3 = q->m4 ? mm_qp_row_state(q->s3, q->z3, r->rz3, fmax(fabs(q->h3), fabs(q->x2))) : 2;
  if (r0 == 0 || r3 == 0 || (r0 != 1 && r3 != 1)) return 0;
  if (!(q->d2 >= 0x1p-120) || (r0 == 1 && !(q->d0 >= 0x1p-120)) || (r3 == 1 && !(q->d3 >= 0x1p-120))) return 0;    /* (3) */
  int e = MM_QP_ILOGB(q->z2);                                                                                        /* (4) */
  if (r0 == 1) { const int e0 = MM_QP_ILOGB(q->a * q->z0); e = e0 < e ? e0 : e; }
  if (r3 == 1) { const int e3 = MM_QP_ILOGB(q->a * q->z3); e = e3 < e ? e3 : e; }
  if (e < 20 || e > 200) return 0;
  const double y = MM_QP_LDEXP(q->x0, 53 - e);  /* x0 / (g/2), exact (g/2 = 2^(e-53)) */
  if (!(fabs(y) < 0x1p+51)) return 0;           /* the lattice is finer than x0 itself */
  const double dist = MM_QP_LDEXP(fabs(y - MM_QP_RINT(y)), e - 53);
  return dist > 2e-7;
}

/* Residuals of the current iterate and cvxopt's stopping test.  Returns 0: go on (call mm_qp_bottom with *r), 1: stop,
 * "optimal", 2: stop, "unknown" (iteration cap); MM_QP_CERTIFY builds also 3: stop, the iterate is frozen and the loop
 * would run to the cap without changing it (mm_qp_frozen): report "unknown" and MM_QP_MAXITERS iterations. */
MMM_FN int mm_qp_top(const MMQpState *q, MMQpRes *r) {
  const int m4 = q->m4;
  const double a = q->a, x0 = q->x0, x2 = q->x2, gap = q->gap;
  /* rx = P x + G' z ; f0 = 1/2 x'Px ; rz = s + G x - h */
  double rx0 = 0.0 + 1.0 * x0, rx2 = 0.0 + 1e18 * x2;
  const double f0 = 0.5 * (((0.0 + x0 * rx0) + x2 * rx2) + 0.0);
  double t = 0.0 + a * q->z0;
  t = t + 1.0 * q->z1;
  t = t + (-1.0) * q->z2;
  double u = 0.0 + (-1.0) * q->z0;
  const double rz0 = ((q->s0 - q->h0) + x0 * a) + x2 * (-1.0);
  const double rz1 = (q->s1 - q->h1) + x0 * 1.0;
  const double rz2 = (q->s2 - q->h2) + x0 * (-1.0);
  double rz3 = 0.0;
  if (m4) {
    t = t + a * q->z3;
    u = u + (-1.0) * q->z3;
    rz3 = ((q->s3 - q->h3) + x0 * a) + x2 * (-1.0);
  }
  rx0 = rx0 + t;
  rx2 = rx2 + u;
  const int dres_ok = ((0.0 + rx0 * rx0) + rx2 * rx2) <= MM_QP_FEASTOL_SQ;  /* dres = resx / resx0 = sqrt(.) / 1 <= feastol */
  r->rx0 = rx0; r->rx2 = rx2; r->rz0 = rz0; r->rz1 = rz1; r->rz2 = rz2; r->rz3 = rz3;
  if (q->iters == MM_QP_MAXITERS) return 2;  /* coneqp: the cap wins over a test met in the same iteration */
  /* met = pres <= feastol and dres <= feastol and (gap <= abstol or (relgap is not None and relgap <= reltol)), evaluated
   * cheapest condition first: the dual residual settles last, so most iterations need neither the norm of rz (a square
   * root), pres, nor the relative gap (two divisions) */
  int met = 0;
  if (dres_ok) {
    const double resz = sqrt(mm_qp_dot(rz0, rz0, rz1, rz1, rz2, rz2, rz3, rz3, m4));
    const double pres = resz / q->resz0;
    if (pres <= MM_QP_FEASTOL) {
      if (gap <= MM_QP_ABSTOL) met = 1;
      else {
        const double pcost = f0, dcost = f0 + mm_qp_dot(q->z0, rz0, q->z1, rz1, q->z2, rz2, q->z3, rz3, m4) - gap;
        if (pcost < 0.0) met = gap / -pcost <= MM_QP_RELTOL;
        else if (dcost > 0.0) met = gap / dcost <= MM_QP_RELTOL;
      }
    }
  }
#ifdef MM_QP_CERTIFY
  if (!met && mm_qp_frozen(q, r)) return 3;
#endif
  return met ? 1 : 0;
}

/* the predictor (I = 0) or corrector (I = 1) solve of one iteration: right-hand side, KKT solve, scaled steps, max step */
#define MM_QP_PASS(I)                                                                                              \
  {                                                                                                                \
    double t0_ = 0.0, t1_ = 0.0, t2_ = 0.0, t3_ = 0.0;                                                             \
    if (I == 1) { t0_ = t0_ - ws0; t1_ = t1_ - ws1; t2_ = t2_ - ws2; }                                             \
    t0_ = t0_ - lsq0; t1_ = t1_ - lsq1; t2_ = t2_ - lsq2;                                                          \
    ds0 = t0_ + sigma * mu; ds1 = t1_ + sigma * mu; ds2 = t2_ + sigma * mu;                                        \
    dx0 = -r->rx0; dx2 = -r->rx2;                                                                                  \
    ds0 = MM_QP_DIVR(ds0, q->l0, rl0); ds1 = MM_QP_DIVR(ds1, q->l1, rl1); ds2 = MM_QP_DIVR(ds2, q->l2, rl2);       \
    dz0 = -r->rz0 - q->d0 * ds0; dz1 = -r->rz1 - q->d1 * ds1; dz2 = -r->rz2 - q->d2 * ds2;                         \
    if (m4) {                                                                                                      \
      if (I == 1) t3_ = t3_ - ws3;                                                                                 \
      t3_ = t3_ - lsq3;                                                                                            \
      ds3 = t3_ + sigma * mu;                                                                                      \
      ds3 = MM_QP_DIVR(ds3, q->l3, rl3);                                                                           \
      dz3 = -r->rz3 - q->d3 * ds3;                                                                                 \
    }                                                                                                              \
    mm_qp_solve(&kkt, m4, &dx0, &dx2, &dz0, &dz1, &dz2, &dz3);                                                     \
    ds0 = ds0 - dz0; ds1 = ds1 - dz1; ds2 = ds2 - dz2;                                                             \
    if (m4) ds3 = ds3 - dz3;                                                                                       \
    dsdz = mm_qp_dot(ds0, dz0, ds1, dz1, ds2, dz2, ds3, dz3, m4);                                                  \
    if (I == 0) { ws0 = ds0 * dz0; ws1 = ds1 * dz1; ws2 = ds2 * dz2; if (m4) ws3 = ds3 * dz3; }                    \
    ds0 = MM_QP_DIVR(ds0, q->l0, rl0); dz0 = MM_QP_DIVR(dz0, q->l0, rl0);                                          \
    ds1 = MM_QP_DIVR(ds1, q->l1, rl1); dz1 = MM_QP_DIVR(dz1, q->l1, rl1);                                          \
    ds2 = MM_QP_DIVR(ds2, q->l2, rl2); dz2 = MM_QP_DIVR(dz2, q->l2, rl2);                                          \
    if (m4) { ds3 = MM_QP_DIVR(ds3, q->l3, rl3); dz3 = MM_QP_DIVR(dz3, q->l3, rl3); }                              \
    const double ts_ = mm_qp_maxneg(ds0, ds1, ds2, ds3, m4), tz_ = mm_qp_maxneg(dz0, dz1, dz2, dz3, m4);           \
    const double tm_ = fmax(0.0, fmax(ts_, tz_));                                                                  \
    if (tm_ == 0) step = 1.0;                                                                                      \
    else if (I == 0) step = fmin(1.0, MM_QP_DIV(1.0, tm_));                                                        \
    else step = fmin(1.0, MM_QP_DIV(MM_QP_STEP, tm_));                                                             \
  }
/* misc.update_scaling for row K: the updated iterates in the current scaling, then the new scaling */
#define MM_QP_UPDATE(K)                                                                                            \
  {                                                                                                                \
    double us_ = (step * ds##K + 1.0) * q->l##K, uz_ = (step * dz##K + 1.0) * q->l##K;                            \
    us_ = MM_QP_SQRT(us_);                                                                                         \
    uz_ = MM_QP_SQRT(uz_);                                                                                         \
    q->d##K = MM_QP_DIV(q->d##K * us_, uz_);                                                                       \
    q->di##K = MM_QP_DIV(1.0, q->d##K);                                                                            \
    q->l##K = us_ * uz_;                                                                                           \
    q->s##K = q->d##K * q->l##K;                                                                                   \
    q->z##K = q->di##K * q->l##K;                                                                                  \
  }
/* misc.compute_scaling for row K (first iteration) */
#define MM_QP_SCALE0(K)                                                                                            \
  { q->d##K = MM_QP_SQRT(MM_QP_DIV(q->s##K, q->z##K)); q->di##K = MM_QP_DIV(1.0, q->d##K); q->l##K = MM_QP_SQRT(q->s##K * q->z##K); }

/* One interior-point iteration from the residuals of mm_qp_top.  Returns 0 when the KKT matrix is singular ("Terminated
 * (singular KKT matrix)": the iterate stands, status "unknown"), else 1. */
MMM_FN int mm_qp_bottom(MMQpState *q, const MMQpRes *r) {
  const int m4 = q->m4;
  const double a = q->a, mm = m4 ? 4.0 : 3.0, gap = q->gap;
  if (q->iters == 0) {
    MM_QP_SCALE0(0) MM_QP_SCALE0(1) MM_QP_SCALE0(2)
    if (m4) MM_QP_SCALE0(3)
  }
  const double lsq0 = q->l0 * q->l0, lsq1 = q->l1 * q->l1, lsq2 = q->l2 * q->l2;
  const double rl0 = MM_QP_RCP(q->l0), rl1 = MM_QP_RCP(q->l1), rl2 = MM_QP_RCP(q->l2);
  double lsq3 = 1.0, rl3 = 1.0;
  if (m4) { lsq3 = q->l3 * q->l3; rl3 = MM_QP_RCP(q->l3); }
  (void)rl0; (void)rl1; (void)rl2; (void)rl3;
  MMQpKkt kkt;
  if (!mm_qp_factor(&kkt, a, m4, q->di0, q->di1, q->di2, q->di3)) return 0;
  const double mu = gap / mm;
  double sigma = 0.0, step = 1.0, dx0 = 0.0, dx2 = 0.0, dsdz = 0.0;
  double ds0 = 0.0, ds1 = 0.0, ds2 = 0.0, ds3 = 0.0, dz0 = 0.0, dz1 = 0.0, dz2 = 0.0, dz3 = 0.0;
  double ws0 = 0.0, ws1 = 0.0, ws2 = 0.0, ws3 = 0.0;
  MM_QP_PASS(0)
  {
    const double sg = fmin(1.0, fmax(0.0, 1.0 - step + MM_QP_DIV(dsdz, gap) * (step * step)));
    sigma = sg * sg * sg;
  }
  MM_QP_PASS(1)
  q->ax0 = step * dx0;
  q->ax2 = step * dx2;
  q->x0 = q->x0 + q->ax0;
  q->x2 = q->x2 + q->ax2;
  MM_QP_UPDATE(0) MM_QP_UPDATE(1) MM_QP_UPDATE(2)
  if (m4) MM_QP_UPDATE(3)
  q->gap = mm_qp_dot(q->l0, q->l0, q->l1, q->l1, q->l2, q->l2, q->l3, q->l3, m4);
  q->iters = q->iters + 1;
  return 1;
}

/* One QP from start to stop.  Returns 1 ("optimal") or 0 ("unknown": iteration cap or singular KKT matrix).
 * d = u[0], s = u[2]. */
MMM_FN int mm_qp_ipm_cbf(double a, double h0, double h1, double h2, double h3, int rows, double *d_out, double *s_out,
                         int *iters_out) {
  MMQpState q;
  MMQpRes r;
  int status = 0, certified = 0;
  (void)certified;
  if (mm_qp_start(&q, a, h0, h1, h2, h3, rows)) {
    for (;;) {
      const int stop = mm_qp_top(&q, &r);
      if (stop) { status = stop == 1; certified = stop == 3; break; }
      if (!mm_qp_bottom(&q, &r)) break;
    }
  }
  *d_out = q.x0; *s_out = q.x2; *iters_out = q.iters;
#ifdef MM_QP_CERTIFY
  if (certified) *iters_out = MM_QP_MAXITERS;
#endif
  return status;
}

#endif /* MM_QP_H */
