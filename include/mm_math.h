/*
 * mm_math.h -- bit-reproducible fp64 elementary functions for the merge env.
 *
 * Why: the shield's lane-change test (cbf.py:324-339) evaluates a quantity that is structurally
 * ~0 whenever the adjacent-vehicle CBF row is the active QP constraint, so its sign is decided by
 * rounding noise.  For the HIP path and its CPU checker to agree on such decisions they must
 * agree on every bit, which needs identical sin/cos/tan/atan/asin/exp/log.  These versions use
 * only IEEE-exact primitives (+ - * / sqrt fma ldexp frexp rint) in a fixed order, so gcc on
 * x86-64 (-mfma, -ffp-contract=off) and hipcc on gfx950 (-ffp-contract=off) give identical
 * bits.  Accuracy ~1 ulp (2 ulp for tan/asin); range-reduced for the arguments this path
 * produces (angles of a few radians, exp of [-1e3, 0], log of positive normal numbers).
 *
 * Constants are derived by tools/gen_math_consts.py (80-digit arithmetic): split pi/2 and ln 2,
 * atan(j/4), and plain Taylor coefficients (the reduced intervals are small enough that Taylor
 * truncation stays below 2^-56).
 *
 * The includer defines MMM_FN (e.g. `static inline` or `__device__ __forceinline__`).
 */
#ifndef MM_MATH_H
#define MM_MATH_H

#ifndef MMM_FN
#define MMM_FN static inline
#endif

/* One Horner step p*z + C with a compile-time coefficient C.  On gfx950 it is spelled as the VOP3 instruction with the
 * coefficient in a scalar register pair: left to itself the compiler (ROCm 7.2, under the register pressure of the fused
 * step kernel) parks all ~70 Taylor coefficients in VGPRs for the whole kernel and emits `v_mov_b64 tmp, C; v_fmac_f64
 * tmp, p, z` per step -- two vector instructions and ~30 resident VGPRs for what is one fused multiply-add.  With "s" the
 * coefficient is two rematerialisable s_mov_b32 on the scalar unit.  Same IEEE operation, same bits as fma(). */
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__)
MMM_FN double mmm_fma_c(double p, double z, double c) {
  double r;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
  return r;
}
#else
#define mmm_fma_c(p, z, c) fma((p), (z), (c))
#endif

#define MMM_PIO2_1 0x1.921fb54400000p+0 /* 1.5707963267341256 */
#define MMM_PIO2_1T 0x1.0b4611a626331p-34 /* 6.0771005065061922e-11 */
#define MMM_PIO2_HI 0x1.921fb54442d18p+0 /* 1.5707963267948966 */
#define MMM_PIO2_LO 0x1.1a62633145c07p-54 /* 6.123233995736766e-17 */
#define MMM_TWO_OVER_PI 0x1.45f306dc9c883p-1 /* 0.63661977236758138 */
#define MMM_LN2_HI 0x1.62e42ff000000p-1 /* 0.69314718060195446 */
#define MMM_LN2_LO -0x1.718432a1b0e26p-35 /* -4.2009150726810846e-11 */
#define MMM_INV_LN2 0x1.71547652b82fep+0 /* 1.4426950408889634 */
#define MMM_SQRT_HALF 0x1.6a09e667f3bcdp-1 /* 0.70710678118654757 */
#define MMM_ATAN_HI_1 0x1.f5b75f92c80ddp-3 /* 0.24497866312686414 */
#define MMM_ATAN_LO_1 0x1.8ab6e3cf7afbdp-57 /* 1.0698755618734451e-17 */
#define MMM_ATAN_HI_2 0x1.dac670561bb4fp-2 /* 0.46364760900080609 */
#define MMM_ATAN_LO_2 0x1.a2b7f222f65e2p-56 /* 2.2698777452961687e-17 */
#define MMM_ATAN_HI_3 0x1.4978fa3269ee1p-1 /* 0.64350110879328437 */
#define MMM_ATAN_LO_3 0x1.2419a87f2a458p-56 /* 1.5834785051444286e-17 */
#define MMM_ATAN_HI_4 0x1.921fb54442d18p-1 /* 0.78539816339744828 */
#define MMM_ATAN_LO_4 0x1.1a62633145c07p-55 /* 3.061616997868383e-17 */
#define MMM_S1 -0x1.5555555555555p-3 /* -0.16666666666666666 */
#define MMM_S2 0x1.1111111111111p-7 /* 0.0083333333333333332 */
#define MMM_S3 -0x1.a01a01a01a01ap-13 /* -0.00019841269841269841 */
#define MMM_S4 0x1.71de3a556c734p-19 /* 2.7557319223985893e-06 */
#define MMM_S5 -0x1.ae64567f544e4p-26 /* -2.505210838544172e-08 */
#define MMM_S6 0x1.6124613a86d09p-33 /* 1.6059043836821613e-10 */
#define MMM_S7 -0x1.ae7f3e733b81fp-41 /* -7.6471637318198164e-13 */
#define MMM_C2 0x1.5555555555555p-5 /* 0.041666666666666664 */
#define MMM_C3 -0x1.6c16c16c16c17p-10 /* -0.0013888888888888889 */
#define MMM_C4 0x1.a01a01a01a01ap-16 /* 2.4801587301587302e-05 */
#define MMM_C5 -0x1.27e4fb7789f5cp-22 /* -2.7557319223985888e-07 */
#define MMM_C6 0x1.1eed8eff8d898p-29 /* 2.08767569878681e-09 */
#define MMM_C7 -0x1.93974a8c07c9dp-37 /* -1.1470745597729725e-11 */
#define MMM_C8 0x1.ae7f3e733b81fp-45 /* 4.7794773323873853e-14 */
#define MMM_A1 -0x1.5555555555555p-2 /* -0.33333333333333331 */
#define MMM_A2 0x1.999999999999ap-3 /* 0.20000000000000001 */
#define MMM_A3 -0x1.2492492492492p-3 /* -0.14285714285714285 */
#define MMM_A4 0x1.c71c71c71c71cp-4 /* 0.1111111111111111 */
#define MMM_A5 -0x1.745d1745d1746p-4 /* -0.090909090909090912 */
#define MMM_A6 0x1.3b13b13b13b14p-4 /* 0.076923076923076927 */
#define MMM_A7 -0x1.1111111111111p-4 /* -0.066666666666666666 */
#define MMM_A8 0x1.e1e1e1e1e1e1ep-5 /* 0.058823529411764705 */
#define MMM_A9 -0x1.af286bca1af28p-5 /* -0.052631578947368418 */
#define MMM_A10 0x1.8618618618618p-5 /* 0.047619047619047616 */
#define MMM_E2 0x1.0000000000000p-1 /* 0.5 */
#define MMM_E3 0x1.5555555555555p-3 /* 0.16666666666666666 */
#define MMM_E4 0x1.5555555555555p-5 /* 0.041666666666666664 */
#define MMM_E5 0x1.1111111111111p-7 /* 0.0083333333333333332 */
#define MMM_E6 0x1.6c16c16c16c17p-10 /* 0.0013888888888888889 */
#define MMM_E7 0x1.a01a01a01a01ap-13 /* 0.00019841269841269841 */
#define MMM_E8 0x1.a01a01a01a01ap-16 /* 2.4801587301587302e-05 */
#define MMM_E9 0x1.71de3a556c734p-19 /* 2.7557319223985893e-06 */
#define MMM_E10 0x1.27e4fb7789f5cp-22 /* 2.7557319223985888e-07 */
#define MMM_E11 0x1.ae64567f544e4p-26 /* 2.505210838544172e-08 */
#define MMM_E12 0x1.1eed8eff8d898p-29 /* 2.08767569878681e-09 */
#define MMM_E13 0x1.6124613a86d09p-33 /* 1.6059043836821613e-10 */
#define MMM_E14 0x1.93974a8c07c9dp-37 /* 1.1470745597729725e-11 */
#define MMM_L1 0x1.5555555555555p-2 /* 0.33333333333333331 */
#define MMM_L2 0x1.999999999999ap-3 /* 0.20000000000000001 */
#define MMM_L3 0x1.2492492492492p-3 /* 0.14285714285714285 */
#define MMM_L4 0x1.c71c71c71c71cp-4 /* 0.1111111111111111 */
#define MMM_L5 0x1.745d1745d1746p-4 /* 0.090909090909090912 */
#define MMM_L6 0x1.3b13b13b13b14p-4 /* 0.076923076923076927 */
#define MMM_L7 0x1.1111111111111p-4 /* 0.066666666666666666 */
#define MMM_L8 0x1.e1e1e1e1e1e1ep-5 /* 0.058823529411764705 */
#define MMM_L9 0x1.af286bca1af28p-5 /* 0.052631578947368418 */
#define MMM_L10 0x1.8618618618618p-5 /* 0.047619047619047616 */
#define MMM_L11 0x1.642c8590b2164p-5 /* 0.043478260869565216 */

/* Experiment (profiles/r04/variants.jsonl): Horner coefficients fetched from constant memory with scalar loads (one
 * s_load_dwordx16 per polynomial) instead of two s_mov_b32 per coefficient.  -DMMM_COEF_TABLES, device builds only. */
#if defined(MMM_COEF_TABLES) && defined(__HIP_DEVICE_COMPILE__)
#define MMM_TABLE(name, ...) __attribute__((weak)) __constant__ double name[] = {__VA_ARGS__}; /* (not const, not static: else folded back into literals) */
#define MMM_K(name, i, lit) (name[i])
#else
#define MMM_TABLE(name, ...)
#define MMM_K(name, i, lit) (lit)
#endif
MMM_TABLE(mmm_t_sin, MMM_S7, MMM_S6, MMM_S5, MMM_S4, MMM_S3, MMM_S2, MMM_S1)
MMM_TABLE(mmm_t_cos, MMM_C8, MMM_C7, MMM_C6, MMM_C5, MMM_C4, MMM_C3, MMM_C2)
MMM_TABLE(mmm_t_atan, MMM_A10, MMM_A9, MMM_A8, MMM_A7, MMM_A6, MMM_A5, MMM_A4, MMM_A3, MMM_A2, MMM_A1)
MMM_TABLE(mmm_t_exp, MMM_E14, MMM_E13, MMM_E12, MMM_E11, MMM_E10, MMM_E9, MMM_E8, MMM_E7, MMM_E6, MMM_E5, MMM_E4, MMM_E3, MMM_E2)
MMM_TABLE(mmm_t_log, MMM_L11, MMM_L10, MMM_L9, MMM_L8, MMM_L7, MMM_L6, MMM_L5, MMM_L4, MMM_L3, MMM_L2, MMM_L1)

/* sin / cos kernels on |r| <= pi/4 (Taylor to r^15 / r^16) */
MMM_FN double mmm_ksin(double r) {
  double z = r * r;
  double p = MMM_K(mmm_t_sin, 0, MMM_S7);
  p = mmm_fma_c(p, z, MMM_K(mmm_t_sin, 1, MMM_S6)); p = mmm_fma_c(p, z, MMM_K(mmm_t_sin, 2, MMM_S5)); p = mmm_fma_c(p, z, MMM_K(mmm_t_sin, 3, MMM_S4));
  p = mmm_fma_c(p, z, MMM_K(mmm_t_sin, 4, MMM_S3)); p = mmm_fma_c(p, z, MMM_K(mmm_t_sin, 5, MMM_S2)); p = mmm_fma_c(p, z, MMM_K(mmm_t_sin, 6, MMM_S1));
  return fma(r * z, p, r);
}
MMM_FN double mmm_kcos(double r) {
  double z = r * r;
  double q = MMM_K(mmm_t_cos, 0, MMM_C8);
  q = mmm_fma_c(q, z, MMM_K(mmm_t_cos, 1, MMM_C7)); q = mmm_fma_c(q, z, MMM_K(mmm_t_cos, 2, MMM_C6)); q = mmm_fma_c(q, z, MMM_K(mmm_t_cos, 3, MMM_C5));
  q = mmm_fma_c(q, z, MMM_K(mmm_t_cos, 4, MMM_C4)); q = mmm_fma_c(q, z, MMM_K(mmm_t_cos, 5, MMM_C3)); q = mmm_fma_c(q, z, MMM_K(mmm_t_cos, 6, MMM_C2));
  return fma(z * z, q, fma(-0.5, z, 1.0));
}
/* Cody-Waite reduction by pi/2 (two-part constant: exact for |k| < 2^19) */
MMM_FN double mmm_reduce(double x, int *quadrant) {
  double k = rint(x * MMM_TWO_OVER_PI);
  double r = fma(-k, MMM_PIO2_1, x);
  r = fma(-k, MMM_PIO2_1T, r);
  *quadrant = ((int)k) & 3;
  return r;
}
MMM_FN double mmm_sin(double x) {
  int q;
  double r = mmm_reduce(x, &q);
  double s = (q & 1) ? mmm_kcos(r) : mmm_ksin(r);
  return (q & 2) ? -s : s;
}
MMM_FN double mmm_cos(double x) {
  int q;
  double r = mmm_reduce(x, &q);
  double c = (q & 1) ? mmm_ksin(r) : mmm_kcos(r);
  return ((q + 1) & 2) ? -c : c;
}
MMM_FN void mmm_sincos(double x, double *s, double *c) {
  int q;
  double r = mmm_reduce(x, &q);
  double ks = mmm_ksin(r), kc = mmm_kcos(r);
  double ss = (q & 1) ? kc : ks, cc = (q & 1) ? ks : kc;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
/* tan for |x| < pi/2 (steering angles are clipped to pi/3) */
MMM_FN double mmm_tan(double x) {
  double s, c;
  mmm_sincos(x, &s, &c);
  return s / c;
}
/* atan(num / den) for 0 <= num <= den, den > 0, WITHOUT forming the ratio: nearest breakpoint c = j/4 of q = num/den
 * (q >= (k - 1/2)/4  <=>  num >= ((k - 1/2)/4) den), t = (q - c) / (1 + q c) = (num - c den) / (den + c num), |t| <= 1/8,
 * Taylor to t^21.  One division where asin(x) = atan2(|x|, sqrt((1-|x|)(1+|x|))) used to need two (the ratio, then t). */
MMM_FN double mmm_atan_ratio(double num, double den, double *lo_out) {
  const int j = (num >= 0.125 * den) + (num >= 0.375 * den) + (num >= 0.625 * den) + (num >= 0.875 * den);
  double c = 0.25 * (double)j;
  double t = (num - c * den) / (den + c * num); /* j == 0: num / den */
  double z = t * t;
  double p = MMM_K(mmm_t_atan, 0, MMM_A10);
  p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 1, MMM_A9)); p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 2, MMM_A8)); p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 3, MMM_A7)); p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 4, MMM_A6));
  p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 5, MMM_A5)); p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 6, MMM_A4)); p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 7, MMM_A3)); p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 8, MMM_A2));
  p = mmm_fma_c(p, z, MMM_K(mmm_t_atan, 9, MMM_A1));
  double pt = fma(t * z, p, t);
  double hi = j == 0 ? 0.0 : (j == 1 ? MMM_ATAN_HI_1 : (j == 2 ? MMM_ATAN_HI_2 : (j == 3 ? MMM_ATAN_HI_3 : MMM_ATAN_HI_4)));
  double lo = j == 0 ? 0.0 : (j == 1 ? MMM_ATAN_LO_1 : (j == 2 ? MMM_ATAN_LO_2 : (j == 3 ? MMM_ATAN_LO_3 : MMM_ATAN_LO_4)));
  *lo_out = lo;
  return hi + (pt + lo);
}
MMM_FN double mmm_atan(double x) {
  double ax = fabs(x), lo;
  const int inv = ax > 1.0; /* atan(ax) = pi/2 - atan(1 / ax): the ratio 1 / ax is never formed either */
  double r = mmm_atan_ratio(inv ? 1.0 : ax, inv ? ax : 1.0, &lo);
  if (inv) r = MMM_PIO2_HI - (r - MMM_PIO2_LO);
  return x < 0 ? -r : r;
}
/* asin(x) = atan2(|x|, sqrt((1-|x|)(1+|x|))), |x| <= 1; *w_out = sqrt((1-|x|)(1+|x|)) = cos(asin x), which the bicycle
 * step reuses (sin / cos of a steering angle that IS an arcsine: no second range reduction, no polynomials) */
MMM_FN double mmm_asin_w(double x, double *w_out) {
  double y = fabs(x), lo;
  double w = sqrt((1.0 - y) * (1.0 + y));
  *w_out = w;
  const int direct = y <= w;
  double r = mmm_atan_ratio(direct ? y : w, direct ? w : y, &lo); /* one division, one polynomial */
  if (!direct) r = MMM_PIO2_HI - (r - MMM_PIO2_LO);
  return x < 0 ? -r : r;
}
MMM_FN double mmm_asin(double x) { double w; return mmm_asin_w(x, &w); }
/* sin / cos of the steering limit pi/3 (the double 0x1.0c152382d7365p+0), correctly rounded */
#define MMM_SIN_PI3 0x1.bb67ae8584caap-1
#define MMM_COS_PI3 0x1.0000000000001p-1
/* exp: x = k ln2 + r, |r| <= ln2/2, Taylor to r^14, exact scaling */
MMM_FN double mmm_exp(double x) {
  if (x < -1000.0) return 0.0;
  double k = rint(x * MMM_INV_LN2);
  double r = fma(-k, MMM_LN2_HI, x);
  r = fma(-k, MMM_LN2_LO, r);
  double p = MMM_K(mmm_t_exp, 0, MMM_E14);
  p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 1, MMM_E13)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 2, MMM_E12)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 3, MMM_E11)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 4, MMM_E10));
  p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 5, MMM_E9)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 6, MMM_E8)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 7, MMM_E7)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 8, MMM_E6));
  p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 9, MMM_E5)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 10, MMM_E4)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 11, MMM_E3)); p = mmm_fma_c(p, r, MMM_K(mmm_t_exp, 12, MMM_E2));
  p = fma(p, r * r, r) + 1.0; /* 1 + r + r^2 (1/2 + ...) */
  return ldexp(p, (int)k);
}
/* log: x = m 2^e with m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), atanh series to s^23 */
MMM_FN double mmm_log(double x) {
  int e;
  double m = frexp(x, &e); /* m in [0.5, 1) */
  if (m < MMM_SQRT_HALF) { m = m * 2.0; e -= 1; }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double p = MMM_K(mmm_t_log, 0, MMM_L11);
  p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 1, MMM_L10)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 2, MMM_L9)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 3, MMM_L8)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 4, MMM_L7));
  p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 5, MMM_L6)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 6, MMM_L5)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 7, MMM_L4)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 8, MMM_L3));
  p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 9, MMM_L2)); p = mmm_fma_c(p, z, MMM_K(mmm_t_log, 10, MMM_L1));
  double l = 2.0 * fma(s * z, p, s);
  double de = (double)e;
  return fma(de, MMM_LN2_HI, l + de * MMM_LN2_LO);
}

/* ---- angle-sum forms used by the DEVICE arithmetic of the bicycle step (mm_kernels.hip: predict; oracle math mode 1) ----
 * The reference evaluates np.arctan(1/2 tan(delta)), np.cos(psi + beta), np.sin(psi + beta), np.sin(beta),
 * np.cos(psi' + beta) and the three corner angles alpha + psi', -alpha + psi' as eleven separate libm calls per vehicle
 * step (kinematics.py:122-141, safe_controller.py:151-172, controller.py:257-267).  With sin / cos of the heading carried
 * along, all of them follow from ONE sincos of the steering angle and ONE of the new heading:
 *   t = tan(beta) = 1/2 tan(delta);  cos(beta) = 1 / sqrt(1 + t^2),  sin(beta) = t / sqrt(1 + t^2)   (|beta| < pi/2)
 *   cos(a + b) = cos a cos b - sin a sin b,   sin(a + b) = sin a cos b + cos a sin b.
 * Same mathematical quantities, a few ulp from the separately evaluated ones -- the same class of deviation as these
 * functions themselves have from libm; both users of this header evaluate exactly these expressions. */
#define MMM_CORNER_COS 0x1.db614bfce6a6ap-1 /* cos(atan(0.4)) = 1 / sqrt(1.16) = 0.9284766908852593 */
#define MMM_CORNER_SIN 0x1.7c4dd663ebb88p-2 /* sin(atan(0.4)) = 0.4 / sqrt(1.16) = 0.3713906763541037 */
MMM_FN double mmm_cos_sum(double sa, double ca, double sb, double cb) { return ca * cb - sa * sb; }
MMM_FN double mmm_sin_sum(double sa, double ca, double sb, double cb) { return sa * cb + ca * sb; }
MMM_FN void mmm_slip_sincos(double t, double *sb, double *cb) { /* beta = atan(t) */
  const double r = sqrt(1.0 + t * t);
  *cb = 1.0 / r;
  *sb = t / r;
}

#endif /* MM_MATH_H */
