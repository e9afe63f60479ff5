// mm_device.h -- per-vehicle device math of the merge env (gfx950, fp64).
//
// Scalar building blocks used by the fused step kernel in mm_kernels.hip.  Each function cites
// the reference lines (hkbharath/MARL-MASS) whose arithmetic it reproduces, operation for
// operation (the library is built with -ffp-contract=off so a*b+c keeps two roundings like the
// reference's Python floats).  No CUDA-isms: wave64, HIP only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mm_abi.h"

#define MM_DEV __device__ __forceinline__
#define MMM_FN __device__ __forceinline__
#include "../../include/mm_math.h"
// device builds stop a QP whose iterate is provably frozen instead of running it to cvxopt's iteration cap (mm_qp_frozen in
// include/mm_qp.h: same result bits; the oracle runs the literal loop).  -DMM_QP_NO_CERTIFY: A-B timing / validation builds.
#ifndef MM_QP_NO_CERTIFY
#define MM_QP_CERTIFY
#endif
#include "../../include/mm_qp.h"  // the shield QP through cvxopt's interior-point algorithm (MM_QP_IPM)

namespace mm {

constexpr double kPi = 3.141592653589793;  // np.pi
constexpr double kVehLength = 5.0, kVehWidth = 2.0, kMaxSpeed = 40.0;    // kinematics.py:27-33
constexpr double kTauA = 0.6, kTauDs = 0.2;                               // controller.py:23-24
constexpr double kLcMaxAcc = 6.0, kLcMinAcc = -12.5;                      // safe_controller.py:17-20
constexpr double kPerception = 180.0, kStoppingSpeed = 1.6667;            // safe_controller.py:21,25
constexpr double kLaneWidth = 4.0;                                        // lane.py:15
constexpr double kCbfAccLo = -12.5, kCbfAccHi = 6.0, kAdjBuffer = 2.0134; // cbf.py:202,27
constexpr double kSineAmp = 3.25, kSinePuls = 2 * kPi / (2 * 100.0), kSinePhase = kPi / 2;
constexpr double kObstX = 420.0, kObstY = 4.0;                            // merge_env_v1.py:247
// controller.py:32-33 (np.sqrt(1 + 6.25) + 0.0075, np.arctan(0.4)) as correctly rounded doubles
// sqrt-free forms of the two distance tests (np.linalg.norm = sqrt(dx*dx + dy*dy), correctly rounded and
// monotone): sqrt(s) < 180  <=>  s < kT180 (the smallest double whose root rounds to >= 180), and
// sqrt(s) > 5  <=>  s > kU5 (the largest double whose root rounds to <= 5).  The oracle keeps the sqrt.
constexpr double kT180 = 0x1.fa3ffffffffffp+14;  // 32399.999999999996
constexpr double kU5 = 0x1.9000000000001p+4;     // 25.000000000000004
constexpr double kCornerLen = 2.7000824035672517;
constexpr double kCornerAlpha = 0.3805063771123649;

// ---- lane table (merge_env_v1.py:222-248); ids in network insertion order -------------------
MM_DEV double lane_sx(int l) { return l == 0 ? 0.0 : (l <= 2 ? 320.0 : (l == 3 ? 420.0 : (l == 4 ? 0.0 : 220.0))); }
MM_DEV double lane_sy(int l) { return l == 2 ? 4.0 : (l == 4 ? 10.5 : (l == 5 ? 7.25 : 0.0)); }
MM_DEV double lane_len(int l) { return l == 0 ? 320.0 : (l == 3 ? 1000.0 : (l == 4 ? 220.0 : 100.0)); }
MM_DEV bool lane_forbidden(int l) { return l == 2 || l >= 4; }
MM_DEV int lane_road(int l) { return l == 0 ? 0 : (l <= 2 ? 1 : l - 1); }  // (from,to) pair id
MM_DEV int lane_rid(int l) { return l == MM_LANE_BC1 ? 1 : 0; }            // lane id within the road

// ---- utils.py ---------------------------------------------------------------------------------
// C fmod(a, b) for b > 0, exact like the libm one.  |a| < 2b (every heading difference on this path)
// needs no division: the result is a, or a -/+ b (exact by Sterbenz).  Otherwise q = floor(|a|/b) can
// be one too large only when |a|/b rounds up to an integer; r = fma(-q, b, |a|) is exact either
// way (the true remainder is a multiple of ulp(b) below b) and one exact correction fixes it.
MM_DEV double fmod_pos(double a, double b) {
  const double aa = fabs(a);
  double r;
  if (aa < b) r = aa;
  else if (aa < 2 * b) r = aa - b;
  else {
    const double q = floor(aa / b);
    r = fma(-q, b, aa);
    if (r < 0) r += b;
  }
  return a < 0 ? -r : r;
}
MM_DEV double py_mod(double a, double b) {  // Python float % for b > 0: result in [0, b)
  double m = fmod_pos(a, b);
  if (m != 0.0) {
    if (m < 0) m += b;
  } else {
    m = 0.0;  // copysign(0, b)
  }
  return m;
}
MM_DEV double wrap_to_pi(double x) {  // utils.py:40-41
  const double a = x + kPi;
  // every lane of the wave already inside [0, 2 pi) (any heading difference on this path): py_mod returns `a` itself there
  if (__all(a >= 0.0 && a < 2 * kPi)) return a - kPi;
  return py_mod(a, 2 * kPi) - kPi;
}
MM_DEV double not_zero(double x) {                                              // utils.py:31-37
  return fabs(x) > 1e-2 ? x : (x > 0 ? 1e-2 : -1e-2);
}
MM_DEV double clipd(double x, double a, double b) { return fmin(fmax(x, a), b); }
// Correctly rounded x / d for a divisor known in advance, in 3 instructions instead of the ~25 of a
// full IEEE fp64 division (Markstein 1990: with y = RN(1/d), q = RN(x*y) and the exact residual
// r = x - q*d (one fma), RN(q + r*y) is the correctly rounded quotient; the one exception, a divisor
// whose significand is all ones, does not occur among the constants used).  Bit-identical to `/`,
// which is what the reference computes; checked against true division in test_math_bits_cpu_vs_gpu
// and by every bit-exact oracle comparison.
MM_DEV double div_c(double x, double d, double y) {
  const double q = x * y;
  const double r = fma(-q, d, x);
  return fma(r, y, q);
}
#define MM_DIVC(x, d) ::mm::div_c((x), (d), 1.0 / (d))  // d must be a compile-time constant

// ---- road/lane.py -----------------------------------------------------------------------------
MM_DEV void lane_local(int l, double x, double y, double &s, double &r) {  // lane.py:164-168,208-210
  s = x - lane_sx(l);
  r = y - lane_sy(l);
  if (l == MM_LANE_KB0) r = r - kSineAmp * mmm_sin(kSinePuls * s + kSinePhase);
}
MM_DEV double lane_heading_at(int l, double s) {  // lane.py:158-159, :204-206
  return l == MM_LANE_KB0 ? 0.0 + mmm_atan(kSineAmp * kSinePuls * mmm_cos(kSinePuls * s + kSinePhase)) : 0.0;
}
MM_DEV double lane_distance(int l, double x, double y) {  // lane.py:97-100
  double s, r;
  lane_local(l, x, y, s, r);
  return fabs(r) + fmax(s - lane_len(l), 0.0) + fmax(0.0 - s, 0.0);
}
MM_DEV bool lane_on_lane(int l, double x, double y) {  // lane.py:61-76, margin 0
  double s, r;
  lane_local(l, x, y, s, r);
  return fabs(r) <= kLaneWidth / 2 + 0 && (-kVehLength <= s && s < lane_len(l) + kVehLength);
}
MM_DEV bool lane_reachable(int l, double x, double y) {  // lane.py:78-90
  if (lane_forbidden(l)) return false;
  double s = x - lane_sx(l), r = y - lane_sy(l);  // only straight lanes are not forbidden
  return fabs(r) <= 2 * kLaneWidth && (0 <= s && s < lane_len(l) + kVehLength);
}
MM_DEV bool lane_after_end(int l, double x) {  // lane.py:92-95 (longitudinal only)
  return (x - lane_sx(l)) > lane_len(l) - kVehLength / 2;
}

// ---- road/road.py -----------------------------------------------------------------------------
// road.py:51-65 get_closest_lane_index: argmin of distance_with_heading (lane.py:102-108), first
// minimum in insertion order.  The five straight lanes share heading 0, so |wrap(h - 0)| is
// computed once; only kb0 needs the sine frame.
MM_DEV int closest_lane(double x, double y, double h) {
  const double ang0 = fabs(wrap_to_pi(h - 0.0));
  int best = 0;
  double bd = 0;
#pragma unroll
  for (int l = 0; l < 5; l++) {
    double s = x - lane_sx(l), r = y - lane_sy(l);
    double d = fabs(r) + fmax(s - lane_len(l), 0.0) + fmax(0.0 - s, 0.0) + 1.0 * ang0;
    if (l == 0 || d < bd) { bd = d; best = l; }
  }
  {
    double s = x - 220.0;
    double tail = fmax(s - 100.0, 0.0), head = fmax(0.0 - s, 0.0);
    // every term of the kb0 distance is >= 0 and fp addition is monotone, so (0 + tail) + head
    // bounds it from below: skip the transcendental frame when kb0 cannot win (ties lose: last id)
    if (!((0.0 + tail) + head >= bd)) {
      double ph = kSinePuls * s + kSinePhase;
      double r = (y - 7.25) - kSineAmp * mmm_sin(ph);
      double lh = 0.0 + mmm_atan(kSineAmp * kSinePuls * mmm_cos(ph));
      double d = fabs(r) + tail + head + 1.0 * fabs(wrap_to_pi(h - lh));
      if (d < bd) { bd = d; best = MM_LANE_KB0; }
    }
  }
  return best;
}
// road.py:67-109 next_lane on a->b->c->d, j->k->b (one successor per node, no route)
MM_DEV int next_lane(int l, double x, double y) {
  if (l == MM_LANE_AB0 || l == MM_LANE_KB0)  // 1 lane -> 2 lanes: min(lane.distance), first wins
    return lane_distance(MM_LANE_BC0, x, y) <= lane_distance(MM_LANE_BC1, x, y) ? MM_LANE_BC0 : MM_LANE_BC1;
  if (l == MM_LANE_JK0) return MM_LANE_KB0;
  return MM_LANE_CD0;  // bc0, bc1 -> cd0 ; cd0 -> cd0 (KeyError branch)
}

// ---- vehicle/controller.py --------------------------------------------------------------------
MM_DEV int speed_to_index(double speed) {  // controller.py:327-337, np.round = half-to-even
  double x = MM_DIVC(speed - 10, 20.0);
  return (int)clipd(rint(x * (5 - 1)), 0, 5 - 1);
}
MM_DEV double index_to_speed(int i) { return 10 + i * (30.0 - 10) / (5 - 1); }  // :313-325

// half_tan: 1/2 tan of the returned (clipped) angle -- what the bicycle step needs of it (beta = atan(1/2 tan delta)).  The
// command is an arcsine, so sin / cos of it are its argument and the square root asin already formed (or the constants at
// the +-pi/3 limit): no second range reduction, no polynomials (device arithmetic, oracle math mode 1 alike)
MM_DEV double steering_control(double x, double y, double heading, double speed, int tl, double &half_tan) {  // :146-187
  constexpr double KP_HEADING = 1 / kTauDs, KP_LATERAL = 1.0 / 3 * KP_HEADING, PURSUIT_TAU = 0.5 * kTauDs;
  double s, r;
  lane_local(tl, x, y, s, r);
  double lane_next = s + speed * PURSUIT_TAU;
  double lfh = lane_heading_at(tl, lane_next);
  double lat_cmd = -KP_LATERAL * r;
  double nz = not_zero(speed);
  double heading_command = mmm_asin(clipd(lat_cmd / nz, -1, 1));
  double heading_ref = lfh + clipd(heading_command, -kPi / 4, kPi / 4);
  double rate = KP_HEADING * wrap_to_pi(heading_ref - heading);
  const double arg = clipd(kVehLength / 2 / nz * rate, -1, 1);
  double w;
  const double steer = mmm_asin_w(arg, &w);
  const bool sat = fabs(steer) > kPi / 3;
  const double ss = sat ? (arg < 0 ? -MMM_SIN_PI3 : MMM_SIN_PI3) : arg;
  const double cs = sat ? MMM_COS_PI3 : w;
  half_tan = 1.0 / 2 * (ss / cs);
  return clipd(steer, -kPi / 3, kPi / 3);
}
MM_DEV double steering_control(double x, double y, double heading, double speed, int tl) {
  double ht;
  return steering_control(x, y, heading, speed, tl, ht);
}

// ---- utils.py rotated-rectangle intersection (:55-121), +angle rotation quirk kept ------------
MM_DEV bool has_corner_inside(double c1x, double c1y, double l1, double w1, double a1, double c2x,
                              double c2y, double l2, double w2, double a2) {
  const double lx = l1 / 2, wy = w1 / 2;
  double c = mmm_cos(a1), s = mmm_sin(a1), c2 = mmm_cos(a2), s2 = mmm_sin(a2);
  const double px[9] = {0, -lx, lx, 0, 0, -lx, -lx, lx, lx};
  const double py[9] = {0, 0, 0, -wy, wy, -wy, wy, -wy, wy};
  bool any = false;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    double rx = c * px[k] + (-s) * py[k], ry = s * px[k] + c * py[k];
    double dx = (c1x + rx) - c2x, dy = (c1y + ry) - c2y;
    double ru0 = c2 * dx + (-s2) * dy, ru1 = s2 * dx + c2 * dy;
    any |= ((-l2 / 2 <= ru0) & (ru0 <= l2 / 2)) & ((-w2 / 2 <= ru1) & (ru1 <= w2 / 2));  // no short-circuit branches
  }
  return any;
}
// Exact early-out for rects_intersect.  The 9-point test can only succeed if a point of box A lies inside box B (B as the
// upstream code rotates it, i.e. by +heading or -heading: the bound below is symmetric in the sign).  Along a WORLD axis
// a box with half-extents (l, w) and heading h reaches at most l + w|h| in x and w + l|h| in y (cos h <= 1, |sin h| <= |h|
// for every h), so if the centres are further apart than the two reaches on either axis the boxes are disjoint and every
// point test is false.  The 1e-6 margin dwarfs the rounding of the test itself (~1e-13): never a different answer.
MM_DEV bool boxes_may_touch(double dx, double dy, double ha, double hb, double lb, double wb) {
  const double aa = fabs(ha), ab = fabs(hb);
  const double la = 0.9 * kVehLength / 2, wa = 0.9 * kVehWidth / 2;
  const double rx = (la + wa * aa) + (lb + wb * ab) + 1e-6, ry = (wa + la * aa) + (wb + lb * ab) + 1e-6;
  return !(fabs(dx) > rx) && !(fabs(dy) > ry);
}
// kinematics.py:202-209 _is_colliding (caller did the 5 m pre-check)
MM_DEV bool rects_intersect(double ax, double ay, double ah, double ox, double oy, double ol,
                            double ow, double oh) {
  return has_corner_inside(ax, ay, 0.9 * kVehLength, 0.9 * kVehWidth, ah, ox, oy, 0.9 * ol, 0.9 * ow, oh) ||
         has_corner_inside(ox, oy, 0.9 * ol, 0.9 * ow, oh, ax, ay, 0.9 * kVehLength, 0.9 * kVehWidth, ah);
}

// ---- counter-based RNG for the device reset (Philox4x32-10, Salmon et al. 2011) ---------------
MM_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
MM_DEV void rng_block(uint64_t seed, uint32_t episode, uint32_t blk, uint32_t out[4]) {
  philox4x32(blk, episode, 0u, 0x4D4D5253u, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
MM_DEV double u53(uint32_t a, uint32_t b) {
  return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0;
}

}  // namespace mm
