/*
 * mm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C, scalar, one-env-at-a-time restatement of the hot path of hkbharath/MARL-MASS
 * (env.reset / env.step of merge-multi-agent-v0/-v1 + the HSS / MASS CBF shield), following the
 * reference Python file by file; every function cites the reference lines it restates
 * (paths relative to the reference root).  It exports the same C ABI as the HIP library
 * (include/mm_abi.h) on HOST pointers.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported CPU baseline.  The product path never routes through it.
 *
 * Parity pin: checked against golden vectors generated from the reference itself
 * (tools/gen_golden.py -> tests/golden/, test_oracle_golden.py).  The shield QP has two modes: its exact
 * KKT closed form, and qp_ipm -- a restatement of cvxopt's coneqp algorithm pinned bit for bit against the
 * reference-side stand-in that answered solvers.qp while the reference produced the ipm_* tapes.  cvxopt 1.2.7
 * itself is NOT available in this image: against the real binary the raw solver output (its BLAS/LAPACK
 * rounding) stays "parity unpinned" (see DESIGN.md section 3).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).  -ffp-contract=off
 * keeps a*b+c as two roundings like CPython/numpy scalar arithmetic.
 */
#include "../include/mm_abi.h"
#include "../include/mm_counts.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.141592653589793 /* np.pi */

/*
 * Elementary-function mode.  0 (default): the platform libm, i.e. the reference's own arithmetic
 * (CPython/numpy call the same libm) -- this is the mode pinned bit-tight to the golden tapes.
 * 1: the DEVICE arithmetic -- the bit-reproducible functions of include/mm_math.h and, in the bicycle step and the corner
 * points, its angle-sum forms (one sincos of the steering angle, one of the new heading) -- identical to what the HIP kernels
 * evaluate, so that HIP-vs-oracle comparisons can demand equality of every bit (decisions such as
 * the LC veto are rounding-noise knife-edges, see mm_math.h).  Mode 1 is itself checked against
 * the golden tapes (<= 1e-9, same flags) in tests/test_oracle_golden.py.
 */
#define MMM_FN static inline
#include "../include/mm_math.h"
static double pow2_libm(double x) { return pow(x, 2); }
static double pow2_mul(double x) { return x * x; }
static double pow4_libm(double x) { return pow(x, 4.0); }
static double pow4_mul(double x) { double y = x * x; return y * y; }
static double (*m_sin)(double) = sin, (*m_cos)(double) = cos, (*m_tan)(double) = tan;
static double (*m_atan)(double) = atan, (*m_asin)(double) = asin, (*m_exp)(double) = exp;
static double (*m_log)(double) = log, (*m_sq)(double) = pow2_libm, (*m_p4)(double) = pow4_libm;
static int g_math_mode = 0;
int32_t orc_set_math(int32_t mode) {
  g_math_mode = mode;
  if (mode == 0) { m_sin = sin; m_cos = cos; m_tan = tan; m_atan = atan; m_asin = asin; m_exp = exp; m_log = log; m_sq = pow2_libm; m_p4 = pow4_libm; }
  else { m_sin = mmm_sin; m_cos = mmm_cos; m_tan = mmm_tan; m_atan = mmm_atan; m_asin = mmm_asin; m_exp = mmm_exp; m_log = mmm_log; m_sq = pow2_mul; m_p4 = pow4_mul; }
  return g_math_mode;
}

/* ------------------------------------------------------------------ constants */
#define VEH_LENGTH 5.0           /* kinematics.py:27 */
#define VEH_WIDTH 2.0            /* kinematics.py:29 */
#define MAX_SPEED 40.0           /* kinematics.py:33 */
#define OBST_LENGTH 2.0          /* objects.py:18 */
#define OBST_WIDTH 2.0           /* objects.py:19 */
#define TAU_A 0.6                /* controller.py:23 */
#define TAU_DS 0.2               /* controller.py:24 */
#define LC_MAX_ACC 6.0           /* safe_controller.py:17 */
#define LC_MIN_ACC (-12.5)       /* safe_controller.py:20 */
#define PERCEPTION_DIST 180.0    /* safe_controller.py:21, abstract.py:41 */
#define STOPPING_SPEED 1.6667    /* safe_controller.py:25 */
#define LANE_WIDTH 4.0           /* lane.py:15 */
#define CBF_ACC_LO (-12.5)       /* cbf.py:202 CBF_AV.ACCELERATION_RANGE */
#define CBF_ACC_HI 6.0
#define ADJ_BUFFER 2.0134        /* cbf.py:27 */

/* lane table: merge_env_v1.py:222-248 (ends = [220, 100, 100, 1000], abstract.py:78) */
static const double LANE_SX[6] = {0.0, 320.0, 320.0, 420.0, 0.0, 220.0};
static const double LANE_SY[6] = {0.0, 0.0, 4.0, 0.0, 10.5, 7.25};
static const double LANE_LEN[6] = {320.0, 100.0, 100.0, 1000.0, 220.0, 100.0};
static const int LANE_FORBIDDEN[6] = {0, 0, 1, 0, 1, 1};
#define SINE_AMP 3.25
#define SINE_PULS (2 * PI / (2 * 100.0)) /* 2*np.pi / (2*ends[1]) */
#define SINE_PHASE (PI / 2)
#define OBST_X 420.0 /* lbc.position(ends[2], 0) */
#define OBST_Y 4.0

typedef struct {
  double x, y, heading, speed, target_speed;
  double act_steer, act_acc;   /* self.action */
  /* math mode 1 only: sin / cos of the last steering_control result, as the device carries them into the bicycle step
   * (valid while act_steer / the veto steering still IS that result; see vehicle_step) */
  double sc_steer, sc_sin, sc_cos; int sc_valid;
  double safe_steer, safe_acc; /* self.safe_action */
  double g_vx;                 /* fg_params["g"]["vx"] */
  double h1[2], h2[2];         /* state_hist[-1], [-2]: x, vx (all the shield reads of a record) */
  int lane, target_lane, speed_index, crashed, hl_action, flags, hist_len, kind; /* kind: 1 CAV, 2 HDV */
  double timer;                /* IDMVehicle.timer (behavior.py:53) */
  double steer_angle;          /* MDPLCVehicle.steering_angle (safe_controller.py:54) */
  int steer_vel;               /* lateral_ctrl == "steer_vel" on this (MDPLC) vehicle (:44) */
  double local_reward, regional_reward;
  /* trace of the last shield call */
  double qp_rows, qp_a, qp_h[4], qp_d, lc_margin;
  int qp_optimal, qp_bounds;   /* sol["status"] != "unknown" (cbf.py:140); check_bounds would raise (cbf.py:87-96) */
  int lon_safe, lon_invariant; /* CBF_AV.update_status cbf.py:341-351 */
  double shield_headway;       /* vehicle.min_headway as the last shield call set it */
} Veh;

typedef struct {
  int n;      /* vehicles on the road (road.vehicles, creation order: CAVs first) */
  int n_ctrl; /* controlled vehicles = the leading kind-1 entries */
  Veh v[MM_MAX_AGENTS];
  int steps, time, n_merge, episode;
} Env;

struct MMHandle_ {
  MMConfig cfg;
  int E, N;
  unsigned char *state;
  MMStateLayout lay;
  int64_t first_env;
  double *metrics;
  char err[256];
  int latched; /* error raised inside step(), reported by mm_poll_errors (the HIP twin cannot return it from a launch) */
};

#ifdef ORC_QP_LOG
/* Diagnostic build only (oracle/Makefile: libmm_oracle_qplog.so, used by tools/ipm_sched_model.py): one text line per
 * shield QP -- env, sub-step clock, vehicle, sweep rank, IPM iterations, status, rows, and the vehicles whose decision of
 * THIS sub-step the QP's right-hand side reads (leader / front-adjacent; -1: none) -- to model wave schedules on the CPU. */
static FILE *g_qplog;
static int g_qplog_env, g_qplog_rank, g_qplog_dep_ol, g_qplog_dep_oa;
static unsigned g_qplog_stepped;
void orc_qplog_open(const char *path) { if (g_qplog) fclose(g_qplog); g_qplog = path ? fopen(path, "w") : NULL; }
#endif

/* ------------------------------------------------------------------ utils.py */

/* utils.py:40-41 wrap_to_pi with Python float % semantics (sign of the divisor) */
static double py_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}
static double wrap_to_pi(double x) { return py_mod(x + PI, 2 * PI) - PI; }

/* utils.py:31-37 */
static double not_zero(double x) {
  const double eps = 1e-2;
  if (fabs(x) > eps) return x;
  else if (x > 0) return eps;
  else return -eps;
}
static double clipd(double x, double a, double b) { return fmin(fmax(x, a), b); }

/* utils.py:55-70 point_in_rotated_rectangle; NOTE rotates by +angle (upstream quirk, kept) */
static int point_in_rotated_rectangle(double px, double py, double cx, double cy, double length,
                                      double width, double angle) {
  double c = m_cos(angle), s = m_sin(angle);
  double dx = px - cx, dy = py - cy;
  double ru0 = c * dx + (-s) * dy;
  double ru1 = s * dx + c * dy;
  return (-length / 2 <= ru0 && ru0 <= length / 2) && (-width / 2 <= ru1 && ru1 <= width / 2);
}

/* utils.py:102-121 has_corner_inside: 9 points of rect1 tested against rect2 */
static int has_corner_inside(double c1x, double c1y, double l1, double w1, double a1, double c2x,
                             double c2y, double l2, double w2, double a2) {
  double lx = l1 / 2, wy = w1 / 2;
  double pts[9][2] = {{0, 0},     {-lx, 0},  {lx, 0},    {0, -wy},  {0, wy},
                      {-lx, -wy}, {-lx, wy}, {lx, -wy}, {lx, wy}};
  double c = m_cos(a1), s = m_sin(a1);
  for (int k = 0; k < 9; k++) {
    double rx = c * pts[k][0] + (-s) * pts[k][1];
    double ry = s * pts[k][0] + c * pts[k][1];
    if (point_in_rotated_rectangle(c1x + rx, c1y + ry, c2x, c2y, l2, w2, a2)) return 1;
  }
  return 0;
}
/* utils.py:90-99 */
static int rotated_rectangles_intersect(double c1x, double c1y, double l1, double w1, double a1,
                                        double c2x, double c2y, double l2, double w2, double a2) {
  return has_corner_inside(c1x, c1y, l1, w1, a1, c2x, c2y, l2, w2, a2) ||
         has_corner_inside(c2x, c2y, l2, w2, a2, c1x, c1y, l1, w1, a1);
}

/* ------------------------------------------------------------------ road/lane.py */

/* lane.py:164-168 (StraightLane) and :208-210 (SineLane); direction = (1, 0) for every lane */
static void lane_local(int lane, double x, double y, double *s, double *r) {
  double lon = x - LANE_SX[lane];
  double lat = y - LANE_SY[lane];
  if (lane == MM_LANE_KB0) lat = lat - SINE_AMP * m_sin(SINE_PULS * lon + SINE_PHASE);
  *s = lon;
  *r = lat;
}
/* lane.py:158-159, :204-206 */
static double lane_heading_at(int lane, double s) {
  if (lane == MM_LANE_KB0) return 0.0 + m_atan(SINE_AMP * SINE_PULS * m_cos(SINE_PULS * s + SINE_PHASE));
  return 0.0;
}
/* lane.py:97-100 distance */
static double lane_distance(int lane, double x, double y) {
  double s, r;
  lane_local(lane, x, y, &s, &r);
  return fabs(r) + fmax(s - LANE_LEN[lane], 0) + fmax(0 - s, 0);
}
/* lane.py:102-108 distance_with_heading (heading_weight = 1) */
static double lane_distance_with_heading(int lane, double x, double y, double heading) {
  double s, r;
  lane_local(lane, x, y, &s, &r);
  double angle = fabs(wrap_to_pi(heading - lane_heading_at(lane, s)));
  return fabs(r) + fmax(s - LANE_LEN[lane], 0) + fmax(0 - s, 0) + 1.0 * angle;
}
/* lane.py:61-76 on_lane (margin 0; longitudinal/lateral recomputed) */
static int lane_on_lane(int lane, double x, double y) {
  double s, r;
  lane_local(lane, x, y, &s, &r);
  return fabs(r) <= LANE_WIDTH / 2 + 0 && (-VEH_LENGTH <= s && s < LANE_LEN[lane] + VEH_LENGTH);
}
/* lane.py:78-90 is_reachable_from */
static int lane_is_reachable_from(int lane, double x, double y) {
  if (LANE_FORBIDDEN[lane]) return 0;
  double s, r;
  lane_local(lane, x, y, &s, &r);
  return fabs(r) <= 2 * LANE_WIDTH && (0 <= s && s < LANE_LEN[lane] + VEH_LENGTH);
}
/* lane.py:92-95 after_end */
static int lane_after_end(int lane, double x, double y) {
  double s, r;
  lane_local(lane, x, y, &s, &r);
  return s > LANE_LEN[lane] - VEH_LENGTH / 2;
}

/* ------------------------------------------------------------------ road/road.py */

/* road.py:51-65 get_closest_lane_index: first minimum in insertion order */
static int closest_lane(double x, double y, double heading) {
  int best = 0;
  double bd = lane_distance_with_heading(0, x, y, heading);
  for (int l = 1; l < 6; l++) {
    double d = lane_distance_with_heading(l, x, y, heading);
    if (d < bd) { bd = d; best = l; }
  }
  return best;
}
/* road.py:67-109 next_lane on the merge graph a->b->c->d, j->k->b (no route, one successor each) */
static int next_lane(int lane, double x, double y) {
  switch (lane) {
    case MM_LANE_AB0: /* 1 lane -> 2 lanes: closest by lane.distance, first min wins */
    case MM_LANE_KB0:
      return lane_distance(MM_LANE_BC0, x, y) <= lane_distance(MM_LANE_BC1, x, y) ? MM_LANE_BC0
                                                                                  : MM_LANE_BC1;
    case MM_LANE_BC0:
    case MM_LANE_BC1: return MM_LANE_CD0; /* 2 lanes -> 1 lane */
    case MM_LANE_CD0: return MM_LANE_CD0; /* KeyError branch: end of network */
    case MM_LANE_JK0: return MM_LANE_KB0; /* same lane count: keep id 0 */
  }
  return lane;
}
static int lane_road(int lane) { /* (from,to) pair id */
  static const int r[6] = {0, 1, 1, 2, 3, 4};
  return r[lane];
}
static int lane_id_in_road(int lane) { return lane == MM_LANE_BC1 ? 1 : 0; }

/* ------------------------------------------------------------------ vehicle/controller.py */

/* controller.py:327-337 speed_to_index (np.round = round-half-even = rint) */
static int speed_to_index(double speed) {
  double x = (speed - 10) / (30 - 10);
  return (int)clipd(rint(x * (5 - 1)), 0, 5 - 1);
}
/* controller.py:313-325 */
static double index_to_speed(int index) { return 10 + index * (30.0 - 10) / (5 - 1); }

/* controller.py:136-144 follow_road */
static void follow_road(Veh *v) {
  if (lane_after_end(v->target_lane, v->x, v->y))
    v->target_lane = next_lane(v->target_lane, v->x, v->y);
}
/* controller.py:146-187 steering_control */
static double steering_control(Veh *v, int target_lane) {
  const double KP_HEADING = 1 / TAU_DS, KP_LATERAL = 1.0 / 3 * KP_HEADING;
  const double PURSUIT_TAU = 0.5 * TAU_DS, MAX_STEER = PI / 3;
  double s, r;
  lane_local(target_lane, v->x, v->y, &s, &r);
  double lane_next = s + v->speed * PURSUIT_TAU;
  double lane_future_heading = lane_heading_at(target_lane, lane_next);
  double lateral_speed_command = -KP_LATERAL * r;
  double heading_command = m_asin(clipd(lateral_speed_command / not_zero(v->speed), -1, 1));
  double heading_ref = lane_future_heading + clipd(heading_command, -PI / 4, PI / 4);
  double heading_rate_command = KP_HEADING * wrap_to_pi(heading_ref - v->heading);
  if (g_math_mode) { /* device arithmetic: the command is an arcsine, so its sin is the argument and its cos the square root asin forms */
    const double arg = clipd(VEH_LENGTH / 2 / not_zero(v->speed) * heading_rate_command, -1, 1);
    double w;
    const double steer = mmm_asin_w(arg, &w);
    const int sat = fabs(steer) > MAX_STEER;
    v->sc_sin = sat ? (arg < 0 ? -MMM_SIN_PI3 : MMM_SIN_PI3) : arg;
    v->sc_cos = sat ? MMM_COS_PI3 : w;
    v->sc_steer = clipd(steer, -MAX_STEER, MAX_STEER);
    v->sc_valid = 1;
    return v->sc_steer;
  }
  double steering_angle =
      m_asin(clipd(VEH_LENGTH / 2 / not_zero(v->speed) * heading_rate_command, -1, 1));
  return clipd(steering_angle, -MAX_STEER, MAX_STEER);
}
/* safe_controller.py:84-98 MDPLCVehicle.steering_control: in "steer_vel" mode the command is a
 * steering VELOCITY tracking a scaled-down reference angle (KP_STEER 20, STEER_TARGET_RF 0.125) */
static double lc_steering_control(Veh *v, int target_lane) {
  double steering_ref = steering_control(v, target_lane);
  if (v->steer_vel) {
    steering_ref = steering_ref * 0.125;
    return 20 * (steering_ref - v->steer_angle);
  }
  return steering_ref;
}
/* controller.py:189-197 */
static double speed_control(const Veh *v, double target_speed) {
  return (1 / TAU_A) * (target_speed - v->speed);
}
/* controller.py:90-134 ControlledVehicle.act; action: 0 LEFT 1 IDLE 2 RIGHT, -1 None */
static void controlled_act(Veh *v, int action) {
  follow_road(v);
  if (action == 2 || action == 0) {
    int road = lane_road(v->target_lane), id = lane_id_in_road(v->target_lane);
    int nl = (road == 1) ? 2 : 1;
    int nid = id + (action == 2 ? 1 : -1);
    if (nid < 0) nid = 0;
    if (nid > nl - 1) nid = nl - 1;
    int cand = (road == 1) ? (nid == 1 ? MM_LANE_BC1 : MM_LANE_BC0) : v->target_lane;
    if (lane_is_reachable_from(cand, v->x, v->y)) v->target_lane = cand;
  }
  double steer = lc_steering_control(v, v->target_lane);
  v->act_acc = speed_control(v, v->target_speed);
  v->act_steer = clipd(steer, -PI / 3, PI / 3);
}
/* controller.py:293-311 MDPVehicle.act (+ safe_controller.py:63-66 hl_action record) */
static void mdp_act(Veh *v, int action, int is_lc) {
  if (is_lc && action >= 0) v->hl_action = action;
  if (action == 3) v->speed_index = speed_to_index(v->speed) + 1;
  else if (action == 4) v->speed_index = speed_to_index(v->speed) - 1;
  else { controlled_act(v, action); return; }
  v->speed_index = (int)clipd(v->speed_index, 0, 5 - 1);
  v->target_speed = index_to_speed(v->speed_index);
  controlled_act(v, -1);
}
/* controller.py:257-267 get_corner; dir 0 = "L", 1 = "R" */
static void get_corner(const Veh *v, int dir, double *cx, double *cy) {
  const double CORNER_LEN = sqrt((VEH_WIDTH / 2) * (VEH_WIDTH / 2) + (VEH_LENGTH / 2) * (VEH_LENGTH / 2)) + 0.0075;
  const double CORNER_ALPHA = m_atan(VEH_WIDTH / VEH_LENGTH);
  if (g_math_mode) { /* device arithmetic: angle sums from sin / cos of the heading (include/mm_math.h) */
    double sh, ch;
    mmm_sincos(v->heading, &sh, &ch);
    *cx = v->x + (CORNER_LEN * mmm_cos_sum(MMM_CORNER_SIN, MMM_CORNER_COS, sh, ch));
    if (dir == 0) *cy = v->y - (CORNER_LEN * mmm_sin_sum(MMM_CORNER_SIN, MMM_CORNER_COS, sh, ch)) + 0.01;
    else *cy = v->y - (CORNER_LEN * mmm_sin_sum(-MMM_CORNER_SIN, MMM_CORNER_COS, sh, ch)) + 0.01;
    return;
  }
  *cx = v->x + (CORNER_LEN * m_cos(CORNER_ALPHA + v->heading));
  if (dir == 0) *cy = v->y - (CORNER_LEN * m_sin(CORNER_ALPHA + v->heading)) + 0.01;
  else *cy = v->y - (CORNER_LEN * m_sin(-CORNER_ALPHA + v->heading)) + 0.01;
}


/* ------------------------------------------------------------------ vehicle/behavior.py (HDVs) */
#define IDM_ACC_MAX 6.0          /* behavior.py:24 */
#define IDM_COMFORT_ACC_MAX 3.0  /* :26 */
#define IDM_COMFORT_ACC_MIN (-5.0)
#define IDM_DISTANCE_WANTED (5.0 + VEH_LENGTH)
#define IDM_TIME_WANTED 1.5
#define MOBIL_MIN_ACC_GAIN 0.1
#define MOBIL_MAX_BRAKING 9.0
#define MOBIL_DELAY 1.0

typedef struct { int present; int is_object; double x, y, heading, speed, target_speed; int lane; } Body;

static Body body_of(const Veh *v) {
  Body b = {1, 0, v->x, v->y, v->heading, v->speed, v->target_speed, v->lane};
  return b;
}
static Body body_obstacle(void) {
  Body b = {1, 1, OBST_X, OBST_Y, 0.0, 0.0, 0.0, -1};
  return b;
}
/* road.py:352-381 neighbour_vehicles on `lane`: front / rear among vehicles then objects */
static void neighbour_vehicles(const Env *e, int i, int lane, Body *front, Body *rear) {
  double s, r, s_front = 0, s_rear = 0;
  lane_local(lane, e->v[i].x, e->v[i].y, &s, &r);
  front->present = rear->present = 0;
  for (int j = 0; j <= e->n; j++) {
    Body b;
    if (j < e->n) { if (j == i) continue; b = body_of(&e->v[j]); }
    else b = body_obstacle();
    double s_v, lat_v;
    lane_local(lane, b.x, b.y, &s_v, &lat_v);
    /* lane.on_lane(position, s_v, lat_v, margin=1), lane.py:61-76 */
    if (!(fabs(lat_v) <= LANE_WIDTH / 2 + 1 && (-VEH_LENGTH <= s_v && s_v < LANE_LEN[lane] + VEH_LENGTH))) continue;
    if (s <= s_v && (!front->present || s_v <= s_front)) { s_front = s_v; *front = b; }
    if (s_v < s && (!rear->present || s_v > s_rear)) { s_rear = s_v; *rear = b; }
  }
}
/* behavior.py:141-156 desired_gap (projected) */
static double desired_gap(const Body *ego, const Body *front) {
  const double ab = -IDM_COMFORT_ACC_MAX * IDM_COMFORT_ACC_MIN;
  double ec = m_cos(ego->heading), es = m_sin(ego->heading);
  double evx = ego->speed * ec, evy = ego->speed * es;
  double fvx = front->speed * m_cos(front->heading), fvy = front->speed * m_sin(front->heading);
  double dv = (evx - fvx) * ec + (evy - fvy) * es;
  return IDM_DISTANCE_WANTED + ego->speed * IDM_TIME_WANTED + ego->speed * dv / (2 * sqrt(ab));
}
/* behavior.py:111-139 acceleration (IDM); ego / front may be absent, ego may be an object */
static double idm_acceleration(const Body *ego, const Body *front) {
  if (!ego->present || ego->is_object) return 0;
  double ego_target_speed = not_zero(ego->target_speed);
  double acceleration = IDM_COMFORT_ACC_MAX * (1 - m_p4(fmax(ego->speed, 0) / ego_target_speed));
  if (front->present) {
    double d = (front->x - LANE_SX[ego->lane]) - (ego->x - LANE_SX[ego->lane]); /* ego.lane_distance_to(front) */
    acceleration -= IDM_COMFORT_ACC_MAX * m_sq(desired_gap(ego, front) / not_zero(d));
  }
  return acceleration;
}
/* behavior.py:225-266 mobil (no route; POLITENESS = 0 removes the followers' terms) */
static int mobil(const Env *e, int i, int lane_index) {
  Body self = body_of(&e->v[i]), new_prec, new_foll, old_prec, old_foll;
  neighbour_vehicles(e, i, lane_index, &new_prec, &new_foll);
  double new_following_pred_a = idm_acceleration(&new_foll, &self);
  if (new_following_pred_a < -MOBIL_MAX_BRAKING) return 0;
  neighbour_vehicles(e, i, e->v[i].lane, &old_prec, &old_foll);
  double self_pred_a = idm_acceleration(&self, &new_prec);
  double self_a = idm_acceleration(&self, &old_prec);
  double jerk = self_pred_a - self_a + 0.0;
  if (jerk < MOBIL_MIN_ACC_GAIN) return 0;
  return 1;
}
/* behavior.py:183-223 change_lane_policy */
static void change_lane_policy(Env *e, int i) {
  Veh *v = &e->v[i];
  if (v->lane != v->target_lane) {
    if (lane_road(v->lane) == lane_road(v->target_lane)) {
      Body self = body_of(v);
      for (int j = 0; j < e->n; j++) {
        const Veh *o = &e->v[j];
        if (j != i && o->lane != v->target_lane && o->target_lane == v->target_lane) {
          double d = (o->x - LANE_SX[v->lane]) - (v->x - LANE_SX[v->lane]);
          Body ob = body_of(o);
          double d_star = desired_gap(&self, &ob);
          if (0 < d && d < d_star) { v->target_lane = v->lane; break; }
        }
      }
    }
    return;
  }
  if (!(MOBIL_DELAY < v->timer)) return; /* utils.do_every */
  v->timer = 0;
  /* side_lanes (road.py:147-158): only road (b,c) has two lanes */
  int side = v->lane == MM_LANE_BC0 ? MM_LANE_BC1 : (v->lane == MM_LANE_BC1 ? MM_LANE_BC0 : -1);
  if (side < 0) return;
  if (!lane_is_reachable_from(side, v->x, v->y)) return;
  if (mobil(e, i, side)) v->target_lane = side;
}
/* behavior.py:74-100 IDMVehicle.act */
static void idm_act(Env *e, int i) {
  Veh *v = &e->v[i];
  if (v->crashed) return;
  Body front, rear, self;
  neighbour_vehicles(e, i, v->lane, &front, &rear);
  follow_road(v);
  change_lane_policy(e, i);
  double steer = clipd(steering_control(v, v->target_lane), -PI / 3, PI / 3);
  self = body_of(v);
  double acc = clipd(idm_acceleration(&self, &front), -IDM_ACC_MAX, IDM_ACC_MAX);
  v->act_steer = steer;
  v->act_acc = acc;
}

/* ------------------------------------------------------------------ safety/decentral_layer.py */

/* decentral_layer.py:15-20 */
static int is_same_lane(const Veh *v, int lane2) {
  int nl = next_lane(v->lane, v->x, v->y);
  return v->lane == lane2 || lane2 == nl;
}
/* decentral_layer.py:23-39 */
static int is_adj_lane(const Veh *v, int lane2) {
  int l1 = v->lane, nl = next_lane(v->lane, v->x, v->y);
  if (lane_road(l1) == lane_road(lane2) && abs(lane_id_in_road(l1) - lane_id_in_road(lane2)) == 1)
    return lane_id_in_road(l1) - lane_id_in_road(lane2);
  else if (lane_road(nl) == lane_road(lane2) &&
           abs(lane_id_in_road(nl) - lane_id_in_road(lane2)) == 1)
    return lane_id_in_road(nl) - lane_id_in_road(lane2);
  return 0;
}
/* kinematics.py:161-173 lane_distance_to in self.lane's frame */
static double lane_distance_to(const Veh *self, const Veh *other) {
  return (other->x - LANE_SX[self->lane]) - (self->x - LANE_SX[self->lane]);
}
/* decentral_layer.py:46-57 */
static int is_approaching_same_lane(const Veh *ve, const Veh *vl) {
  if (lane_distance_to(ve, vl) < 0) return 0;
  double y_dist = vl->y - ve->y;
  int dist_cond = fabs(y_dist) <= 3.5;
  int heading_cond = (y_dist < 0) ? (vl->heading > 0.037) : (vl->heading < -0.037);
  return dist_cond && heading_cond;
}

typedef struct { int present; double x, heading, vx, speed; } NState; /* dict subset the shield reads */

/* decentral_layer.py:60-77 simplified_control */
static void simplified_control(const NState *s, double acc, double steer, double vl, double dt,
                               double *v_out, double *dpsi_out) {
  if (!s->present) { *v_out = 0; *dpsi_out = 0; return; }
  double speed = s->speed;
  double v = s->vx + acc * dt;
  v = v > 0 ? v : 0; /* max(0, v) */
  double beta = m_atan(0.5 * m_tan(steer));
  *dpsi_out = (speed / vl * m_sin(beta)) + s->heading;
  *v_out = v;
}

/* road.py:257-267 close_vehicles_to: indices of <= count nearest others, sorted by |lane distance| (stable) */
static int close_vehicles_to(const Env *e, int i, double distance, int count, int *out) {
  int idx[MM_MAX_AGENTS], m = 0;
  double key[MM_MAX_AGENTS];
  const Veh *me = &e->v[i];
  for (int j = 0; j < e->n; j++) {
    if (j == i) continue;
    double dx = e->v[j].x - me->x, dy = e->v[j].y - me->y;
    if (sqrt(dx * dx + dy * dy) < distance) {
      idx[m] = j;
      key[m] = fabs(lane_distance_to(me, &e->v[j]));
      m++;
    }
  }
  for (int a = 1; a < m; a++) { /* stable insertion sort */
    int ja = idx[a]; double ka = key[a]; int b = a - 1;
    while (b >= 0 && key[b] > ka) { idx[b + 1] = idx[b]; key[b + 1] = key[b]; b--; }
    idx[b + 1] = ja; key[b + 1] = ka;
  }
  if (m > count) m = count;
  for (int a = 0; a < m; a++) out[a] = idx[a];
  return m;
}

/*
 * cvxopt.solvers.qp (cbf.py:134) -> coneqp for dense P (diagonal here, cbf.py:40-44), q, G, h, no equality
 * constraints, default options: a C restatement of the published interior-point algorithm (cvxopt
 * documentation, "Cone Programming" / coneqp; Vandenberghe 2010) specialised to the 'l' cone, following
 * the order of operations of tools/refshim/cvxopt/coneqp.py (the reference-side stand-in that made the
 * ipm_* fixtures) line by line, so the two agree bit for bit.  cvxopt 1.2.7 itself is not installable
 * in this image: "parity unpinned" against the real binary (BLAS/LAPACK rounding), pinned against the
 * stand-in.  n = 3 unknowns, m = 3 | 4 rows.  Returns 1 "optimal" / 0 "unknown"; *iters = iterations.
 */
#define QP_N 3
static double qp_dot(const double *a, const double *b, int n) {
  double t = 0.0;
  for (int i = 0; i < n; i++) t += a[i] * b[i];
  return t;
}
typedef struct { double Gs[4][QP_N], L[QP_N][QP_N], di[4]; int m; } QpKkt;
static int qp_factor(QpKkt *f, const double *Pd, const double G[4][QP_N], int m, const double *di) { /* misc.kkt_chol2 */
  double S[QP_N][QP_N];
  f->m = m;
  for (int k = 0; k < m; k++) { f->di[k] = di[k]; for (int j = 0; j < QP_N; j++) f->Gs[k][j] = di[k] * G[k][j]; }
  for (int i = 0; i < QP_N; i++) {
    for (int j = 0; j <= i; j++) {
      double t = 0.0;
      for (int k = 0; k < m; k++) t += f->Gs[k][i] * f->Gs[k][j];
      S[i][j] = t;
    }
    S[i][i] = S[i][i] + Pd[i];
  }
  for (int j = 0; j < QP_N; j++) {
    double t = S[j][j];
    for (int k = 0; k < j; k++) t -= f->L[j][k] * f->L[j][k];
    if (!(t > 0.0)) return 0; /* potrf: ArithmeticError */
    f->L[j][j] = sqrt(t);
    for (int i = j + 1; i < QP_N; i++) {
      t = S[i][j];
      for (int k = 0; k < j; k++) t -= f->L[i][k] * f->L[j][k];
      f->L[i][j] = t / f->L[j][j];
    }
  }
  return 1;
}
static void qp_solve(const QpKkt *f, double *x, double *z) {
  const int m = f->m;
  for (int k = 0; k < m; k++) z[k] = z[k] * f->di[k];
  for (int j = 0; j < QP_N; j++) {
    double t = 0.0;
    for (int k = 0; k < m; k++) t += f->Gs[k][j] * z[k];
    x[j] = x[j] + t;
  }
  for (int j = 0; j < QP_N; j++) {
    x[j] = x[j] / f->L[j][j];
    for (int i = j + 1; i < QP_N; i++) x[i] = x[i] - x[j] * f->L[i][j];
  }
  for (int j = QP_N - 1; j >= 0; j--) {
    double t = x[j];
    for (int i = QP_N - 1; i > j; i--) t -= f->L[i][j] * x[i];
    x[j] = t / f->L[j][j];
  }
  for (int k = 0; k < m; k++) {
    double t = -z[k];
    for (int j = 0; j < QP_N; j++) t += x[j] * f->Gs[k][j];
    z[k] = t;
  }
}
static double qp_maxneg(const double *a, int m) { /* misc.max_step, 'l' block */
  double t = -a[0];
  for (int k = 1; k < m; k++) if (-a[k] > t) t = -a[k];
  return t;
}
static int qp_ipm(const double *Pd, const double *q, const double G[4][QP_N], const double *h, int m, double *x,
                  int *iters_out) {
  QpKkt kkt;
  double s[4], z[4], d[4], di[4], lmbda[4], lmbdasq[4], rx[QP_N], rz[4], dx[QP_N], dz[4], ds[4], ws3[4];
  const double resx0 = fmax(1.0, sqrt(qp_dot(q, q, QP_N))), resz0 = fmax(1.0, sqrt(qp_dot(h, h, m)));
  for (int k = 0; k < m; k++) di[k] = 1.0;
  if (!qp_factor(&kkt, Pd, G, m, di)) { x[0] = x[1] = x[2] = NAN; *iters_out = 0; return 0; } /* ValueError("Rank...") */
  for (int j = 0; j < QP_N; j++) x[j] = -q[j];
  for (int k = 0; k < m; k++) z[k] = h[k];
  qp_solve(&kkt, x, z);
  for (int k = 0; k < m; k++) s[k] = -z[k];
  {
    double nrm = sqrt(qp_dot(s, s, m)), ts = qp_maxneg(s, m);
    if (ts >= -1e-8 * fmax(nrm, 1.0)) { const double a = 1.0 + ts; for (int k = 0; k < m; k++) s[k] = s[k] + a; }
    nrm = sqrt(qp_dot(z, z, m));
    double tz = qp_maxneg(z, m);
    if (tz >= -1e-8 * fmax(nrm, 1.0)) { const double a = 1.0 + tz; for (int k = 0; k < m; k++) z[k] = z[k] + a; }
  }
  double gap = qp_dot(s, z, m);
  for (int iters = 0; iters <= 100; iters++) {
    for (int j = 0; j < QP_N; j++) rx[j] = q[j] + Pd[j] * x[j];
    const double f0 = 0.5 * (qp_dot(x, rx, QP_N) + qp_dot(x, q, QP_N));
    for (int j = 0; j < QP_N; j++) {
      double t = 0.0;
      for (int k = 0; k < m; k++) t += G[k][j] * z[k];
      rx[j] = rx[j] + t;
    }
    const double resx = sqrt(qp_dot(rx, rx, QP_N));
    for (int k = 0; k < m; k++) {
      double t = s[k] - h[k];
      for (int j = 0; j < QP_N; j++) t += x[j] * G[k][j];
      rz[k] = t;
    }
    const double resz = sqrt(qp_dot(rz, rz, m));
    const double pcost = f0, dcost = f0 + qp_dot(z, rz, m) - gap;
    int have_rel = 0;
    double relgap = 0;
    if (pcost < 0.0) { relgap = gap / -pcost; have_rel = 1; }
    else if (dcost > 0.0) { relgap = gap / dcost; have_rel = 1; }
    const double pres = resz / resz0, dres = resx / resx0;
    if ((pres <= 1e-7 && dres <= 1e-7 && (gap <= 1e-7 || (have_rel && relgap <= 1e-6))) || iters == 100) {
      *iters_out = iters;
      return iters == 100 ? 0 : 1;
    }
    if (iters == 0)
      for (int k = 0; k < m; k++) { d[k] = sqrt(s[k] / z[k]); di[k] = 1.0 / d[k]; lmbda[k] = sqrt(s[k] * z[k]); }
    for (int k = 0; k < m; k++) lmbdasq[k] = lmbda[k] * lmbda[k];
    if (!qp_factor(&kkt, Pd, G, m, di)) { *iters_out = iters; return 0; } /* "Terminated (singular KKT matrix)" */
    const double mu = gap / m;
    double sigma = 0.0, eta = 0.0, step = 1.0;
    for (int i = 0; i < 2; i++) {
      for (int k = 0; k < m; k++) {
        double t = 0.0;
        if (i == 1) t = t - ws3[k];
        t = t - lmbdasq[k];
        ds[k] = t + sigma * mu;
      }
      for (int j = 0; j < QP_N; j++) dx[j] = (-1.0 + eta) * rx[j];
      for (int k = 0; k < m; k++) dz[k] = (-1.0 + eta) * rz[k];
      for (int k = 0; k < m; k++) { ds[k] = ds[k] / lmbda[k]; dz[k] = dz[k] - d[k] * ds[k]; } /* f4_no_ir */
      qp_solve(&kkt, dx, dz);
      for (int k = 0; k < m; k++) ds[k] = ds[k] - dz[k];
      const double dsdz = qp_dot(ds, dz, m);
      if (i == 0) for (int k = 0; k < m; k++) ws3[k] = ds[k] * dz[k];
      for (int k = 0; k < m; k++) { ds[k] = ds[k] / lmbda[k]; dz[k] = dz[k] / lmbda[k]; }
      const double ts = qp_maxneg(ds, m), tz = qp_maxneg(dz, m);
      const double t = fmax(0.0, fmax(ts, tz));
      if (t == 0) step = 1.0;
      else if (i == 0) step = fmin(1.0, 1.0 / t);
      else step = fmin(1.0, 0.99 / t);
      if (i == 0) {
        const double sg = fmin(1.0, fmax(0.0, 1.0 - step + dsdz / gap * (step * step)));
        sigma = sg * sg * sg;
        eta = 0.0;
      }
    }
    for (int j = 0; j < QP_N; j++) x[j] = x[j] + step * dx[j];
    for (int k = 0; k < m; k++) { /* updated iterates in the current scaling + misc.update_scaling */
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
    gap = qp_dot(lmbda, lmbda, m);
  }
  return 0; /* not reached */
}
/* the shield's QP through the IPM: P = diag(1, 1, 1e18), q = 0, rows [a 0 -1; 1 0 0; -1 0 0; (a 0 -1)] (cbf.py:288-322,386-403) */
static int qp_ipm_cbf(double a, const double *hh, int rows, double *d_out, int *iters) {
  const double Pd[3] = {1.0, 1.0, 1e18}, q[3] = {0.0, 0.0, 0.0};
  const double G[4][QP_N] = {{a, 0.0, -1.0}, {1.0, 0.0, 0.0}, {-1.0, 0.0, 0.0}, {a, 0.0, -1.0}};
  double x[3];
  const int ok = qp_ipm(Pd, q, G, hh, rows, x, iters);
  *d_out = x[0];
  return ok;
}

/*
 * safety_layer -> safe_action_hss / safe_action_mass (decentral_layer.py:290-518, :521-764,
 * :767-817) with multi_agent_state (:85-257) and the CBF rows of cbf.py:262-430.  The QP
 * (cbf.py:134, cvxopt) is solved by its exact KKT closed form.
 * Returns 0, or MM_ERR_QP_BOUNDS when check_bounds (cbf.py:87-96) would raise.
 */
static int safety_layer(const MMConfig *cfg, Env *e, int i, double dt, double *safe_acc,
                        double *safe_steer) {
  Veh *veh = &e->v[i];
  const int mass = cfg->shield == MM_SHIELD_MASS;
  const double eta = cfg->cbf_eta, TAU = cfg->cbf_tau;
  double v_min = veh->speed + LC_MIN_ACC * dt;
  if (mass) v_min = v_min > 0 ? v_min : 0; /* max(0, v_min), :798 */
  double v_max = veh->speed + LC_MAX_ACC * dt;

  /* s_e = vehicle.to_dict() (+ speed), vx floored at 1 (:307-309) */
  NState s_e = {1, veh->x, veh->heading, veh->speed * m_cos(veh->heading), veh->speed};
  s_e.vx = s_e.vx > 1 ? s_e.vx : 1;
  double sf_ol[2] = {s_e.x + PERCEPTION_DIST + 1, 0.0};
  double sf_oa[2] = {s_e.x + PERCEPTION_DIST + 1, 0.0};
  double sf_oar[2] = {s_e.x - PERCEPTION_DIST - 1, 0.0};

  /* ---- multi_agent_state (:85-257) */
  NState s_ol = {0}, s_oa = {0}, s_oar = {0};
  double a_ol_acc = 0, a_ol_steer = 0, a_oa_acc = 0, a_oa_steer = 0;
  double gp_ol = 0, gp_oa = 0;
  int constrain_adj = 0;
  int near[MM_MAX_AGENTS];
  int m = close_vehicles_to(e, i, PERCEPTION_DIST, 5, near);
#ifdef ORC_QP_LOG
  g_qplog_dep_ol = g_qplog_dep_oa = -1;
#endif
  for (int k = 0; k < m; k++) {
    Veh *o = &e->v[near[k]];
    int v_a = is_adj_lane(veh, o->lane);
    int a_v = is_adj_lane(o, veh->lane);
    int appr = is_approaching_same_lane(veh, o);
    double ld = lane_distance_to(veh, o);
    if (!appr && (v_a || a_v)) {
      if (!s_oar.present && ld < 0) {
        s_oar.present = 1; /* veh.to_dict(): current state */
        s_oar.x = o->x; s_oar.heading = o->heading;
        s_oar.vx = o->speed * m_cos(o->heading); s_oar.speed = o->speed;
      } else if (!s_oa.present && ld >= 0) {
        s_oa.present = 1; /* veh.state_hist[-2] */
        s_oa.x = o->h2[0]; s_oa.vx = o->h2[1]; /* heading / speed only feed the unused dpsi term */
#ifdef ORC_QP_LOG
        if (mass && o->kind == 1 && ((g_qplog_stepped >> near[k]) & 1u)) g_qplog_dep_oa = near[k];
#endif
        if (mass) {
          if (o->kind == 1) { a_oa_acc = o->safe_acc; a_oa_steer = o->safe_steer; gp_oa = o->g_vx; }
          else { a_oa_acc = CBF_ACC_LO; a_oa_steer = 0; gp_oa = 1; } /* HDV: no safe_action / fg_params (:129-135) */
          /* :138-160: the corner test always overrides the collaborate_adj expression */
          double cx, cy;
          if (v_a == -1 || a_v == 1) get_corner(o, 0, &cx, &cy);
          else get_corner(o, 1, &cx, &cy);
          constrain_adj = !lane_on_lane(o->lane, cx, cy);
        }
      }
    } else if (o->kind == 2 && veh->lane == MM_LANE_AB0 && o->lane == MM_LANE_KB0 && ld >= 0) {
      /* :162-184 adjacent HDV on the merging lane: constrain against its "digital twin" half a second
       * of ego travel ahead.  The reference edits the HDV's history record IN PLACE, so later egos of
       * the same sub-step read the shifted x as well -- reproduced by writing o->h2[0]. */
      o->h2[0] = o->h2[0] + 0.5 * (veh->speed * m_cos(veh->heading));
      s_oa.present = 1; s_oa.x = o->h2[0]; s_oa.vx = o->h2[1];
      constrain_adj = 1;
      a_oa_acc = CBF_ACC_LO; a_oa_steer = 0; gp_oa = 1;
    } else if (!s_ol.present && (is_same_lane(veh, o->lane) || appr) && ld > 0) {
      s_ol.present = 1;
      s_ol.x = o->h2[0]; s_ol.vx = o->h2[1];
#ifdef ORC_QP_LOG
      if (mass && o->kind == 1 && ((g_qplog_stepped >> near[k]) & 1u)) g_qplog_dep_ol = near[k];
#endif
      if (mass) {
        if (o->kind == 1) { a_ol_acc = o->safe_acc; a_ol_steer = o->safe_steer; gp_ol = o->g_vx; }
        else { a_ol_acc = CBF_ACC_LO; a_ol_steer = 0; gp_ol = 1; }
      }
    }
  }
  /* obstacles (:213-246): one Obstacle at (420, 4) */
  if (!(veh->x > OBST_X)) {
    if ((!s_ol.present || OBST_X <= s_ol.x) && fabs(OBST_Y - veh->y) <= 2) {
      s_ol.present = 1; s_ol.x = OBST_X; s_ol.heading = 0; s_ol.vx = 0.0;
      s_ol.speed = 0.0 / m_cos(0.0); /* "cos_h" branch of simplified_control */
#ifdef ORC_QP_LOG
      g_qplog_dep_ol = -1;
#endif
      if (mass) { a_ol_acc = 0; a_ol_steer = 0; gp_ol = 0; }
    }
    double ady = fabs(OBST_Y - veh->y);
    if ((!s_oa.present || OBST_X <= s_oa.x) && (2 < ady && ady <= 4)) {
      s_oa.present = 1; s_oa.x = OBST_X; s_oa.heading = 0; s_oa.vx = 0.0;
      s_oa.speed = 0.0 / m_cos(0.0);
#ifdef ORC_QP_LOG
      g_qplog_dep_oa = -1;
#endif
      if (mass) { a_oa_acc = 0; a_oa_steer = 0; gp_oa = 0; constrain_adj = 0; }
    }
  }
  if (s_oa.present) { sf_oa[0] = s_oa.x; sf_oa[1] = s_oa.heading; }
  if (s_ol.present) { sf_ol[0] = s_ol.x; sf_ol[1] = s_ol.heading; }
  if (s_oar.present) { sf_oar[0] = s_oar.x; sf_oar[1] = s_oar.heading; }

  /* ---- g (diagonal 8x8 after the transpose, :392-442 / :626-676), x, f = x */
  double g[8];
  g[0] = veh->g_vx * dt; g[1] = 1 * dt;
  g[2] = mass ? gp_ol * dt : 1 * dt; g[3] = 1 * dt;
  g[4] = mass ? gp_oa * dt : 1 * dt; g[5] = 1 * dt;
  g[6] = 1 * dt; g[7] = 1 * dt;
  double x[8] = {s_e.x, s_e.heading, sf_ol[0], sf_ol[1], sf_oa[0], sf_oa[1], sf_oar[0], sf_oar[1]};

  /* ---- safe distances (:448-470) */
  double sv_oar = s_oar.present ? s_oar.vx : 0;
  sv_oar = sv_oar + CBF_ACC_HI * dt;
  sv_oar = sv_oar > 1 ? sv_oar : 1;
  double buffer = (CBF_ACC_HI + 0.1) * dt * TAU;
  double sd0 = s_e.vx * TAU + VEH_LENGTH + buffer;
  veh->shield_headway = (sf_ol[0] - s_e.x - VEH_LENGTH) / s_e.vx; /* set_min_headway :466 / :700 */
  double sd1 = sd0;
  double sd2 = sv_oar * TAU + VEH_LENGTH + buffer;

  /* ---- predicted inputs (:473-488 HSS worst case, :706-714 MASS decided actions) */
  double u[8];
  simplified_control(&s_e, veh->act_acc, veh->act_steer, VEH_LENGTH, dt, &u[0], &u[1]);
  if (mass) {
    simplified_control(&s_ol, a_ol_acc, a_ol_steer, VEH_LENGTH, dt, &u[2], &u[3]);
    simplified_control(&s_oa, a_oa_acc, a_oa_steer, VEH_LENGTH, dt, &u[4], &u[5]);
  } else {
    simplified_control(&s_ol, CBF_ACC_LO, 0, VEH_LENGTH, dt, &u[2], &u[3]);
    simplified_control(&s_oa, CBF_ACC_LO, 0, VEH_LENGTH, dt, &u[4], &u[5]);
  }
  simplified_control(&s_oar, CBF_ACC_HI, 0, VEH_LENGTH, dt, &u[6], &u[7]);

  /* ---- control_barrier (cbf.py:110-161): define_pq, get_G, get_h */
  double q_lon = -VEH_LENGTH - sd0;
  double q_lona = -VEH_LENGTH - sd1;
  double q_lonr = -VEH_LENGTH - sd2;
  if (mass && constrain_adj) q_lona = -VEH_LENGTH - sd1 - ADJ_BUFFER; /* cbf.py:380-384 (CBF_CAV only) */
  double a = g[0]; /* G[0] = [g_e.vx*dt, 0, -1] */
  double px_lon = x[2] - x[0], px_lona = x[4] - x[0], px_lonr = x[0] - x[6];
  double h0 = px_lon + (eta - 1) * px_lon + eta * q_lon + (-(g[0] * u[0]) + g[2] * u[2]);
  double h1 = v_max - u[0];
  double h2 = -v_min + u[0];
  int rows = 3;
  double h3 = NAN, hc = h0;
  if (mass && constrain_adj) {
    h3 = px_lona + (eta - 1) * px_lona + eta * q_lona + (-(g[0] * u[0]) + g[4] * u[4]);
    rows = 4;
    hc = h3 < h0 ? h3 : h0;
  }
  double hi = h1, lo = -h2, d;
  veh->qp_optimal = 1;
  if (cfg->qp_solver == MM_QP_IPM) { /* the iterate cvxopt's coneqp stops at (qp_ipm above) */
    const double hh[4] = {h0, h1, h2, h3};
    int iters;
    veh->qp_optimal = qp_ipm_cbf(a, hh, rows, &d, &iters);
#ifdef ORC_QP_LOG
    if (g_qplog)
      fprintf(g_qplog, "%d %d %d %d %d %d %d %d %d\n", g_qplog_env, e->time, i, g_qplog_rank, iters, veh->qp_optimal, rows,
              g_qplog_dep_ol, rows == 4 ? g_qplog_dep_oa : -1);
#endif
  } else { /* exact KKT solution of min 1/2(d^2 + e^2 + 1e18 s^2): see tools/refshim/cvxopt */
    if (a > 0) d = fmin(0.0, hc / a);
    else if (a < 0) d = fmax(0.0, hc / a);
    else d = 0.0;
    d = fmin(fmax(d, lo), hi);
  }
  double u_safe0 = u[0] + d, u_safe1;
  veh->qp_rows = rows; veh->qp_a = a; veh->qp_h[0] = h0; veh->qp_h[1] = h1; veh->qp_h[2] = h2;
  veh->qp_h[3] = h3; veh->qp_d = d;
  int rc = 0;
  veh->qp_bounds = (u_safe0 - 0.001 > v_max || u_safe0 + 0.001 < v_min); /* cbf.py:87-96 */
  if (veh->qp_bounds) rc = MM_ERR_QP_BOUNDS;

  { /* update_status (cbf.py:341-351) with u_status = [u_safe (QP), u_ll[2:]] */
    double hls_lon = px_lon + q_lon;
    double hlds_lon = px_lon + ((-g[0]) * u_safe0 + g[2] * u[2]) + q_lon;
    veh->lon_safe = hls_lon >= -1e-6;
    veh->lon_invariant = (hlds_lon + (eta - 1) * hls_lon) >= -1e-6;
  }
  u_safe1 = veh->act_steer; /* :493 / :721 lateral control is not constrained */
  double um[8] = {u_safe0, u_safe1, u[2], u[3], u[4], u[5], u[6], u[7]}; /* u_safe_ma */
  int flags = veh->flags & MM_FLAG_COLLABORATE_ADJ;
  if (constrain_adj) flags |= MM_FLAG_IS_COLLABORATING;
  flags |= MM_FLAG_IS_LC_SAFE;

  /* is_lc_allowed (cbf.py:324-339) */
  double hls_lona = px_lona + q_lona;
  double hlds_lona = px_lona + ((-g[0]) * um[0] + g[4] * um[4]) + q_lona;
  double hls_lonr = px_lonr + q_lonr;
  double hlds_lonr = px_lonr + (g[0] * um[0] + (-g[6]) * um[6]) + q_lonr;
  int lc_allowed = ((hls_lona >= 0) && (hlds_lona + (eta - 1) * hls_lona) >= 0) &&
                   ((hls_lonr >= 0) && (hlds_lonr + (eta - 1) * hls_lonr) >= 0);
  veh->lc_margin = fmin(fmin(hls_lona, hlds_lona + (eta - 1) * hls_lona),
                        fmin(hls_lonr, hlds_lonr + (eta - 1) * hls_lonr));
  if (!mass) {
    if (!lc_allowed) { /* :501-506 */
      veh->target_lane = veh->lane;
      u_safe1 = lc_steering_control(veh, veh->target_lane);
      flags &= ~MM_FLAG_IS_LC_SAFE;
    }
  } else {
    double cx, cy;
    int can_abort_lc = 1; /* :728-736: both front corners still on the current lane */
    get_corner(veh, 0, &cx, &cy);
    can_abort_lc = can_abort_lc && lane_on_lane(veh->lane, cx, cy);
    get_corner(veh, 1, &cx, &cy);
    can_abort_lc = can_abort_lc && lane_on_lane(veh->lane, cx, cy);
    if (can_abort_lc && !lc_allowed) { /* :739-744 */
      veh->target_lane = veh->lane;
      u_safe1 = lc_steering_control(veh, veh->target_lane);
      flags &= ~MM_FLAG_IS_LC_SAFE;
    } else if ((veh->hl_action == 2 || veh->hl_action == 0) && veh->speed < STOPPING_SPEED) {
      u_safe0 = u[0]; /* :746-750 */
    }
    /* can_collaborate_adj (cbf.py:424-430) on u_safe_ma (copied before the edits above) */
    double inv = hlds_lona + (eta - 1) * hls_lona;
    if (inv >= -1e-6) flags |= MM_FLAG_COLLABORATE_ADJ; else flags &= ~MM_FLAG_COLLABORATE_ADJ;
  }
  veh->flags = flags;
  *safe_acc = (u_safe0 - s_e.vx) / dt; /* derived_acceleration :80-82 */
  *safe_steer = u_safe1;
  return rc;
}

/* ------------------------------------------------------------------ vehicle step */

/* kinematics.py:143-152 (+ safe_controller.py:100-104 for MDPLCVehicle) */
static void clip_actions(Veh *v, int is_lc) {
  if (v->crashed) { v->act_steer = 0; v->act_acc = -1.0 * v->speed; }
  if (v->speed > MAX_SPEED) v->act_acc = fmin(v->act_acc, 1.0 * (MAX_SPEED - v->speed));
  else if (v->speed < -MAX_SPEED) v->act_acc = fmax(v->act_acc, 1.0 * (MAX_SPEED - v->speed));
  if (is_lc) v->act_acc = clipd(v->act_acc, LC_MIN_ACC, LC_MAX_ACC);
}

/* kinematics.py:122-141 Vehicle.step / safe_controller.py:106-185 MDPLCVehicle.step ("steer") */
static int vehicle_step(const MMConfig *cfg, Env *e, int i, double dt) {
  Veh *v = &e->v[i];
  const int hdv = v->kind == 2;
  const int is_lc = cfg->env_kind == MM_ENV_V1 && !hdv; /* MDPLCVehicle; HDVs use Vehicle.step */
  int rc = 0;
  if (hdv) v->timer += dt; /* IDMVehicle.step behavior.py:102-109 */
  clip_actions(v, is_lc);
  double steer = v->act_steer, acc = v->act_acc;
  v->qp_rows = 0; v->qp_a = NAN; v->qp_d = NAN; v->lc_margin = NAN;
  v->qp_h[0] = v->qp_h[1] = v->qp_h[2] = v->qp_h[3] = NAN;
  if (is_lc && cfg->shield != MM_SHIELD_NONE && v->hist_len >= 2) /* gate :232-239 */
    rc = safety_layer(cfg, e, i, dt, &acc, &steer);
  if (is_lc) { v->safe_steer = steer; v->safe_acc = acc; }
  double beta;
  if (g_math_mode) {
    /* device arithmetic (math mode 1): the same bicycle step with the eleven trigonometric calls folded into one sincos
     * of the steering angle and one of the new heading (include/mm_math.h, "angle-sum forms"); beta itself is never formed */
    const int sv = is_lc && v->steer_vel;
    double ss, cs, sb, cb, sh, ch;
    if (!sv && v->sc_valid && steer == v->sc_steer) { ss = v->sc_sin; cs = v->sc_cos; } /* steer IS the last steering_control result */
    else mmm_sincos(sv ? v->steer_angle : steer, &ss, &cs); /* crashed (0 -> (0, 1)), a persisting IDM action, steer_vel */
    mmm_slip_sincos(1.0 / 2 * (ss / cs), &sb, &cb);
    mmm_sincos(v->heading, &sh, &ch);
    double vx = v->speed * mmm_cos_sum(sh, ch, sb, cb);
    double vy = v->speed * mmm_sin_sum(sh, ch, sb, cb);
    v->x += vx * dt;
    v->y += vy * dt;
    double d_heading = v->speed * sb / (VEH_LENGTH / 2);
    v->heading += sv ? d_heading : d_heading * dt;
    v->speed += acc * dt;
    if (sv) v->steer_angle += steer * dt;
    v->speed = v->speed > 0 ? v->speed : 0;
    if (is_lc) {
      double sh2, ch2;
      mmm_sincos(v->heading, &sh2, &ch2);
      v->g_vx = mmm_cos_sum(sh2, ch2, sb, cb);
    }
    beta = 0;
  } else
  if (is_lc && v->steer_vel) { /* safe_controller.py:124-150: 2nd-order steering response */
    beta = m_atan(1.0 / 2 * m_tan(v->steer_angle));
    double vx = v->speed * m_cos(v->heading + beta);
    double vy = v->speed * m_sin(v->heading + beta);
    v->x += vx * dt;
    v->y += vy * dt;
    double d_heading = v->speed * m_sin(beta) / (VEH_LENGTH / 2);
    v->heading += d_heading; /* sic: no dt (:135) */
    v->speed += acc * dt;
    v->steer_angle += steer * dt;
  } else {
    beta = m_atan(1.0 / 2 * m_tan(steer));
    double vx = v->speed * m_cos(v->heading + beta);
    double vy = v->speed * m_sin(v->heading + beta);
    v->x += vx * dt;
    v->y += vy * dt;
    v->heading += v->speed * m_sin(beta) / (VEH_LENGTH / 2) * dt;
    v->speed += acc * dt;
  }
  if (!g_math_mode) {
    v->speed = v->speed > 0 ? v->speed : 0; /* max(0, speed) */
    if (is_lc) v->g_vx = m_cos(v->heading + beta);
  }
  v->lane = closest_lane(v->x, v->y, v->heading); /* on_state_update kinematics.py:154-159 */
  if (is_lc || (hdv && cfg->env_kind == MM_ENV_V1)) { /* log_step safe_controller.py:187-201 / behavior.py:505-521 (IDMVehicleHist) */
    memcpy(v->h2, v->h1, sizeof v->h1);
    v->h1[0] = v->x; v->h1[1] = v->speed * m_cos(v->heading);
    if (v->hist_len < 2) v->hist_len++;
  }
  return rc;
}

/* kinematics.py:202-209 _is_colliding */
static int is_colliding(const Veh *a, double ox, double oy, double ol, double ow, double oh) {
  double dx = ox - a->x, dy = oy - a->y;
  if (sqrt(dx * dx + dy * dy) > VEH_LENGTH) return 0;
  return rotated_rectangles_intersect(a->x, a->y, 0.9 * VEH_LENGTH, 0.9 * VEH_WIDTH, a->heading,
                                      ox, oy, 0.9 * ol, 0.9 * ow, oh);
}

static void sort_by_x_desc(const Env *e, int *order) { /* sorted(key=x, reverse=True): stable */
  for (int i = 0; i < e->n; i++) order[i] = i;
  for (int a = 1; a < e->n; a++) {
    int ja = order[a], b = a - 1;
    while (b >= 0 && e->v[order[b]].x < e->v[ja].x) { order[b + 1] = order[b]; b--; }
    order[b + 1] = ja;
  }
}

/* merge_env_v1.py:168-172 */
static int is_terminal(const MMConfig *cfg, const Env *e) {
  int any_crashed = 0, any_neg = 0;
  for (int i = 0; i < e->n_ctrl; i++) { any_crashed |= e->v[i].crashed; any_neg |= e->v[i].x < 0; }
  return any_crashed || e->steps >= cfg->duration * cfg->policy_frequency || any_neg;
}

/* abstract.py:512-532 _simulate with road.act / road.step (road.py:269-292) */
static int simulate(const MMConfig *cfg, Env *e, const int32_t *actions, double *trace, int64_t A,
                    int64_t base) {
  const int is_lc = cfg->env_kind == MM_ENV_V1;
  const int nsub = cfg->simulation_frequency / cfg->policy_frequency;
  const double dt = 1.0 / cfg->simulation_frequency;
  int rc = 0, order[MM_MAX_AGENTS];
  for (int k = 0; k < nsub; k++) {
    if (e->time % nsub == 0) /* action_type.act(action): action.py:226-231 */
      for (int i = 0; i < e->n_ctrl; i++) /* an action outside 0..4 (KeyError in the reference, latched by mm_step) acts as IDLE */
        mdp_act(&e->v[i], (actions[i] < 0 || actions[i] >= MM_N_ACTIONS) ? 1 : actions[i], is_lc);
    sort_by_x_desc(e, order); /* road.act: front to back; HDVs read what earlier vehicles already decided */
    for (int r = 0; r < e->n; r++) {
      if (e->v[order[r]].kind == 2) idm_act(e, order[r]);
      else mdp_act(&e->v[order[r]], -1, is_lc);
    }
    sort_by_x_desc(e, order); /* road.step */
#ifdef ORC_QP_LOG
    g_qplog_stepped = 0; g_qplog_env = (int)(base / (e->n > 0 ? e->n : 1));
#endif
    for (int r = 0; r < e->n; r++) {
#ifdef ORC_QP_LOG
      g_qplog_rank = r;
#endif
      int rr = vehicle_step(cfg, e, order[r], dt);
      if (rr) rc = rr;
#ifdef ORC_QP_LOG
      g_qplog_stepped |= 1u << order[r];
#endif
    }
    for (int i = 0; i < e->n; i++) { /* collision loop road.py:288-292, kinematics.py:175-200 */
      Veh *v = &e->v[i];
      for (int j = 0; j < e->n; j++) {
        Veh *o = &e->v[j];
        if (v->crashed || j == i) continue;
        if (is_colliding(v, o->x, o->y, VEH_LENGTH, VEH_WIDTH, o->heading)) {
          double s = fabs(v->speed) <= fabs(o->speed) ? v->speed : o->speed;
          v->speed = o->speed = s;
          v->crashed = o->crashed = 1;
        }
      }
      if (!v->crashed && is_colliding(v, OBST_X, OBST_Y, OBST_LENGTH, OBST_WIDTH, 0.0)) {
        v->speed = fabs(v->speed) <= 0 ? v->speed : 0.0;
        v->crashed = 1;
      }
    }
    e->time += 1;
    if (trace) {
      for (int i = 0; i < e->n; i++) {
        const Veh *v = &e->v[i];
        double *t = trace + (int64_t)k * MM_T_COUNT * A + base + i;
        t[MM_T_X * A] = v->x; t[MM_T_Y * A] = v->y; t[MM_T_HEADING * A] = v->heading;
        t[MM_T_SPEED * A] = v->speed; t[MM_T_ACT_STEER * A] = v->act_steer;
        t[MM_T_ACT_ACC * A] = v->act_acc;
        t[MM_T_SAFE_STEER * A] = (is_lc && v->kind == 1) ? v->safe_steer : v->act_steer;
        t[MM_T_SAFE_ACC * A] = (is_lc && v->kind == 1) ? v->safe_acc : v->act_acc;
        t[MM_T_LANE * A] = v->lane; t[MM_T_TARGET_LANE * A] = v->target_lane;
        t[MM_T_CRASHED * A] = v->crashed; t[MM_T_FLAGS * A] = v->flags;
        t[MM_T_QP_ROWS * A] = v->qp_rows; t[MM_T_QP_A * A] = v->qp_a;
        t[MM_T_QP_H0 * A] = v->qp_h[0]; t[MM_T_QP_H1 * A] = v->qp_h[1];
        t[MM_T_QP_H2 * A] = v->qp_h[2]; t[MM_T_QP_H3 * A] = v->qp_h[3];
        t[MM_T_QP_D * A] = v->qp_d; t[MM_T_LC_MARGIN * A] = v->lc_margin;
        if (v->qp_rows > 0) {
          t[MM_T_STATUS * A] = (double)(MM_ST_RAN | (v->qp_optimal ? MM_ST_IS_OPTIMAL : 0) | (v->lon_safe ? MM_ST_IS_SAFE : 0) |
                                        (v->lon_invariant ? MM_ST_IS_INVARIANT : 0) | (v->qp_bounds ? MM_ST_QP_BOUNDS : 0));
          t[MM_T_HEADWAY * A] = v->shield_headway;
        }
      }
    }
    if (is_terminal(cfg, e)) break;
  }
  return rc;
}

/* ------------------------------------------------------------------ observation.py */

/* observation.py:195-226 (Kinematics) / :244-273 (KinematicLC) for observer i; out[5*F] */
static void observe_agent(const MMConfig *cfg, const Env *e, int i, double *out) {
  const int F = cfg->env_kind == MM_ENV_V1 ? 6 : 5;
  const Veh *me = &e->v[i];
  for (int k = 0; k < 5 * F; k++) out[k] = 0.0;
  double evx = me->speed * m_cos(me->heading), evy = me->speed * m_sin(me->heading);
  double rows[5][6];
  int nrows = 1;
  rows[0][0] = 1; rows[0][1] = me->x; rows[0][2] = me->y; rows[0][3] = evx; rows[0][4] = evy;
  rows[0][5] = me->heading;
  int near[MM_MAX_AGENTS];
  int m = close_vehicles_to(e, i, PERCEPTION_DIST, 5 - 1, near);
  for (int k = 0; k < m; k++) {
    const Veh *o = &e->v[near[k]];
    double ovx = o->speed * m_cos(o->heading), ovy = o->speed * m_sin(o->heading);
    rows[nrows][0] = 1; rows[nrows][1] = o->x - me->x; rows[nrows][2] = o->y - me->y;
    rows[nrows][3] = ovx - evx; rows[nrows][4] = ovy - evy; rows[nrows][5] = o->heading;
    /* MDPLCVehicle.to_dict under "steer_vel": a neighbour's heading is taken relative to the observer's
     * (safe_controller.py:75-81); IDM vehicles use the base to_dict and stay absolute */
    if (o->steer_vel) rows[nrows][5] = o->heading - me->heading;
    nrows++;
  }
  /* normalize_obs :181-193 with utils.lmap :16-18; ranges :171-176, :238-239; clip=False */
  const double lo[6] = {0, -5.0 * 30, -12, -1.5 * 30, -1.5 * 30, -PI / 2};
  const double hi[6] = {0, 5.0 * 30, 12, 1.5 * 30, 1.5 * 30, PI / 2};
  for (int r = 0; r < nrows; r++) {
    out[r * F + 0] = rows[r][0];
    for (int f = 1; f < F; f++)
      out[r * F + f] = -1 + (rows[r][f] - lo[f]) * (1 - (-1)) / (hi[f] - lo[f]);
  }
}

/* abstract.py:219-240 + the row-aliasing of :202,:475 (mask = OR over agents when masking) */
static void action_mask(const MMConfig *cfg, const Env *e, uint8_t *out /* [n][5] */) {
  uint8_t m[5] = {1, 1, 1, 1, 1};
  if (cfg->action_masking) {
    memset(m, 0, 5);
    for (int i = 0; i < e->n_ctrl; i++) {
      const Veh *v = &e->v[i];
      m[1] = 1;
      if (v->lane == MM_LANE_BC1 && lane_is_reachable_from(MM_LANE_BC0, v->x, v->y)) m[0] = 1;
      if (v->lane == MM_LANE_BC0 && lane_is_reachable_from(MM_LANE_BC1, v->x, v->y)) m[2] = 1;
      if (v->speed_index < 5 - 1) m[3] = 1;
      if (v->speed_index > 0) m[4] = 1;
    }
  }
  for (int i = 0; i < e->n_ctrl; i++) memcpy(out + 5 * i, m, 5);
}

/* ------------------------------------------------------------------ rewards / info */

/* abstract.py:620-635 */
static double compute_headway_distance(const Env *e, int i) {
  const Veh *veh = &e->v[i];
  double hd = 60;
  int nl = next_lane(veh->lane, veh->x, veh->y);
  for (int j = 0; j < e->n; j++) {
    const Veh *v = &e->v[j];
    if (v->lane == veh->lane && v->x > veh->x) {
      double d = v->x - veh->x;
      if (d < hd) hd = d;
    }
    if (veh->lane != MM_LANE_BC1 && v->lane == nl && v->x > veh->x) {
      double d = v->x - veh->x;
      if (d < hd) hd = d;
    }
  }
  return hd;
}
/* merge_env_v1.py:64-89 (v1 "default" agent_reward delegates here, :446-447) */
static double agent_reward(const MMConfig *cfg, const Env *e, int i) {
  const Veh *v = &e->v[i];
  if (cfg->env_kind == MM_ENV_V1 && cfg->agent_reward != 0) {
    /* MergeEnvLCMARL._agent_reward, "srew" / "mrew" (merge_env_v1.py:439-474) */
    const int is_mrew = cfg->agent_reward == 2;
    double lo = cfg->reward_speed_lo, hi = cfg->reward_speed_hi;
    if ((v->flags & MM_FLAG_IS_COLLABORATING) && is_mrew) hi = lo + (hi - lo) / 2;
    double scaled = 0 + (v->speed - lo) * (1 - 0) / (hi - lo);
    double merging2 = 0;
    if (v->lane == MM_LANE_BC1 && (!is_mrew || (v->flags & MM_FLAG_IS_LC_SAFE)))
      merging2 = -m_exp(-m_sq(v->x - 420) / (10 * 100));
    double hd2 = compute_headway_distance(e, i);
    double hc2 = v->speed > 0 ? -1 * m_log(hd2 / (cfg->headway_time * v->speed)) : 0;
    return cfg->collision_reward * (-1 * v->crashed) + (cfg->high_speed_reward * clipd(scaled, 0, 1)) +
           cfg->merging_lane_cost * merging2 + cfg->headway_cost * (hc2 < 0 ? hc2 : 0);
  }
  double scaled_speed = 0 + (v->speed - cfg->reward_speed_lo) * (1 - 0) /
                                (cfg->reward_speed_hi - cfg->reward_speed_lo);
  double merging = 0;
  if (v->lane == MM_LANE_BC1) merging = -m_exp(-m_sq(v->x - 420) / (10 * 100));
  double hd = compute_headway_distance(e, i);
  double hc = v->speed > 0 ? m_log(hd / (cfg->headway_time * v->speed)) : 0;
  return cfg->collision_reward * (-1 * v->crashed) + (cfg->high_speed_reward * clipd(scaled_speed, 0, 1)) +
         cfg->merging_lane_cost * merging + cfg->headway_cost * (hc < 0 ? hc : 0);
}
/* road.py:294-350 surrounding_vehicles with the hard-coded lane cases; returns front/rear index or -1 */
static void surrounding_vehicles(const Env *e, int i, int lane_index, int *front, int *rear) {
  static const uint8_t allow[6] = {
      /* ab0 */ (1 << MM_LANE_AB0) | (1 << MM_LANE_BC0),
      /* bc0 */ (1 << MM_LANE_AB0) | (1 << MM_LANE_BC0) | (1 << MM_LANE_CD0),
      /* bc1 */ (1 << MM_LANE_KB0) | (1 << MM_LANE_BC1),
      /* cd0 */ (1 << MM_LANE_BC0) | (1 << MM_LANE_CD0),
      /* jk0 */ (1 << MM_LANE_JK0) | (1 << MM_LANE_KB0),
      /* kb0 */ (1 << MM_LANE_JK0) | (1 << MM_LANE_KB0) | (1 << MM_LANE_BC1)};
  double s = e->v[i].x, s_front = 0, s_rear = 0;
  *front = *rear = -1;
  for (int j = 0; j < e->n; j++) {
    if (j == i) continue;
    if (!((allow[lane_index] >> e->v[j].lane) & 1)) continue;
    double s_v = e->v[j].x;
    if (s <= s_v && (*front < 0 || s_v <= s_front)) { s_front = s_v; *front = j; }
    if (s_v < s && (*rear < 0 || s_v > s_rear)) { s_rear = s_v; *rear = j; }
  }
}
/* merge_env_v1.py:91-124 */
static void regional_reward(Env *e) {
  for (int i = 0; i < e->n_ctrl; i++) {
    Veh *v = &e->v[i];
    int fl = -1, rl = -1, fr = -1, rr = -1;
    if (v->lane == MM_LANE_AB0 || v->lane == MM_LANE_BC0 || v->lane == MM_LANE_CD0) {
      surrounding_vehicles(e, i, v->lane, &fl, &rl);
      if (v->lane == MM_LANE_BC0) surrounding_vehicles(e, i, MM_LANE_BC1, &fr, &rr);
      else if (v->lane == MM_LANE_AB0 && v->x > 220) surrounding_vehicles(e, i, MM_LANE_KB0, &fr, &rr);
    } else {
      surrounding_vehicles(e, i, v->lane, &fr, &rr);
      if (v->lane == MM_LANE_BC1) surrounding_vehicles(e, i, MM_LANE_BC0, &fl, &rl);
      else if (v->lane == MM_LANE_KB0) surrounding_vehicles(e, i, MM_LANE_AB0, &fl, &rl);
    }
    int list[5] = {fl, fr, i, rl, rr};
    double sum = 0; int cnt = 0;
    for (int k = 0; k < 5; k++)
      if (list[k] >= 0 && e->v[list[k]].kind == 1) { sum += e->v[list[k]].local_reward; cnt++; } /* isinstance(v, MDPVehicle) */
    v->regional_reward = sum / cnt;
  }
}
/* merge_env_v1.py:373-386 */
static double min_time_headway(const Env *e) {
  double mh = INFINITY;
  for (int i = 0; i < e->n_ctrl; i++) {
    const Veh *v = &e->v[i];
    double hd = compute_headway_distance(e, i);
    if (fabs(OBST_Y - v->y) <= 2 && OBST_X > v->x) {
      double d = OBST_X - v->x;
      if (d < hd) hd = d;
    }
    hd = hd - VEH_LENGTH;
    double vx = v->speed * m_cos(v->heading);
    double th = hd / (vx > 1 ? vx : 1);
    if (th < mh) mh = th;
  }
  return mh;
}

/* ------------------------------------------------------------------ reset */

/* Philox4x32-10 (Salmon et al. 2011); the device reset kernel implements the same stream. */
static void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                       uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static uint32_t rng_u32(uint64_t seed, uint32_t episode, uint32_t t) {
  uint32_t o[4];
  philox4x32(t >> 2, episode, 0u, 0x4D4D5253u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return o[t & 3];
}
static double rng_f64(uint64_t seed, uint32_t episode, uint32_t t) { /* 53-bit, like numpy's rk_double */
  uint32_t a = rng_u32(seed, episode, t) >> 5, b = rng_u32(seed, episode, t + 1) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* Vehicle/ControlledVehicle/MDPVehicle/MDPLCVehicle.__init__ from (x, y, heading, speed):
 * kinematics.py:36-53, controller.py:35-50,277-291, safe_controller.py:27-61 */
static void init_vehicle(Veh *v) {
  v->lane = closest_lane(v->x, v->y, v->heading);
  v->target_lane = v->lane;
  v->target_speed = v->speed;
  v->timer = 0;
  if (v->kind == 2) { /* IDMVehicle.__init__ behavior.py:42-53: timer = (sum(position) * pi) % LANE_CHANGE_DELAY */
    v->speed_index = 0;
    v->timer = py_mod((v->x + v->y) * PI, MOBIL_DELAY);
  } else {
    v->speed_index = speed_to_index(v->target_speed);
    v->target_speed = index_to_speed(v->speed_index);
  }
  v->act_steer = v->act_acc = 0;
  v->sc_valid = 0;
  v->safe_steer = v->safe_acc = 0;
  v->g_vx = NAN; /* fg_params = None */
  v->steer_angle = 0; /* safe_controller.py:54 */
  memset(v->h1, 0, sizeof v->h1);
  memset(v->h2, 0, sizeof v->h2);
  v->crashed = 0; v->hl_action = MM_HL_NONE; v->flags = 0; v->hist_len = 0;
  v->local_reward = v->regional_reward = 0;
}

/* MergeEnv._num_vehicles (merge_env_v1.py:180-211; MergeEnvLCMARL :476-495 sets mixed_traffic from traffic_type) with the
 * device RNG: block 64 of the episode's stream, word 0 -> num_CAV, word 1 -> num_HDV, each uniform over its 3 values
 * (np.random.choice(np.arange(lo, lo + 3), 1)).  traffic_density 0 = the fixed counts of the BASELINE configs. */
static void episode_counts(const MMConfig *cfg, int N, uint64_t seed, uint32_t episode, int *n_cav, int *n_hdv) {
  *n_cav = N - cfg->n_hdv; *n_hdv = cfg->n_hdv;
  if (cfg->traffic_density <= 0) return;
  mm_counts_from_draw(cfg->traffic_density, cfg->mixed_traffic, cfg->num_cav, (int)(((uint64_t)rng_u32(seed, episode, 4u * 64u) * 3u) >> 32),
                      (int)(((uint64_t)rng_u32(seed, episode, 4u * 64u + 1u) * 3u) >> 32), n_cav, n_hdv); /* include/mm_counts.h */
}

/* merge_env_v1.py:265-364 _make_vehicles for N CAVs / 0 HDVs with the device RNG stream
 * (draw plan documented in DESIGN.md "Device reset"): N/2 on ab0 first, the rest on jk0. */
static void spawn_env(Env *e, int n_cav, int n_hdv, uint64_t seed, uint32_t episode) {
  int slots_s[6] = {10, 60, 110, 160, 210, 260}, slots_m[6] = {5, 55, 105, 155, 205, 255};
  const uint32_t coin = rng_u32(seed, episode, 0);
  int n_s = (n_cav != 1) ? n_cav / 2 : (int)(coin & 1u);           /* CAVs on the main road */
  int n_m = n_cav - n_s;                                            /* CAVs on the ramp */
  int n_sh = (n_hdv != 1) ? n_hdv / 2 : (int)((coin >> 1) & 1u);    /* HDVs: remaining slots (:298-308) */
  int n_mh = n_hdv - n_sh;
  /* one partial Fisher-Yates per road: CAV slots first, HDV slots continue the same shuffle, so
   * all spawn points are distinct like the reference's remove-then-choice (:287-308) */
  for (int i = 0; i < n_s + n_sh && i < 6; i++) {
    int j = i + (int)(((uint64_t)rng_u32(seed, episode, (uint32_t)i) * (uint32_t)(6 - i)) >> 32);
    int t = slots_s[i]; slots_s[i] = slots_s[j]; slots_s[j] = t;
  }
  for (int i = 0; i < n_m + n_mh && i < 6; i++) {
    int j = i + (int)(((uint64_t)rng_u32(seed, episode, (uint32_t)(6 + i)) * (uint32_t)(6 - i)) >> 32);
    int t = slots_m[i]; slots_m[i] = slots_m[j]; slots_m[j] = t;
  }
  const int N = n_cav + n_hdv;
  e->n = N; e->n_ctrl = n_cav;
  for (int k = 0; k < N; k++) { /* creation order: CAV main, CAV ramp, HDV main, HDV ramp (:326-362) */
    Veh *v = &e->v[k];
    memset(v, 0, sizeof *v);
    double speed = rng_f64(seed, episode, 12u + 4u * k) * 2 + 25;
    double noise = rng_f64(seed, episode, 12u + 4u * k + 2u) * 8 - 4;
    if (k < n_s) { v->x = slots_s[k] + noise; v->y = 0.0; }
    else if (k < n_cav) { v->x = slots_m[k - n_s] + noise; v->y = 10.5; }
    else if (k < n_cav + n_sh) { v->x = slots_s[n_s + (k - n_cav)] + noise; v->y = 0.0; }
    else { v->x = slots_m[n_m + (k - n_cav - n_sh)] + noise; v->y = 10.5; }
    v->heading = 0; v->speed = speed;
    v->kind = k < n_cav ? 1 : 2;
    init_vehicle(v);
  }
  e->steps = e->time = 0;
  e->n_merge = n_m;
}

/* ------------------------------------------------------------------ SoA <-> Env */

static void load_env(const struct MMHandle_ *h, int64_t e_idx, Env *e) {
  const int64_t A = (int64_t)h->E * h->N, base = e_idx * h->N;
  const double *F = (const double *)(h->state + h->lay.f64_offset);
  const uint8_t *B = h->state + h->lay.u8_offset;
  const int32_t *I = (const int32_t *)(h->state + h->lay.env_offset);
  e->n = 0; e->n_ctrl = 0;
  for (int a = 0; a < h->N; a++) {
    int64_t i = base + a;
    if (B[MM_B_KIND * A + i] == 0) continue;
    Veh *v = &e->v[e->n++];
    memset(v, 0, sizeof *v);
    v->x = F[MM_F_X * A + i]; v->y = F[MM_F_Y * A + i]; v->heading = F[MM_F_HEADING * A + i];
    v->speed = F[MM_F_SPEED * A + i]; v->target_speed = F[MM_F_TARGET_SPEED * A + i];
    v->safe_steer = F[MM_F_SAFE_STEER * A + i]; v->safe_acc = F[MM_F_SAFE_ACC * A + i];
    v->g_vx = F[MM_F_G_VX * A + i];
    v->steer_angle = F[MM_F_STEER_ANGLE * A + i];
    for (int k = 0; k < 2; k++) { v->h1[k] = F[(MM_F_H1_X + k) * A + i]; v->h2[k] = F[(MM_F_H2_X + k) * A + i]; }
    v->lane = B[MM_B_LANE * A + i]; v->target_lane = B[MM_B_TARGET_LANE * A + i];
    v->speed_index = B[MM_B_SPEED_INDEX * A + i]; v->crashed = B[MM_B_CRASHED * A + i];
    v->hl_action = B[MM_B_HL_ACTION * A + i]; v->flags = B[MM_B_FLAGS * A + i];
    v->hist_len = B[MM_B_HIST_LEN * A + i]; v->kind = B[MM_B_KIND * A + i];
    if (v->kind == 1 && e->n_ctrl == e->n - 1) e->n_ctrl = e->n; /* controlled vehicles are a prefix */
    v->steer_vel = v->kind == 1 && h->cfg.env_kind == MM_ENV_V1 && h->cfg.lateral_control == MM_LATERAL_STEER_VEL;
    if (v->kind == 2) { /* HDV: the SAFE_* planes persist its last IDM action, G_VX its MOBIL timer */
      v->act_steer = v->safe_steer; v->act_acc = v->safe_acc; v->timer = v->g_vx;
    }
    v->sc_valid = 0;
  }
  e->steps = I[MM_E_STEPS * h->E + e_idx]; e->time = I[MM_E_TIME * h->E + e_idx];
  e->n_merge = I[MM_E_N_MERGE * h->E + e_idx]; e->episode = I[MM_E_EPISODE * h->E + e_idx];
}
static void store_env(struct MMHandle_ *h, int64_t e_idx, const Env *e) {
  const int64_t A = (int64_t)h->E * h->N, base = e_idx * h->N;
  double *F = (double *)(h->state + h->lay.f64_offset);
  uint8_t *B = h->state + h->lay.u8_offset;
  int32_t *I = (int32_t *)(h->state + h->lay.env_offset);
  for (int a = 0; a < h->N; a++) {
    int64_t i = base + a;
    if (a >= e->n) {
      if (B[MM_B_KIND * A + i] != 0) { /* the slot empties (ragged batch, re-drawn counts): every plane reads zero from now on */
        for (int k = 0; k < MM_F_COUNT; k++) F[k * A + i] = 0.0;
        for (int k = 0; k < MM_B_COUNT; k++) B[k * A + i] = 0;
      }
      B[MM_B_KIND * A + i] = 0;
      continue;
    }
    const Veh *v = &e->v[a];
    F[MM_F_X * A + i] = v->x; F[MM_F_Y * A + i] = v->y; F[MM_F_HEADING * A + i] = v->heading;
    F[MM_F_SPEED * A + i] = v->speed; F[MM_F_TARGET_SPEED * A + i] = v->target_speed;
    F[MM_F_SAFE_STEER * A + i] = v->kind == 2 ? v->act_steer : v->safe_steer;
    F[MM_F_SAFE_ACC * A + i] = v->kind == 2 ? v->act_acc : v->safe_acc;
    F[MM_F_G_VX * A + i] = v->kind == 2 ? v->timer : v->g_vx;
    F[MM_F_STEER_ANGLE * A + i] = v->steer_angle;
    for (int k = 0; k < 2; k++) { F[(MM_F_H1_X + k) * A + i] = v->h1[k]; F[(MM_F_H2_X + k) * A + i] = v->h2[k]; }
    B[MM_B_LANE * A + i] = (uint8_t)v->lane; B[MM_B_TARGET_LANE * A + i] = (uint8_t)v->target_lane;
    B[MM_B_SPEED_INDEX * A + i] = (uint8_t)v->speed_index; B[MM_B_CRASHED * A + i] = (uint8_t)v->crashed;
    B[MM_B_HL_ACTION * A + i] = (uint8_t)v->hl_action; B[MM_B_FLAGS * A + i] = (uint8_t)v->flags;
    B[MM_B_HIST_LEN * A + i] = (uint8_t)v->hist_len; B[MM_B_KIND * A + i] = (uint8_t)v->kind;
  }
  I[MM_E_STEPS * h->E + e_idx] = e->steps; I[MM_E_TIME * h->E + e_idx] = e->time;
  I[MM_E_N_MERGE * h->E + e_idx] = e->n_merge; I[MM_E_EPISODE * h->E + e_idx] = e->episode;
}

static void write_obs(const struct MMHandle_ *h, const Env *e, int64_t e_idx, void *obs, uint8_t *avail) {
  const int F = h->cfg.env_kind == MM_ENV_V1 ? 6 : 5, S = 5 * F;
  double row[30];
  for (int a = 0; a < h->N; a++) {
    if (a < e->n_ctrl) observe_agent(&h->cfg, e, a, row);
    else memset(row, 0, sizeof row);
    int64_t o = (e_idx * h->N + a) * S;
    if (obs) {
      if (h->cfg.obs_f64) memcpy((double *)obs + o, row, S * sizeof(double));
      else for (int k = 0; k < S; k++) ((float *)obs)[o + k] = (float)row[k];
    }
  }
  if (avail) {
    uint8_t m[MM_MAX_AGENTS * 5];
    memset(m, 0, sizeof m);
    action_mask(&h->cfg, e, m);
    memcpy(avail + e_idx * h->N * 5, m, (size_t)h->N * 5);
  }
}

/* ------------------------------------------------------------------ C ABI */

int32_t mm_abi_version(void) { return MM_ABI_VERSION; }

static uint64_t align256(uint64_t x) { return (x + 255u) & ~(uint64_t)255u; }

int32_t mm_state_layout(int32_t E, int32_t N, MMStateLayout *out) {
  if (!out || E <= 0 || N <= 0 || N > MM_MAX_AGENTS) return MM_ERR_INVALID_ARG;
  uint64_t A = (uint64_t)E * (uint64_t)N, off = 0;
  out->f64_offset = off; off = align256(off + A * 8u * MM_F_COUNT);
  out->u8_offset = off; off = align256(off + A * MM_B_COUNT);
  out->env_offset = off; off = align256(off + (uint64_t)E * 4u * MM_E_COUNT);
  out->seed_offset = off; off = align256(off + (uint64_t)E * 8u);
  out->total_bytes = off;
  return MM_OK;
}

static _Thread_local char g_create_err[256] = "null handle"; /* why the last mm_create of this thread refused */
static int check_cfg(const MMConfig *c, int N, char *err) {
  if (!c || c->abi_version != MM_ABI_VERSION) { snprintf(err, 256, "ABI version mismatch"); return MM_ERR_INVALID_ARG; }
  if (c->env_kind != MM_ENV_V0 && c->env_kind != MM_ENV_V1) { snprintf(err, 256, "unknown env_kind %d", c->env_kind); return MM_ERR_INVALID_ARG; }
  if (c->shield < MM_SHIELD_NONE || c->shield > MM_SHIELD_MASS) { snprintf(err, 256, "Undefined safety_type:%d", c->shield); return MM_ERR_INVALID_ARG; }
  if (c->policy_frequency <= 0 || c->simulation_frequency < c->policy_frequency ||
      c->simulation_frequency / c->policy_frequency > 3) { snprintf(err, 256, "unsupported frequencies"); return MM_ERR_INVALID_ARG; }
  if (N > 12) { snprintf(err, 256, "N=%d exceeds the 6+6 spawn slots", N); return MM_ERR_INVALID_ARG; }
  if (c->n_hdv < 0 || c->n_hdv >= N) { snprintf(err, 256, "n_hdv=%d must leave at least one controlled vehicle of N=%d", c->n_hdv, N); return MM_ERR_INVALID_ARG; }
  if (c->qp_solver != MM_QP_EXACT && c->qp_solver != MM_QP_IPM) { snprintf(err, 256, "unknown qp_solver %d", c->qp_solver); return MM_ERR_INVALID_ARG; }
  if (mm_counts_check(c, N, 0, err, 256)) return MM_ERR_INVALID_ARG; /* slots and spawn points for every composition (include/mm_counts.h) */
  return MM_OK;
}

int32_t mm_create(const MMConfig *cfg, int32_t E, int32_t N, int32_t device, void *state,
                  uint64_t state_bytes, int64_t first_env, MMHandle *out) {
  (void)device;
  if (!out || !state) return MM_ERR_INVALID_ARG;
  struct MMHandle_ *h = calloc(1, sizeof *h);
  snprintf(h->err, sizeof h->err, "mm_create: E, N, the state buffer size or the configuration is invalid");
  if (mm_state_layout(E, N, &h->lay) != MM_OK || state_bytes < h->lay.total_bytes ||
      check_cfg(cfg, N, h->err) != MM_OK) { snprintf(g_create_err, sizeof g_create_err, "%s", h->err); free(h); return MM_ERR_INVALID_ARG; }
  h->err[0] = 0;
  h->cfg = *cfg; h->E = E; h->N = N; h->state = state; h->first_env = first_env;
  uint64_t *seeds = (uint64_t *)(h->state + h->lay.seed_offset);
  for (int64_t e = 0; e < E; e++) seeds[e] = cfg->seed + (uint64_t)(first_env + e);
  *out = h;
  return MM_OK;
}
int32_t mm_destroy(MMHandle h) { free(h); return MM_OK; }
int32_t mm_set_config(MMHandle h, const MMConfig *cfg) {
  if (!h) return MM_ERR_INVALID_ARG;
  int rc = check_cfg(cfg, h->N, h->err);
  if (rc == MM_OK) h->cfg = *cfg;
  return rc;
}
int32_t mm_set_metrics_buffer(MMHandle h, double *metrics) { h->metrics = metrics; return MM_OK; }
/* (the CPU twin adds every step's sums to the caller's buffer directly: there is nothing to defer or to flush) */
int32_t mm_defer_metrics(MMHandle h, int32_t deferred, MMStream stream) {
  (void)stream;
  if (!h) return MM_ERR_INVALID_ARG;
  if (deferred && !h->metrics) { snprintf(h->err, sizeof h->err, "mm_defer_metrics: no metrics buffer (mm_set_metrics_buffer first)"); return MM_ERR_INVALID_ARG; }
  return MM_OK;
}
int32_t mm_flush_metrics(MMHandle h, MMStream stream) { (void)stream; return h ? MM_OK : MM_ERR_INVALID_ARG; }
const char *mm_last_error(MMHandle h) { return h ? h->err : g_create_err; }

int32_t mm_reset(MMHandle h, const uint8_t *env_mask, const uint64_t *seeds_in, void *obs,
                 uint8_t *avail, MMStream stream) {
  (void)stream;
  if (!h) return MM_ERR_INVALID_ARG;
  if (mm_counts_check(&h->cfg, h->N, 1, h->err, sizeof h->err)) return MM_ERR_INVALID_ARG; /* the spawn below is N - n_hdv CAVs + n_hdv HDVs */
  uint64_t *seeds = (uint64_t *)(h->state + h->lay.seed_offset);
#pragma omp parallel for schedule(static)
  for (int64_t e_idx = 0; e_idx < h->E; e_idx++) {
    Env e;
    if (env_mask && !env_mask[e_idx]) { load_env(h, e_idx, &e); write_obs(h, &e, e_idx, obs, avail); continue; }
    load_env(h, e_idx, &e);
    if (seeds_in) seeds[e_idx] = seeds_in[e_idx];
    int episode = e.episode;
    int n_cav, n_hdv;
    episode_counts(&h->cfg, h->N, seeds[e_idx], (uint32_t)episode, &n_cav, &n_hdv);
    spawn_env(&e, n_cav, n_hdv, seeds[e_idx], (uint32_t)episode);
    e.episode = episode + 1;
    store_env(h, e_idx, &e);
    write_obs(h, &e, e_idx, obs, avail);
  }
  return MM_OK;
}

int32_t mm_init_from_kinematics(MMHandle h, const uint8_t *env_mask, MMStream stream) {
  (void)stream;
  const int64_t A = (int64_t)h->E * h->N;
  const uint8_t *B = h->state + h->lay.u8_offset;
#pragma omp parallel for schedule(static)
  for (int64_t e_idx = 0; e_idx < h->E; e_idx++) {
    if (env_mask && !env_mask[e_idx]) continue;
    Env e;
    load_env(h, e_idx, &e);
    int n_m = 0;
    for (int a = 0; a < e.n; a++) {
      init_vehicle(&e.v[a]);
      if (a < e.n_ctrl && (e.v[a].lane == MM_LANE_JK0 || e.v[a].lane == MM_LANE_KB0)) n_m++;
    }
    (void)A; (void)B;
    e.steps = e.time = 0; e.n_merge = n_m;
    store_env(h, e_idx, &e);
  }
  return MM_OK;
}

int32_t mm_observe(MMHandle h, void *obs, uint8_t *avail, MMStream stream) {
  (void)stream;
#pragma omp parallel for schedule(static)
  for (int64_t e_idx = 0; e_idx < h->E; e_idx++) {
    Env e;
    load_env(h, e_idx, &e);
    write_obs(h, &e, e_idx, obs, avail);
  }
  return MM_OK;
}

int32_t mm_step(MMHandle h, const int32_t *actions, const MMStepOut *out, MMStream stream) {
  (void)stream;
  if (!h || !actions || !out) return MM_ERR_INVALID_ARG;
  const MMConfig *cfg = &h->cfg;
  const int64_t A = (int64_t)h->E * h->N;
  int rc_all = MM_OK;
  uint64_t *seeds = (uint64_t *)(h->state + h->lay.seed_offset);
  { /* DiscreteMetaAction.act: self.actions[action] raises KeyError outside 0..4 (action.py:194-196) */
    const uint8_t *kind = h->state + h->lay.u8_offset + (int64_t)MM_B_KIND * A;
    for (int64_t i = 0; i < A; i++)
      if (kind[i] == 1 && (actions[i] < 0 || actions[i] >= MM_N_ACTIONS)) {
        h->latched = MM_ERR_INVALID_ARG;
        snprintf(h->err, sizeof h->err, "action %d of agent %lld is outside 0..4", (int)actions[i], (long long)i);
        break;
      }
  }
  if (out->trace)
    for (int64_t k = 0; k < 3 * (int64_t)MM_T_COUNT * A; k++) out->trace[k] = NAN;
  double m_sum[7] = {0}, m_min = INFINITY;
#pragma omp parallel for schedule(static) reduction(+ : m_sum[:7]) reduction(min : m_min)
  for (int64_t e_idx = 0; e_idx < h->E; e_idx++) {
    Env e;
    load_env(h, e_idx, &e);
    const int64_t base = e_idx * h->N;
    if (e.n_ctrl == 0) continue;
    e.steps += 1; /* abstract.py:457 */
    int rc = simulate(cfg, &e, actions + base, out->trace, A, base);
    if (rc) { /* check_bounds' ValueError (cbf.py:87-96): latched, reported by mm_poll_errors like the HIP twin */
#pragma omp critical
      { rc_all = rc; snprintf(h->err, sizeof h->err, "Error in QP. Invalid accceleration (env %lld)", (long long)e_idx); }
    }
    /* AbstractEnv.step abstract.py:469-498 then MergeEnv.step merge_env_v1.py:126-166 */
    int done = is_terminal(cfg, &e);
    double rsum = 0, ssum = 0, tsum = 0;
    for (int i = 0; i < e.n_ctrl; i++) {
      e.v[i].local_reward = agent_reward(cfg, &e, i);
      rsum += e.v[i].local_reward;
      ssum += e.v[i].speed;
    }
    for (int i = 0; i < e.n; i++) tsum += e.v[i].speed; /* traffic_speed over road.vehicles (:147-151) */
    double reward = rsum / e.n_ctrl, avg_speed = ssum / e.n_ctrl, traffic_speed = tsum / e.n;
    regional_reward(&e);
    double mh = min_time_headway(&e);
    double merge_pct = NAN;
    int any_crashed = 0;
    for (int i = 0; i < e.n_ctrl; i++) any_crashed |= e.v[i].crashed;
    if (done) {
      int n_rem = 0;
      for (int i = 0; i < e.n_ctrl; i++)
        if (e.v[i].lane == MM_LANE_BC1 || e.v[i].lane == MM_LANE_KB0 || e.v[i].lane == MM_LANE_JK0) n_rem++;
      merge_pct = e.n_merge > 0 ? (double)(e.n_merge - n_rem) / e.n_merge * 100 : 100.0;
    }
    if (out->reward) out->reward[e_idx] = reward;
    if (out->done) out->done[e_idx] = (uint8_t)done;
    if (out->average_speed) out->average_speed[e_idx] = avg_speed;
    if (out->traffic_speed) out->traffic_speed[e_idx] = traffic_speed;
    if (out->min_headway) out->min_headway[e_idx] = mh;
    if (out->merge_percent) out->merge_percent[e_idx] = merge_pct;
    const int T = cfg->duration * cfg->policy_frequency;
    for (int a = 0; a < h->N; a++) {
      const int live = a < e.n, ctrl = a < e.n_ctrl;
      const Veh *v = &e.v[a];
      if (out->agents_rewards) out->agents_rewards[base + a] = ctrl ? v->local_reward : 0;
      if (out->regional_rewards) out->regional_rewards[base + a] = ctrl ? v->regional_reward : 0;
      if (out->agents_dones) /* merge_env_v1.py:174-178 (controlled vehicles; other slots read 1) */
        out->agents_dones[base + a] = ctrl ? (uint8_t)(v->crashed || e.steps >= T || v->x < 0) : 1;
      if (out->crashed) out->crashed[base + a] = live ? (uint8_t)v->crashed : 0;
      if (out->agents_info) {
        out->agents_info[(base + a) * 3 + 0] = live ? v->x : 0;
        out->agents_info[(base + a) * 3 + 1] = live ? v->y : 0;
        out->agents_info[(base + a) * 3 + 2] = live ? v->speed : 0;
      }
    }
    m_sum[0] += reward; m_sum[2] += avg_speed; m_sum[3] += traffic_speed; m_sum[4] += 1;
    if (done) { m_sum[1] += any_crashed; m_sum[5] += merge_pct; m_sum[6] += 1; }
    if (mh < m_min) m_min = mh;
    if (done && cfg->auto_reset) { /* caller-side `if done: env.reset()` (marl/mappo.py:133-135) */
      int episode = e.episode;
      int n_cav = e.n_ctrl, n_hdv = e.n - e.n_ctrl; /* fixed counts: the episode keeps the env's own composition */
      if (cfg->traffic_density > 0) episode_counts(cfg, h->N, seeds[e_idx], (uint32_t)episode, &n_cav, &n_hdv);
      spawn_env(&e, n_cav, n_hdv, seeds[e_idx], (uint32_t)episode);
      e.episode = episode + 1;
    }
    store_env(h, e_idx, &e);
    write_obs(h, &e, e_idx, out->obs, out->action_mask);
  }
  if (h->metrics) {
    for (int k = 0; k < 7; k++) h->metrics[k] += m_sum[k];
    if (m_min < h->metrics[7]) h->metrics[7] = m_min;
  }
  if (rc_all != MM_OK && h->latched == MM_OK) h->latched = rc_all;
  return MM_OK;
}

int32_t mm_poll_errors(MMHandle h, MMStream stream) {
  (void)stream;
  if (!h) return MM_ERR_INVALID_ARG;
  const int rc = h->latched;
  h->latched = MM_OK;
  return rc;
}

int32_t mm_shield_actions(MMHandle h, const double *act_steer, const double *act_acc, double *safe_steer,
                          double *safe_acc, uint8_t *status, double *margin, double *headway, MMStream stream) {
  (void)stream;
  if (!h || !act_steer || !act_acc || !safe_steer || !safe_acc) return MM_ERR_INVALID_ARG;
  const MMConfig *cfg = &h->cfg;
  const double dt = 1.0 / cfg->simulation_frequency;
  int rc_all = MM_OK;
#pragma omp parallel for schedule(static)
  for (int64_t e_idx = 0; e_idx < h->E; e_idx++) {
    Env e;
    load_env(h, e_idx, &e);
    const int64_t base = e_idx * h->N;
    for (int a = 0; a < h->N; a++) {
      const int64_t i = base + a;
      safe_steer[i] = act_steer[i]; safe_acc[i] = act_acc[i];
      if (status) status[i] = 0;
      if (margin) margin[i] = NAN;
      if (headway) headway[i] = NAN;
      if (a >= e.n_ctrl) continue;
      /* gate of get_safe_action (safe_controller.py:229-239) */
      if (cfg->env_kind != MM_ENV_V1 || cfg->shield == MM_SHIELD_NONE || e.v[a].hist_len < 2) continue;
      Env w = e; /* safety_layer mutates the vehicle (and an on-ramp HDV's record): work on a copy */
      w.v[a].act_steer = act_steer[i]; w.v[a].act_acc = act_acc[i];
      double acc, steer;
      int rc = safety_layer(cfg, &w, a, dt, &acc, &steer);
      if (rc) {
#pragma omp critical
        { rc_all = rc; snprintf(h->err, sizeof h->err, "Error in QP. Invalid accceleration (env %lld)", (long long)e_idx); }
      }
      safe_steer[i] = steer; safe_acc[i] = acc;
      const int fl = w.v[a].flags;
      if (status)
        status[i] = (uint8_t)(MM_ST_RAN | (w.v[a].qp_optimal ? MM_ST_IS_OPTIMAL : 0) | (w.v[a].lon_safe ? MM_ST_IS_SAFE : 0) |
                              (w.v[a].lon_invariant ? MM_ST_IS_INVARIANT : 0) | (w.v[a].qp_bounds ? MM_ST_QP_BOUNDS : 0) |
                              ((fl & MM_FLAG_IS_LC_SAFE) ? MM_ST_IS_LC_SAFE : 0) |
                              ((fl & MM_FLAG_IS_COLLABORATING) ? MM_ST_IS_COLLABORATING : 0) |
                              ((fl & MM_FLAG_COLLABORATE_ADJ) ? MM_ST_COLLABORATE_ADJ : 0));
      if (margin) margin[i] = w.v[a].lc_margin;
      if (headway) headway[i] = w.v[a].shield_headway;
    }
  }
  return rc_all;
}

static int qp_structure_ok(const double *g, int rows) { /* the G of get_G (cbf.py:288-304,386-403) and no other */
  if (rows != 3 && rows != 4) return 0;
  int ok = g[1] == 0.0 && g[2] == -1.0 && g[3] == 1.0 && g[4] == 0.0 && g[5] == 0.0 && g[6] == -1.0 && g[7] == 0.0 &&
           g[8] == 0.0;
  if (rows == 4) ok = ok && g[9] == g[0] && g[10] == 0.0 && g[11] == -1.0;
  return ok;
}
int32_t mm_shield_qp(MMHandle h, int32_t n, const double *G, const double *hvec, const int32_t *rows, int32_t solver,
                     double *u_out, uint8_t *status, int32_t *iters, MMStream stream) {
  (void)stream;
  if (n < 0 || (n > 0 && (!G || !hvec || !rows || !u_out)) || (solver != MM_QP_EXACT && solver != MM_QP_IPM)) return MM_ERR_INVALID_ARG;
  int bad = 0;
  for (int k = 0; k < n; k++) {
    const double *g = G + (int64_t)k * 12, *hh = hvec + (int64_t)k * 4;
    if (iters) iters[k] = 0;
    if (!qp_structure_ok(g, rows[k])) {
      u_out[k * 3 + 0] = u_out[k * 3 + 1] = u_out[k * 3 + 2] = NAN;
      if (status) status[k] = MM_QPS_BAD_STRUCTURE;
      bad = 1;
      continue;
    }
    if (solver == MM_QP_IPM) { /* the general dense algorithm on the G as given */
      const double Pd[3] = {1.0, 1.0, 1e18}, q[3] = {0.0, 0.0, 0.0};
      double Gm[4][QP_N] = {{0}}, x[3];
      for (int r = 0; r < rows[k]; r++) for (int j = 0; j < QP_N; j++) Gm[r][j] = g[r * 3 + j];
      int it;
      const int ok = qp_ipm(Pd, q, Gm, hh, rows[k], x, &it);
      u_out[k * 3 + 0] = x[0]; u_out[k * 3 + 1] = x[1]; u_out[k * 3 + 2] = x[2];
      if (status) status[k] = ok ? MM_QPS_OPTIMAL : MM_QPS_UNKNOWN;
      if (iters) iters[k] = it;
      continue;
    }
    double a = g[0], hc = hh[0];
    if (rows[k] == 4 && hh[3] < hc) hc = hh[3];
    double hi = hh[1], lo = -hh[2], d;
    if (a > 0) d = fmin(0.0, hc / a);
    else if (a < 0) d = fmax(0.0, hc / a);
    else d = 0.0;
    d = fmin(fmax(d, lo), hi);
    double s = a * d - hc;
    u_out[k * 3 + 0] = d; u_out[k * 3 + 1] = 0.0; u_out[k * 3 + 2] = s > 0 ? s : 0.0;
    if (status) status[k] = MM_QPS_OPTIMAL;
  }
  if (bad) {
    if (h) snprintf(h->err, sizeof h->err, "mm_shield_qp: G is not of the form get_G builds (cbf.py:288-304,386-403)");
    return MM_ERR_INVALID_ARG;
  }
  return MM_OK;
}

int32_t mm_math_eval(int32_t fn, int32_t n, const double *x, const double *x2, double *y, MMStream stream) {
  (void)stream;
  for (int i = 0; i < n; i++) {
    switch (fn) {
      case 0: y[i] = mmm_sin(x[i]); break;
      case 1: y[i] = mmm_cos(x[i]); break;
      case 2: y[i] = mmm_tan(x[i]); break;
      case 3: y[i] = mmm_atan(x[i]); break;
      case 4: y[i] = mmm_asin(x[i]); break;
      case 5: y[i] = mmm_exp(x[i]); break;
      case 6: y[i] = mmm_log(x[i]); break;
      case 7: y[i] = sqrt(x[i]); break;
      case 8: y[i] = x[i] / x2[i]; break;
      case 9: y[i] = x[i] / x2[i]; break; /* the HIP side evaluates its 3-instruction constant-divisor form */
      default: return MM_ERR_INVALID_ARG;
    }
  }
  return MM_OK;
}

/* ------------------------------------------------------------------ unit hooks for tests */
/* include/mm_qp.h (the sparsity-specialised IPM the HIP kernels run) compiled for the host, so that the CPU
 * suite can hold it against the general dense qp_ipm above on every recorded QP without a GPU. */
#include "../include/mm_qp.h"
int32_t orc_qp_ipm_header(int32_t n, const double *a, const double *hvec, const int32_t *rows, double *d, double *s,
                          int32_t *iters, uint8_t *status) {
  for (int k = 0; k < n; k++) {
    int it;
    const double *hh = hvec + (int64_t)k * 4;
    status[k] = (uint8_t)mm_qp_ipm_cbf(a[k], hh[0], hh[1], hh[2], hh[3], rows[k], &d[k], &s[k], &it);
    iters[k] = it;
  }
  return MM_OK;
}
/* Pure functions exposed so tests can pin them against tests/golden/units.npz. */
int32_t orc_closest_lane(double x, double y, double h) { return closest_lane(x, y, h); }
int32_t orc_next_lane(int32_t lane, double x, double y) { return next_lane(lane, x, y); }
void orc_lane_local(int32_t lane, double x, double y, double *s, double *r) { lane_local(lane, x, y, s, r); }
double orc_lane_heading_at(int32_t lane, double s) { return lane_heading_at(lane, s); }
double orc_lane_distance_with_heading(int32_t lane, double x, double y, double h) { return lane_distance_with_heading(lane, x, y, h); }
int32_t orc_on_lane(int32_t lane, double x, double y) { return lane_on_lane(lane, x, y); }
int32_t orc_is_reachable_from(int32_t lane, double x, double y) { return lane_is_reachable_from(lane, x, y); }
int32_t orc_after_end(int32_t lane, double x, double y) { return lane_after_end(lane, x, y); }
double orc_steering_control(double x, double y, double heading, double speed, int32_t target_lane) {
  Veh v; memset(&v, 0, sizeof v); v.x = x; v.y = y; v.heading = heading; v.speed = speed;
  return steering_control(&v, target_lane);
}
void orc_get_corner(double x, double y, double heading, int32_t dir, double *cx, double *cy) {
  Veh v; memset(&v, 0, sizeof v); v.x = x; v.y = y; v.heading = heading;
  get_corner(&v, dir, cx, cy);
}
int32_t orc_speed_to_index(double speed) { return speed_to_index(speed); }
double orc_wrap_to_pi(double x) { return wrap_to_pi(x); }
int32_t orc_rect_intersect(double c1x, double c1y, double l1, double w1, double a1, double c2x,
                           double c2y, double l2, double w2, double a2) {
  return rotated_rectangles_intersect(c1x, c1y, l1, w1, a1, c2x, c2y, l2, w2, a2);
}

/* batched forms of the unit hooks (keeps the CPU test-suite fast) */
void orc_batch_pose(int32_t n, const double *x, const double *y, const double *h, double *local /*[n][6][2]*/,
                    double *lane_heading /*[n][6]*/, double *dist /*[n][6]*/, uint8_t *onlane, uint8_t *reach,
                    uint8_t *after, int32_t *nxt /*[n][6]*/, int32_t *closest /*[n]*/) {
  for (int k = 0; k < n; k++) {
    for (int l = 0; l < 6; l++) {
      double s, r;
      lane_local(l, x[k], y[k], &s, &r);
      local[(k * 6 + l) * 2] = s; local[(k * 6 + l) * 2 + 1] = r;
      lane_heading[k * 6 + l] = lane_heading_at(l, s);
      dist[k * 6 + l] = lane_distance_with_heading(l, x[k], y[k], h[k]);
      onlane[k * 6 + l] = (uint8_t)lane_on_lane(l, x[k], y[k]);
      reach[k * 6 + l] = (uint8_t)lane_is_reachable_from(l, x[k], y[k]);
      after[k * 6 + l] = (uint8_t)lane_after_end(l, x[k], y[k]);
      nxt[k * 6 + l] = next_lane(l, x[k], y[k]);
    }
    closest[k] = closest_lane(x[k], y[k], h[k]);
  }
}
void orc_batch_steering(int32_t n, const double *x, const double *y, const double *h, const double *v,
                        const int32_t *lane, double *steer, double *corner /*[n][2][2]*/) {
  for (int k = 0; k < n; k++) {
    steer[k] = orc_steering_control(x[k], y[k], h[k], v[k], lane[k]);
    orc_get_corner(x[k], y[k], h[k], 0, &corner[k * 4 + 0], &corner[k * 4 + 1]);
    orc_get_corner(x[k], y[k], h[k], 1, &corner[k * 4 + 2], &corner[k * 4 + 3]);
  }
}
void orc_batch_rect(int32_t n, const double *rect /*[n][6]*/, uint8_t *hit, uint8_t *hit_obstacle) {
  for (int k = 0; k < n; k++) {
    const double *r = rect + k * 6;
    hit[k] = (uint8_t)rotated_rectangles_intersect(r[0], r[1], 4.5, 1.8, r[2], r[3], r[4], 4.5, 1.8, r[5]);
    hit_obstacle[k] = (uint8_t)rotated_rectangles_intersect(r[0], r[1], 4.5, 1.8, r[2], r[3], r[4], 1.8, 1.8, 0.0);
  }
}

/* include/mm_abi.h mm_geom_eval: the oracle's own functions on the same row layouts (host pointers) */
int32_t mm_geom_eval(int32_t fn, int32_t n, const double *in, double *out, MMStream stream) {
  (void)stream;
  if (fn < MM_GEOM_POSE || fn > MM_GEOM_SPEED_INDEX || n <= 0 || !in || !out) return MM_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++) {
    if (fn == MM_GEOM_POSE) {
      const double x = in[i * 3 + 0], y = in[i * 3 + 1], h = in[i * 3 + 2];
      double *o = out + (int64_t)i * 19;
      o[0] = closest_lane(x, y, h);
      for (int l = 0; l < 6; l++) {
        o[1 + l] = next_lane(l, x, y);
        o[7 + l] = lane_is_reachable_from(l, x, y);
        o[13 + l] = lane_after_end(l, x, y);
      }
    } else if (fn == MM_GEOM_STEER) {
      const double *r = in + (int64_t)i * 5;
      out[i] = orc_steering_control(r[0], r[1], r[2], r[3], (int32_t)r[4]);
    } else if (fn == MM_GEOM_RECT) {
      const double *r = in + (int64_t)i * 6;
      double *o = out + (int64_t)i * 4;
      Veh a;
      memset(&a, 0, sizeof a);
      a.x = r[0]; a.y = r[1]; a.heading = r[2];
      o[0] = is_colliding(&a, r[3], r[4], VEH_LENGTH, VEH_WIDTH, r[5]);   /* kinematics.py:202-209: pre-check + 9-point test */
      o[1] = is_colliding(&a, r[3], r[4], OBST_LENGTH, OBST_WIDTH, 0.0);
      o[2] = rotated_rectangles_intersect(r[0], r[1], 0.9 * VEH_LENGTH, 0.9 * VEH_WIDTH, r[2], r[3], r[4], 0.9 * VEH_LENGTH, 0.9 * VEH_WIDTH, r[5]);
      o[3] = rotated_rectangles_intersect(r[0], r[1], 0.9 * VEH_LENGTH, 0.9 * VEH_WIDTH, r[2], r[3], r[4], 0.9 * OBST_LENGTH, 0.9 * OBST_WIDTH, 0.0);
    } else {
      out[i] = speed_to_index(in[i]);
    }
  }
  return MM_OK;
}

/* thread control for the cpu_baseline leg of bench.py (OpenMP over envs) */
#ifdef _OPENMP
#include <omp.h>
int32_t orc_set_threads(int32_t n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int32_t orc_set_threads(int32_t n) { (void)n; return 1; }
#endif

/* marl/mappo.py:147-156,364-370: reward scaling + discounted returns of a rollout (see include/mm_abi.h) */
int32_t mm_discount_returns(const double *rewards, const uint8_t *dones, const double *final_value, int32_t T,
                            int64_t n_env, int32_t n_agent, double gamma, double reward_scale, double *returns,
                            MMStream stream) {
  (void)stream;
  if (!rewards || !dones || !final_value || !returns || T < 0 || n_env < 0 || n_agent < 1) return MM_ERR_INVALID_ARG;
  const int64_t A = n_env * n_agent;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A; i++) {
    const int64_t e = i / n_agent;
    double running = final_value[i];
    for (int t = T - 1; t >= 0; t--) {
      double r = rewards[(int64_t)t * A + i];
      if (reward_scale > 0) r = r / reward_scale;
      if (dones[(int64_t)t * n_env + e]) running = 0.0;
      running = running * gamma + r;
      returns[(int64_t)t * A + i] = running;
    }
  }
  return MM_OK;
}

/* marl/mappo.py:220-236 exploration_action / action for a batch (see include/mm_abi.h).  Always uses the
 * bit-reproducible exp of mm_math.h so that the HIP library draws the same actions from the same key. */
int32_t mm_sample_actions(const float *logp, int64_t n, int32_t n_a, uint64_t seed, uint64_t *counter,
                          int32_t *actions, MMStream stream) {
  (void)stream;
  if (!logp || !actions || !counter || n < 0 || n_a < 1 || n_a > 8) return MM_ERR_INVALID_ARG;
  const uint64_t ctr = *counter;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    double cdf[8], acc = 0;
    for (int k = 0; k < n_a; k++) { acc = acc + mmm_exp((double)logp[i * n_a + k]); cdf[k] = acc; }
    uint32_t o[4];
    philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32) ^ 0x53414D50u,
               (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const double u = ((o[0] >> 5) * 67108864.0 + (o[1] >> 6)) / 9007199254740992.0;
    int a = 0;
    for (int k = 0; k < n_a; k++) a += (cdf[k] / cdf[n_a - 1] <= u) ? 1 : 0; /* searchsorted(cdf / cdf[-1], u, "right") */
    actions[i] = a < n_a - 1 ? a : n_a - 1;
  }
  *counter = ctr + 1;
  return MM_OK;
}

/* Actor forward (Model_common.py:5-22) + the sampling above; plain fp32 loops in natural order.  The HIP
 * build sums the same fp32 products in MFMA order, so log-probabilities agree to rounding (tests: 1e-4),
 * and given equal log-probabilities the drawn actions are identical. */
int32_t mm_policy_act(const float *obs, int64_t n, int32_t n_s, const float *W1, const float *b1, const float *W2,
                      const float *b2, const float *W3, const float *b3, int32_t hidden, int32_t n_a, uint64_t seed,
                      uint64_t *counter, int32_t *actions, float *logp, MMStream stream) {
  if (!obs || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !counter || !actions) return MM_ERR_INVALID_ARG;
  if (n < 0 || n_s < 1 || n_s > 32 || hidden != 128 || n_a < 1 || n_a > 8) return MM_ERR_INVALID_ARG;
  float *lp_all = logp ? logp : (float *)malloc((size_t)(n > 0 ? n : 1) * n_a * sizeof(float));
  if (!lp_all) return MM_ERR_INVALID_ARG;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    float h1[128], h2[128], lg[8];
    for (int o = 0; o < 128; o++) {
      float acc = b1[o];
      for (int k = 0; k < n_s; k++) acc = fmaf(W1[o * n_s + k], obs[i * n_s + k], acc);
      h1[o] = acc > 0 ? acc : 0;
    }
    for (int o = 0; o < 128; o++) {
      float acc = b2[o];
      for (int k = 0; k < 128; k++) acc = fmaf(W2[o * 128 + k], h1[k], acc);
      h2[o] = acc > 0 ? acc : 0;
    }
    float mx = -INFINITY;
    for (int o = 0; o < n_a; o++) {
      float acc = 0;
      for (int k = 0; k < 128; k++) acc = fmaf(h2[k], W3[o * 128 + k], acc);
      lg[o] = acc + b3[o];
      mx = lg[o] > mx ? lg[o] : mx;
    }
    float se = 0;
    for (int o = 0; o < n_a; o++) se += expf(lg[o] - mx);
    const float lse = mx + logf(se);
    for (int o = 0; o < n_a; o++) lp_all[i * n_a + o] = lg[o] - lse;
  }
  int32_t rc = mm_sample_actions(lp_all, n, n_a, seed, counter, actions, stream);
  if (!logp) free(lp_all);
  return rc;
}
