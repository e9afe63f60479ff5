// mm_kernels.hip -- fused env.step / reset / observe kernels + the C ABI of include/mm_abi.h.
//
// Mapping (CDNA4, wave64): one LANE per vehicle, the G = pow2 >= N lanes of an env are contiguous
// in a wave ("env group"), 64/G envs per wave, ONE wave per workgroup (64 threads; no s_barrier anywhere in
// the step kernel).  All cross-vehicle reads of an env (neighbour search, collision, observation, rewards,
// shield) happen inside the group -- DPP row permutations, ds_bpermute, or an LDS mailbox the owner posts
// and its partners read by address: no inter-workgroup traffic.  LDS holds per-thread "cold slots"
// (register relief), those mailboxes and, at the end, the transposed observation rows.  One launch = one
// env.step for every env: 3 simulation sub-steps + rewards/info + optional re-spawn + observation.
//
// Sub-step structure (reference: abstract.py:512-532, road.py:269-292):
//   act      per-lane  ControlledVehicle.act (follow_road, steering/speed control); IDM/MOBIL for HDVs
//   predict  per-lane  bicycle integration + lane argmin + corner tests for the nominal steering
//                      (lazily also for the LC-veto steering) -- ALL transcendentals live here
//   sweep              the reference's front-to-back Gauss-Seidel over the env's vehicles (road.py:286)
//                      as a parallel fixed point (classification pass, MASS acceleration rounds, veto
//                      passes) with the literal serial sweep as fallback / validation form; the general
//                      kernels (MIXED) run the serial form only.  See DESIGN.md section 2.
//   collide  per-lane  pair tests by DPP exchange; the rare order-dependent speed fix-up is serial
// Also here: reset / observe / stand-alone shield / QP kernels, the categorical sampler and the f32-MFMA
// policy kernel of the rollout counterpart.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "mm_device.h"
#define MM_COUNTS_FN __host__ __device__ inline
#include "../../include/mm_counts.h"

using namespace mm;

struct DevCfg {
  int env_kind, shield, nsub, T, action_masking, auto_reset, obs_f64, N, E, debug_flags, n_hdv;
  double dt, collision_reward, high_speed_reward, headway_cost, headway_time, merging_lane_cost;
  double rs_lo, rs_hi, eta, tau;
  double inv_dt, rs_span, rs_ispan, rs_span2, rs_ispan2;  // host-computed reciprocals for div_c (correctly rounded 1/d)
  int agent_reward;
  int steer_vel;  // lateral_control == "steer_vel" (v1 CAVs; handled by the MIXED = general instantiations)
  int *err;       // device error latch of the handle (MM_LATCH_* bits), read by mm_poll_errors
  int traffic_density, mixed_traffic, num_cav;  // per-episode vehicle-count draw (MergeEnv._num_vehicles); 0 = fixed counts
};
// conditions the reference raises from inside step(); a launch can only latch them (include/mm_abi.h: mm_poll_errors)
constexpr int kMetricsAccBit = 30;  // internal bit of DevCfg::debug_flags (set by dev_cfg, never by the caller): metrics partials accumulate
#define MM_LATCH_QP_BOUNDS 1  // CBFType.check_bounds, cbf.py:87-96
#define MM_LATCH_BAD_ACTION 2 // DiscreteMetaAction.act: self.actions[action] KeyError, action.py:194-196
#define MM_LATCH_BAD_QP 4     // mm_shield_qp: G is not of the form get_G builds
#define MM_LATCH_INTERNAL 8   // a loop guard of the kernels fired (never expected; reported as MM_ERR_DEVICE)
enum { MM_LW_STEP = 0, MM_LW_QP = 1, MM_LW_SHIELD = 2, MM_LW_COUNT = 3 };  // latch words of a handle (MMHandle_::dev_err)
struct DevState {
  double *F;
  uint8_t *B;
  int32_t *I;
  uint64_t *seeds;
  long long A;
  int E, N;
};

// Hand-off buffer of the SPLIT interior-point step (device memory owned by the handle, never the caller's): with
// qp_solver = MM_QP_IPM a CAV-only shielded batch steps as nsub + 1 launches of the "phase" form of step_kernel (lane per
// vehicle: act / predict / commit / collide / rewards) with one launch of sweep_kernel (lane per ENV: the front-to-back
// shield sweep with its interior-point QPs) after each act / predict.  Planes are [field][vehicle a][env e] with the env
// index innermost: a sweep wave (64 consecutive envs) reads "field f of vehicle a" as one coalesced 512-byte line.
// (What the sweep needs of the vehicles' pre-step STATE -- pose, speed, g.vx, previous safe acceleration, history records --
// it reads from the state planes themselves: the phase kernel stores them before the sweep launch anyway.)
enum {
  SW_CPSI, SW_ACCN,                                              // cos of the pre-step heading, the clipped nominal acceleration
  SW_AX, SW_AY, SW_AH, SW_ACPSI, SW_AGVX, SW_ASTEER, SW_ASPSI,   // candidate A (nominal steering): predicted post-state
  SW_BX, SW_BY, SW_BH, SW_BCPSI, SW_BGVX, SW_BSTEER, SW_BSPSI,   // candidate B (LC-veto steering), only where META says needB
  SW_ACC,                                                        // sweep -> phase kernel: the decided acceleration
  SW_F_COUNT
};
enum { SW_WPK, SW_APK, SW_BPK, SW_META, SW_RES, SW_CLS, SW_I_COUNT };
// SW_META: bit0 live, bit1 shield_on, bit2 needB, bit3 HDV, bits 8..15 Vehicle flags, bits 16..23 hl_action
// (an HDV has no decision: for it SW_RES bit4 says "an ego edited my history record" -- the on-ramp digital twin,
// decentral_layer.py:164-184 -- and SW_ACC then holds the edited state_hist[-2].x)
// SW_RES (written by the sweep for every ego it ran): bit0 ran, bit1 veto, bit2 committed candidate B, bits 8..15 new flags
// SW_CLS: multi_agent_state's slot selection for this vehicle as ego, evaluated by the phase kernel for all vehicles at once
//   (leader | front-adjacent << 4 | rear-adjacent << 8 creation indices, bits 12..14 the slot exists, bit 15 constrain_adj)
//   under the assumption that every vehicle ahead of it in the sweep commits candidate A; the sweep kernel classifies again
//   itself once a vehicle of the env has committed candidate B
struct SweepBuf {
  double *F;       // [SW_F_COUNT][N][Ep]
  int *I;          // [SW_I_COUNT][N][Ep]
  uint8_t *order;  // [N][Ep]: creation index of the live vehicle with sweep rank r (x descending, stable), 0xFF: none
  uint8_t *envf;   // [Ep]: the env is still stepping (no terminal sub-step so far in this policy step)
  uint8_t *urgent; // [Ep]: scheduling hint only (never a result): the env had a QP that ran past 12 iterations in its last sweep
  long long Ep;    // E rounded up to whole waves
  int N;
};
MM_DEV double &sw_f(const SweepBuf &sb, int f, int a, long long e) { return sb.F[((long long)f * sb.N + a) * sb.Ep + e]; }
MM_DEV int &sw_i(const SweepBuf &sb, int f, int a, long long e) { return sb.I[((long long)f * sb.N + a) * sb.Ep + e]; }

// Translation units.  The library is built from this one source compiled six times (Makefile):
//   MM_TU=1  everything except the "general" step kernels (MIXED = true: HDVs and/or steer_vel),
//   MM_TU=2  only those (round 1 built them with conservative SGPR spilling, DESIGN.md "toolchain note"; no longer),
//   MM_TU=3  only the MM_QP_IPM fidelity-mode step kernels (IPM = true: the general kernels with the QP solved by
//            cvxopt's interior-point algorithm, include/mm_qp.h), same flags as TU 2.
//   MM_TU=4 / 5  only the exact-mode / the interior-point step kernels in the 6- / 12-lane rotation layouts (kPow2 below).
// MM_TU=0 (default) is the single-TU form used by the tuning / diagnostic builds.
#ifndef MM_TU
#define MM_TU 0
#endif

// Diagnostic build only (-DMM_STAMPS): per-phase cycle sums (s_memtime) accumulated by lane 0 of each
// wave into a __device__ array the host reads with mm_debug_read_stamps.  Never in the product build.
#ifdef MM_STAMPS
// lane k of every wave accumulates phase k in a register; one plain store per wave at the end (no atomics in flight)
#define MM_STAMP_WAVES (1 << 16)
__device__ unsigned long long g_stamps_w[MM_STAMP_WAVES * 16];
__device__ unsigned long long g_stamps_s[4096 * 8];  // sweep kernel: per wave {top, post, setup, bottom, trips, setup trips, -, -} (cycles / counts)
#define SSTAMP(k)                                                                        \
  do {                                                                                   \
    unsigned long long _t = __builtin_amdgcn_s_memtime();                                \
    if ((threadIdx.x & 63) == (k)) _t_acc += _t - _t_last;                               \
    _t_last = __builtin_amdgcn_s_memtime();                                              \
  } while (0)
#define SCOUNT(k) do { if ((threadIdx.x & 63) == (k)) _t_acc += 1; } while (0)
#define STAMP(k)                                                                         \
  do {                                                                                   \
    unsigned long long _t = __builtin_amdgcn_s_memtime();                                \
    if ((threadIdx.x & 63) == (k)) _t_acc += _t - _t_last;                               \
    _t_last = __builtin_amdgcn_s_memtime();                                              \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#define SSTAMP(k) do {} while (0)
#define SCOUNT(k) do {} while (0)
#endif

struct Veh {
  double x, y, h, v, tspeed;
  double act_steer, act_acc, safe_steer, safe_acc, gvx;
  double h1x, h1vx, h2x, h2vx;  // x / vx of state_hist[-1], [-2]
  double sang;                  // MDPLCVehicle.steering_angle ("steer_vel" lateral control only)
  int lane, tlane, sidx, crashed, hl, flags, hist_len;
  int kind;  // 0 absent, 1 controlled CAV, 2 HDV (for an HDV `gvx` holds the MOBIL timer, see mm_abi.h)
  bool present;
};

// ------------------------------------------------------------------------------------------------
// wave helpers (64 lanes; an env group never straddles a wave)
// ------------------------------------------------------------------------------------------------
MM_DEV int lane_id() { return threadIdx.x & 63; }
// dynamic-source shuffles go through the LDS crossbar (ds_bpermute_b32)
MM_DEV double shfl_d(double v, int src) { return __shfl(v, src, 64); }
MM_DEV int shfl_i(int v, int src) { return __shfl(v, src, 64); }

// Partner exchange inside an env group: lane ^ M for a compile-time M < 16, built from DPP row
// permutations (v_mov_b32_dpp: no LDS round trip, a few cycles) instead of ds_bpermute:
//   M = 1,2,3 quad_perm; 7 row_half_mirror; 8 row_ror:8; 15 row_mirror; the rest are compositions.
// Every lane of the wave must be active where these are used (they sit in uniform control flow).
template <int M>
MM_DEV int dppx_i(int v) {
  static_assert(M >= 1 && M <= 15, "xor mask within a 16-lane DPP row");
  if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);        // quad_perm:[1,0,3,2]
  else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
  else if constexpr (M == 3) return __builtin_amdgcn_update_dpp(0, v, 0x1B, 0xF, 0xF, false);   // quad_perm:[3,2,1,0]
  else if constexpr (M == 7) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);  // row_ror:8
  else if constexpr (M == 15) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); // row_mirror
  else if constexpr (M < 7) return dppx_i<7>(dppx_i<7 - M>(v));
  else if constexpr (M < 12) return dppx_i<8>(dppx_i<M - 8>(v));
  else return dppx_i<15>(dppx_i<15 - M>(v));
}
template <int M>
MM_DEV double dppx_d(double v) {
  long long b = __double_as_longlong(v);
  int lo = dppx_i<M>((int)(unsigned)(b & 0xFFFFFFFFll)), hi = dppx_i<M>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// compile-time loop over the partners a ^ 1 .. a ^ (G-1) of a group
template <int M, int G, class F>
MM_DEV void for_partners_impl(F &f) {
  if constexpr (M < G) {
    f(std::integral_constant<int, M>{});
    for_partners_impl<M + 1, G>(f);
  }
}
template <int G, class F>
MM_DEV void for_partners(F f) { for_partners_impl<1, G>(f); }

// Group layouts.  A power-of-two group (2, 4, 8, 16 lanes) pairs lane a with a ^ m and exchanges by DPP (above).  The
// exact-mode step kernels also come with 6- and 12-lane groups (N = 5..6 and 9..12: ten or five envs per wave
// instead of eight or four, and 5 / 11 partners per loop instead of 7 / 15): there partner m of lane a is (a + m) mod G
// and the exchange goes through the LDS crossbar (ds_bpermute).  A wave holds 64 / G whole groups; with 6 or 12 lanes its
// last 4 lanes belong to no env.  Every exchange in the kernels is written as "what partner m holds", never as a message
// prepared for a specific reader, so the two pairings are interchangeable (the one message-style exchange, the 16-lane
// classification, is not compiled for the rotation layouts: they use the LDS mailbox like the 8-lane groups).
template <int G> constexpr bool kPow2 = (G & (G - 1)) == 0;
#ifndef MM_ROUNDS_LITE
#define MM_ROUNDS_LITE 1
#endif
constexpr bool kRoundsLite = MM_ROUNDS_LITE != 0;
#ifndef MM_VETO_PRIOR
#define MM_VETO_PRIOR 1
#endif
constexpr bool kVetoPrior = MM_VETO_PRIOR != 0;  // parallel form: the previous sub-step's veto as the first guess of the veto passes  // exact-mode MASS rounds: acceleration only, shield_post once (see the rounds)
template <int M, int G>
MM_DEV int pidx(int a) {  // creation index of partner M of the vehicle with index a
  if constexpr (kPow2<G>) return a ^ M;
  else return a + M >= G ? a + M - G : a + M;
}
template <int G>
MM_DEV int pidx_rt(int a, int m) {  // the same for a run-time m in 0..G-1
  if constexpr (kPow2<G>) return a ^ m;
  else return a + m >= G ? a + m - G : a + m;
}
template <int M, int G>
MM_DEV int plane(int lane, int a) {  // lane (or LDS column) of partner M; `lane` = the caller's own lane / column
  if constexpr (kPow2<G>) return lane ^ M;
  else return lane - a + pidx<M, G>(a);
}
template <int G>
MM_DEV int plane_rt(int lane, int a, int m) {
  if constexpr (kPow2<G>) return lane ^ m;
  else return lane - a + pidx_rt<G>(a, m);
}
template <int M, int G>
MM_DEV int px_i(int v, int a) {
  if constexpr (kPow2<G>) return dppx_i<M>(v);
  else return shfl_i(v, plane<M, G>(lane_id(), a));
}
template <int M, int G>
MM_DEV double px_d(double v, int a) {
  if constexpr (kPow2<G>) return dppx_d<M>(v);
  else return shfl_d(v, plane<M, G>(lane_id(), a));
}

template <int G>
MM_DEV unsigned group_ballot(bool p, int gb) {
  unsigned long long b = __ballot(p);
  return (unsigned)((b >> gb) & ((1ull << G) - 1ull));
}
MM_DEV void atomic_min_d(double *addr, double val) {  // CAS loop: valid for any sign, LDS or global
  unsigned long long *p = (unsigned long long *)addr, old = *p, assumed;
  do {
    assumed = old;
    if (__longlong_as_double((long long)assumed) <= val) break;
    old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(val));
  } while (assumed != old);
}
template <int G>
MM_DEV double group_min_d(double v, int a) {
  if constexpr (!kPow2<G>) {  // rotation layout: doubling steps around the ring of G <= 16 lanes (overlapping windows: a minimum does not mind)
    v = fmin(v, px_d<1, G>(v, a));
    v = fmin(v, px_d<2, G>(v, a));
    if constexpr (G > 4) v = fmin(v, px_d<4, G>(v, a));
    if constexpr (G > 8) v = fmin(v, px_d<8, G>(v, a));
    return v;
  }
  if constexpr (G >= 2) v = fmin(v, dppx_d<1>(v));
  if constexpr (G >= 4) v = fmin(v, dppx_d<2>(v));
  if constexpr (G >= 8) v = fmin(v, dppx_d<7>(v));  // after xor 1,2 every quad is uniform: xor 7 == xor 4
  if constexpr (G >= 16) v = fmin(v, dppx_d<8>(v));
  return v;
}

// ------------------------------------------------------------------------------------------------
// state load / store
// ------------------------------------------------------------------------------------------------
MM_DEV void load_veh(const DevState &st, long long i, bool valid, Veh &v, bool with_sang = false) {
  memset(&v, 0, sizeof v);
  v.present = false;
  v.hl = MM_HL_NONE;
  if (!valid) return;
  const long long A = st.A;
  v.kind = st.B[MM_B_KIND * A + i];
  v.present = v.kind != 0;
  if (!v.present) return;
  v.x = st.F[MM_F_X * A + i]; v.y = st.F[MM_F_Y * A + i]; v.h = st.F[MM_F_HEADING * A + i];
  v.v = st.F[MM_F_SPEED * A + i]; v.tspeed = st.F[MM_F_TARGET_SPEED * A + i];
  v.safe_steer = st.F[MM_F_SAFE_STEER * A + i]; v.safe_acc = st.F[MM_F_SAFE_ACC * A + i];
  v.gvx = st.F[MM_F_G_VX * A + i];
  v.h1x = st.F[MM_F_H1_X * A + i]; v.h1vx = st.F[MM_F_H1_VX * A + i];
  v.h2x = st.F[MM_F_H2_X * A + i]; v.h2vx = st.F[MM_F_H2_VX * A + i];
  v.lane = st.B[MM_B_LANE * A + i]; v.tlane = st.B[MM_B_TARGET_LANE * A + i];
  v.sidx = st.B[MM_B_SPEED_INDEX * A + i]; v.crashed = st.B[MM_B_CRASHED * A + i];
  v.hl = st.B[MM_B_HL_ACTION * A + i]; v.flags = st.B[MM_B_FLAGS * A + i];
  v.hist_len = st.B[MM_B_HIST_LEN * A + i];
  if (v.kind == 2) { v.act_steer = v.safe_steer; v.act_acc = v.safe_acc; }  // last IDM action persists
  if (with_sang) v.sang = st.F[MM_F_STEER_ANGLE * A + i];
}
MM_DEV void store_veh(const DevState &st, long long i, const Veh &v, bool with_sang = false) {
  const long long A = st.A;
  if (with_sang) st.F[MM_F_STEER_ANGLE * A + i] = v.sang;
  st.F[MM_F_X * A + i] = v.x; st.F[MM_F_Y * A + i] = v.y; st.F[MM_F_HEADING * A + i] = v.h;
  st.F[MM_F_SPEED * A + i] = v.v; st.F[MM_F_TARGET_SPEED * A + i] = v.tspeed;
  st.F[MM_F_SAFE_STEER * A + i] = v.kind == 2 ? v.act_steer : v.safe_steer;
  st.F[MM_F_SAFE_ACC * A + i] = v.kind == 2 ? v.act_acc : v.safe_acc;
  st.F[MM_F_G_VX * A + i] = v.gvx;
  st.F[MM_F_H1_X * A + i] = v.h1x; st.F[MM_F_H1_VX * A + i] = v.h1vx;
  st.F[MM_F_H2_X * A + i] = v.h2x; st.F[MM_F_H2_VX * A + i] = v.h2vx;
  st.B[MM_B_LANE * A + i] = (uint8_t)v.lane; st.B[MM_B_TARGET_LANE * A + i] = (uint8_t)v.tlane;
  st.B[MM_B_SPEED_INDEX * A + i] = (uint8_t)v.sidx; st.B[MM_B_CRASHED * A + i] = (uint8_t)v.crashed;
  st.B[MM_B_HL_ACTION * A + i] = (uint8_t)v.hl; st.B[MM_B_FLAGS * A + i] = (uint8_t)v.flags;
  st.B[MM_B_HIST_LEN * A + i] = (uint8_t)v.hist_len;
}

// ------------------------------------------------------------------------------------------------
// per-vehicle control (controller.py)
// ------------------------------------------------------------------------------------------------
// controller.py:90-134 ControlledVehicle.act; action 0 LEFT / 2 RIGHT change lanes, else none
// safe_controller.py:84-98 MDPLCVehicle.steering_control in "steer_vel" mode: a steering VELOCITY that
// tracks the scaled-down reference angle (KP_STEER 20, STEER_TARGET_RF 0.125)
MM_DEV double steer_vel_command(double steering_ref, double sang) { return 20 * (steering_ref * 0.125 - sang); }
// st_t: 1/2 tan of the steering command (see steering_control; NaN: not known)
// STEER = false (general kernels): the steering command is left to ONE steering_control call for CAVs and HDVs together,
// after the HDVs have decided their target lanes (steer_lanes below); steering does not enter anything in between
MM_DEV void steer_lane(Veh &v, bool sv, double &st_t) {
  double steer = steering_control(v.x, v.y, v.h, v.v, v.tlane, st_t);
  if (sv) { steer = steer_vel_command(steer, v.sang); st_t = __builtin_nan(""); }
  v.act_steer = clipd(steer, -kPi / 3, kPi / 3);
}
template <bool STEER = true>
MM_DEV void controlled_act(Veh &v, int action, bool sv, double &st_t) {
  if (lane_after_end(v.tlane, v.x)) v.tlane = next_lane(v.tlane, v.x, v.y);  // follow_road :136-144
  if (action == 2 || action == 0) {
    // only road (b,c) has two lanes; elsewhere the clipped candidate is the lane itself
    int cand = v.tlane;
    if (lane_road(v.tlane) == 1) cand = (action == 2) ? MM_LANE_BC1 : MM_LANE_BC0;
    if (lane_reachable(cand, v.x, v.y)) v.tlane = cand;
  }
  v.act_acc = (1 / kTauA) * (v.tspeed - v.v);  // speed_control :189-197
  if constexpr (STEER) steer_lane(v, sv, st_t);
}
// The high-level act of a policy step: action_type.act -> MDPLCVehicle.act / MDPVehicle.act /
// ControlledVehicle.act (safe_controller.py:63-66, controller.py:293-311, :90-125).  Road.act()
// runs ControlledVehicle.act() again right after (abstract.py:516-521) and overwrites the steering
// / acceleration this call computes, so only its side effects are kept: hl_action, speed index /
// target speed, follow_road and the lane-change target.
template <int KIND>
MM_DEV void hl_act(Veh &v, int action) {
  if (KIND == MM_ENV_V1 && action >= 0) v.hl = action;
  if (action == 3 || action == 4) {
    int si = speed_to_index(v.v) + (action == 3 ? 1 : -1);
    v.sidx = si < 0 ? 0 : (si > 4 ? 4 : si);
    v.tspeed = index_to_speed(v.sidx);
  }
  if (lane_after_end(v.tlane, v.x)) v.tlane = next_lane(v.tlane, v.x, v.y);  // follow_road :136-144
  if (action == 2 || action == 0) {
    int cand = v.tlane;
    if (lane_road(v.tlane) == 1) cand = (action == 2) ? MM_LANE_BC1 : MM_LANE_BC0;
    if (lane_reachable(cand, v.x, v.y)) v.tlane = cand;
  }
}

// kinematics.py:143-152 clip_actions (+ safe_controller.py:100-104)
MM_DEV void clip_actions(Veh &v, bool lc_vehicle, double &st_t) {
  if (v.crashed) { v.act_steer = 0; st_t = 0; v.act_acc = -1.0 * v.v; }
  if (v.v > kMaxSpeed) v.act_acc = fmin(v.act_acc, 1.0 * (kMaxSpeed - v.v));
  else if (v.v < -kMaxSpeed) v.act_acc = fmax(v.act_acc, 1.0 * (kMaxSpeed - v.v));
  if (lc_vehicle) v.act_acc = clipd(v.act_acc, kLcMinAcc, kLcMaxAcc);  // MDPLCVehicle only
}

// controller.py:257-267 get_corner("L"/"R") + lane.on_lane of those corners on `lane`
// -> bit0: the front-left corner is off `lane`, bit1: the front-right one
// (the three corner angles alpha + h, -alpha + h as angle sums from sin h, cos h: include/mm_math.h "angle-sum forms")
MM_DEV int corner_flags(double x, double y, double sh, double ch, int lane) {
  double cx = x + (kCornerLen * mmm_cos_sum(MMM_CORNER_SIN, MMM_CORNER_COS, sh, ch));
  double cyL = y - (kCornerLen * mmm_sin_sum(MMM_CORNER_SIN, MMM_CORNER_COS, sh, ch)) + 0.01;
  double cyR = y - (kCornerLen * mmm_sin_sum(-MMM_CORNER_SIN, MMM_CORNER_COS, sh, ch)) + 0.01;
  double s = cx - lane_sx(lane);
  double off = (lane == MM_LANE_KB0) ? kSineAmp * mmm_sin(kSinePuls * s + kSinePhase) : 0.0;
  bool lon = (-kVehLength <= s && s < lane_len(lane) + kVehLength);
  double rL = cyL - lane_sy(lane), rR = cyR - lane_sy(lane);
  if (lane == MM_LANE_KB0) { rL = rL - off; rR = rR - off; }
  return (!(fabs(rL) <= kLaneWidth / 2 + 0 && lon) ? 1 : 0) | (!(fabs(rR) <= kLaneWidth / 2 + 0 && lon) ? 2 : 0);
}
// "pose code" of a (pre- or post-step) pose as the shield reads it: closest lane | its next_lane << 3 |
// corner bits << 6.  Kept packed in one VGPR end to end (messages, LDS park, classification) instead of
// two ints and two lane masks.
MM_DEV int pose_code(int lane, int nl, int off) { return lane | nl << 3 | off << 6; }

// Predicted post-state of Vehicle.step for a given steering (kinematics.py:122-141,
// safe_controller.py:151-172): everything that does not depend on the acceleration.
struct Cand {
  double x, y, h, gvx, cpsi, spsi;  // cpsi / spsi: cos / sin of the new heading
  int lane;  // closest lane of the post-state
  int pk;    // its pose code (SHIELDED only)
};
// CORNERS: the front-corner bits are read by MASS only (constrain_adj, can_abort_lc): HSS kernels skip them
// Device arithmetic of the step (the oracle's math mode 1 evaluates the same expressions): with sin / cos of the current
// heading carried along (sh, ch), the reference's eleven trigonometric calls -- arctan(1/2 tan delta), cos / sin(psi + beta),
// sin(beta), cos(psi' + beta), cos(psi') and the three corner angles -- become ONE sincos of the steering angle and ONE of
// the new heading plus angle sums; beta itself is never formed (include/mm_math.h, "angle-sum forms").
// GENERAL: the instantiation can meet a steering angle that did not come out of steering_control (kernels with HDVs / steer_vel)
template <int KIND, bool SHIELDED, bool CORNERS = SHIELDED, bool GENERAL = true>
MM_DEV Cand predict(const Veh &v, double steer, double st_t, double sh, double ch, double dt, bool sv = false) {
  Cand c;
  // "steer_vel" (safe_controller.py:124-150): the slip angle comes from the steering-angle STATE and the
  // heading advances by d_heading without the dt factor (sic, :135)
  // st_t: 1/2 tan(steer) as steering_control left it (NaN: unknown -- a persisting IDM action, steer_vel -- and the
  // general sincos runs)
  double ht = st_t, sb, cb;
  if constexpr (GENERAL) {
    if (__any(sv || st_t != st_t)) {
      double s2, c2;
      mmm_sincos(sv ? v.sang : steer, &s2, &c2);
      if (sv || st_t != st_t) ht = 1.0 / 2 * (s2 / c2);
    }
  }
  mmm_slip_sincos(ht, &sb, &cb);  // beta = atan(1/2 tan(delta)): its sin and cos
  double vx = v.v * mmm_cos_sum(sh, ch, sb, cb), vy = v.v * mmm_sin_sum(sh, ch, sb, cb);
  c.x = v.x + vx * dt;
  c.y = v.y + vy * dt;
  const double d_heading = MM_DIVC(v.v * sb, 2.5);  // / (LENGTH / 2)
  c.h = v.h + (sv ? d_heading : d_heading * dt);
  mmm_sincos(c.h, &c.spsi, &c.cpsi);
  c.gvx = (KIND == MM_ENV_V1) ? mmm_cos_sum(c.spsi, c.cpsi, sb, cb) : 0.0;
  c.lane = closest_lane(c.x, c.y, c.h);  // on_state_update kinematics.py:154-159
  c.pk = c.lane;
  if (SHIELDED) c.pk = pose_code(c.lane, next_lane(c.lane, c.x, c.y), CORNERS ? corner_flags(c.x, c.y, c.spsi, c.cpsi, c.lane) : 0);
  return c;
}

// decentral_layer.py:23-39 is_adj_lane(vehicle, lane2) given vehicle.lane and its next_lane.  Only road
// (b,c) has two lanes, so "same road and ids differ by one" means the pair {bc0, bc1}: bc0 sees bc1
// at -1 (to its right), bc1 sees bc0 at +1.
MM_DEV int adj_pair(int l1, int l2) {
  return (l1 == MM_LANE_BC0 && l2 == MM_LANE_BC1) ? -1 : ((l1 == MM_LANE_BC1 && l2 == MM_LANE_BC0) ? 1 : 0);
}
MM_DEV int adj_lane(int l1, int nl1, int l2) {
  const int f = adj_pair(l1, l2);
  return f != 0 ? f : adj_pair(nl1, l2);
}
// The same relation as 2-bit codes (0: none, 1: +1, 3: -1) in packed 6-entry tables, for the classification
// loops (one table build per ego instead of four compare chains per pair):
//   adj_row(l1) >> 2*l2 & 3 = code of adj_pair(l1, l2);   adj_col(l2) >> 2*l1 & 3 = code of adj_pair(l1, l2)
MM_DEV unsigned adj_row(int l1) { return l1 == MM_LANE_BC0 ? (3u << (2 * MM_LANE_BC1)) : (l1 == MM_LANE_BC1 ? (1u << (2 * MM_LANE_BC0)) : 0u); }
MM_DEV unsigned adj_col(int l2) { return l2 == MM_LANE_BC1 ? (3u << (2 * MM_LANE_BC0)) : (l2 == MM_LANE_BC0 ? (1u << (2 * MM_LANE_BC1)) : 0u); }

struct QpTrace {
  double rows, a, h0, h1, h2, h3, d, margin;
};

// ------------------------------------------------------------------------------------------------
// device reset: merge_env_v1.py:265-364 with the Philox stream documented in DESIGN.md
// ------------------------------------------------------------------------------------------------
MM_DEV void init_vehicle(Veh &v) {  // kinematics.py:36-53, controller.py:35-50,277-291, safe_controller.py:27-61
  v.lane = closest_lane(v.x, v.y, v.h);
  v.tlane = v.lane;
  v.act_steer = v.act_acc = 0;
  v.safe_steer = v.safe_acc = 0;
  if (v.kind == 2) {  // IDMVehicle.__init__ behavior.py:42-53: target_speed = speed, timer = (sum(pos) * pi) % 1
    v.sidx = 0;
    v.tspeed = v.v;
    v.gvx = py_mod((v.x + v.y) * kPi, 1.0);
  } else {
    v.sidx = speed_to_index(v.v);
    v.tspeed = index_to_speed(v.sidx);
    v.gvx = __builtin_nan("");
  }
  v.h1x = v.h1vx = v.h2x = v.h2vx = 0;
  v.sang = 0;  // safe_controller.py:54
  v.crashed = 0; v.hl = MM_HL_NONE; v.flags = 0; v.hist_len = 0;
}
// MergeEnv._num_vehicles (merge_env_v1.py:180-211, :476-495): block 64 of the episode's stream, word 0 -> num_CAV,
// word 1 -> num_HDV, uniform over 3 values each; traffic_density 0 keeps the given (fixed) counts
MM_DEV void episode_counts(const DevCfg &c, uint64_t seed, uint32_t episode, int &n_cav, int &n_hdv) {
  if (c.traffic_density <= 0) return;
  uint32_t w[4];
  rng_block(seed, episode, 64u, w);
  mm_counts_from_draw(c.traffic_density, c.mixed_traffic, c.num_cav, (int)(((uint64_t)w[0] * 3u) >> 32), (int)(((uint64_t)w[1] * 3u) >> 32),
                      &n_cav, &n_hdv);  // include/mm_counts.h (shared with the oracle and the configuration check)
}
MM_DEV int spawn_vehicle(Veh &v, int a, int n_cav, int n_hdv, uint64_t seed, uint32_t episode) {
  uint32_t r[12], blk[4];
  rng_block(seed, episode, 0, &r[0]);
  rng_block(seed, episode, 1, &r[4]);
  rng_block(seed, episode, 2, &r[8]);
  const int n_s = (n_cav != 1) ? n_cav / 2 : (int)(r[0] & 1u);
  const int n_m = n_cav - n_s;
  const int n_sh = (n_hdv != 1) ? n_hdv / 2 : (int)((r[0] >> 1) & 1u);
  // creation order: CAV main, CAV ramp, HDV main, HDV ramp; HDV slots continue the CAV shuffle
  bool main_road;
  int k;  // index into the road's shuffled slot list
  if (a < n_s) { main_road = true; k = a; }
  else if (a < n_cav) { main_road = false; k = a - n_s; }
  else if (a < n_cav + n_sh) { main_road = true; k = n_s + (a - n_cav); }
  else { main_road = false; k = n_m + (a - n_cav - n_sh); }
  int slots[6];
#pragma unroll
  for (int i = 0; i < 6; i++) slots[i] = (main_road ? 10 : 5) + 50 * i;
  int chosen = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {  // partial Fisher-Yates == choice(replace=False)
    if (i <= k) {
      const uint32_t ri = main_road ? r[i] : r[6 + i];
      int j = i + (int)(((uint64_t)ri * (uint32_t)(6 - i)) >> 32);
      int si = slots[i], sj = 0;
#pragma unroll
      for (int q = 0; q < 6; q++) if (q == j) sj = slots[q];
#pragma unroll
      for (int q = 0; q < 6; q++) if (q == j) slots[q] = si;
      slots[i] = sj;
      if (i == k) chosen = sj;
    }
  }
  rng_block(seed, episode, 3u + (uint32_t)a, blk);
  double speed = u53(blk[0], blk[1]) * 2 + 25;
  double noise = u53(blk[2], blk[3]) * 8 - 4;
  v.x = chosen + noise;
  v.y = main_road ? 0.0 : 10.5;
  v.h = 0;
  v.v = speed;
  v.kind = a < n_cav ? 1 : 2;
  v.present = true;
  init_vehicle(v);
  return n_m;
}

// ------------------------------------------------------------------------------------------------
// observation (envs/common/observation.py:181-273) + action mask (abstract.py:219-240)
// ------------------------------------------------------------------------------------------------
template <int G, int KIND, bool LEND = false>
MM_DEV void observe(const DevCfg &c, const Veh &v, int a, int gb, long long i, bool valid, void *obs,
                    uint8_t *avail, float *stage = nullptr, const double *sincos_h = nullptr) {
  constexpr int F = (KIND == MM_ENV_V1) ? 6 : 5;
  constexpr int S = 5 * F;
  const bool ctrl = v.present && v.kind != 2;  // only controlled vehicles observe / have an action mask
  // the wave's LDS scratch: first the mailbox of the neighbour gather (5 doubles per lane), then the float32 row staging
  float *sw;
  if constexpr (LEND) {
    sw = stage + (threadIdx.x >> 6) * (64 * S);  // the caller lends LDS it no longer needs
  } else {
    __shared__ float s_obs[4][64 * S];
    sw = s_obs[threadIdx.x >> 6];
  }
  static_assert(64 * S * sizeof(float) >= 5 * 64 * sizeof(double), "the mailbox must fit in the row staging");
  if (obs) {
    const int lane = lane_id();
    double sps, cps;
    if (sincos_h) { sps = sincos_h[0]; cps = sincos_h[1]; }  // the step kernel carries sin / cos of the heading along
    else mmm_sincos(v.h, &sps, &cps);
    const double vx = v.v * cps, vy = v.v * sps;  // Vehicle.velocity kinematics.py:215-217
    const double sx = lane_sx(v.lane);
    // Every lane posts what its neighbours read of it (x, y, vx, vy, heading) in the wave's mailbox; the rows of the 4
    // nearest are then GATHERED from their owners' columns by lane index instead of being dragged through the partner
    // loop as 4 x 5 select chains (one v_cndmask pair per double, partner and row: 280 of them at G = 8).
    double *mb = (double *)sw;
    mb[0 * 64 + lane] = v.x; mb[1 * 64 + lane] = v.y; mb[2 * 64 + lane] = vx; mb[3 * 64 + lane] = vy;
    if (KIND == MM_ENV_V1) mb[4 * 64 + lane] = v.h;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pass 1: sort keys of close_vehicles_to (road.py:257-267): |lane_distance_to|, inf if not within 180 m
    double key[G];
    key[0] = 0;
    for_partners<G>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const int pl = plane<m, G>(lane, a);
      double px = mb[0 * 64 + pl], py = mb[1 * 64 + pl];
      bool pp = px_i<m, G>((int)v.present, a) != 0;
      double dx = px - v.x, dy = py - v.y;
      bool close = pp && (dx * dx + dy * dy) < kT180;  // norm < PERCEPTION_DISTANCE, sqrt-free
      key[m] = close ? fabs((px - sx) - (v.x - sx)) : INFINITY;
    });
    // pass 2: the 4 nearest in stable order become rows 1..4: sel[q] = the partner offset m of row q (0: none)
    int sel[4] = {0, 0, 0, 0};
    int rank[G];  // position of partner m in the stable order: (key, creation index) is a strict total order, so each
#pragma unroll    // unordered pair is compared once and credited to the one that comes later
    for (int m = 0; m < G; m++) rank[m] = 0;
#pragma unroll
    for (int m = 1; m < G; m++)
#pragma unroll
      for (int m2 = m + 1; m2 < G; m2++) {
        const bool m2_first = key[m2] < key[m] || (key[m2] == key[m] && pidx_rt<G>(a, m2) < pidx_rt<G>(a, m));
        rank[m] += m2_first ? 1 : 0;
        rank[m2] += m2_first ? 0 : 1;
      }
    for_partners<G>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const bool use = key[m] < INFINITY;
#pragma unroll
      for (int q = 0; q < 4; q++) sel[q] = (use && rank[m] == q) ? m : sel[q];
    });
    double row[4][F - 1];
    bool have[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      have[q] = sel[q] != 0;
      const int pl = plane_rt<G>(lane, a, sel[q]);  // (no row: my own column, masked below)
      row[q][0] = mb[0 * 64 + pl] - v.x; row[q][1] = mb[1 * 64 + pl] - v.y;
      row[q][2] = mb[2 * 64 + pl] - vx; row[q][3] = mb[3 * 64 + pl] - vy;
      if (KIND == MM_ENV_V1) {
        double ph = mb[4 * 64 + pl];
        if (c.steer_vel) {  // MDPLCVehicle.to_dict under "steer_vel" (safe_controller.py:75-81):
          const int pkind = shfl_i(v.kind, pl);  // a CAV neighbour's heading is relative to the observer's
          if (pkind == 1) ph = ph - v.h;
        }
        row[q][F - 2] = ph;
      }
    }
    // the mailbox is read: the same LDS now stages the rows
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // normalize_obs :181-193 via utils.lmap :16-18 (no clip); ranges :171-176, :238-239
    const double lo[5] = {-5.0 * 30, -12, -1.5 * 30, -1.5 * 30, -kPi / 2};
    const double span[5] = {300.0, 24.0, 90.0, 90.0, kPi / 2 - (-kPi / 2)};  // x[1] - x[0] of lmap
    const double ispan[5] = {1.0 / 300.0, 1.0 / 24.0, 1.0 / 90.0, 1.0 / 90.0, 1.0 / (kPi / 2 - (-kPi / 2))};
    auto lmap = [&](double val, int f) { return -1 + div_c((val - lo[f]) * (1 - (-1)), span[f], ispan[f]); };
    const double ego[5] = {v.x, v.y, vx, vy, v.h};
    auto emit = [&](auto put) {
      put(0, ctrl ? 1.0 : 0.0);
#pragma unroll
      for (int f = 0; f < F - 1; f++) put(1 + f, ctrl ? lmap(ego[f], f) : 0.0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool hq = ctrl && have[q];
        put((q + 1) * F, hq ? 1.0 : 0.0);
#pragma unroll
        for (int f = 0; f < F - 1; f++)
          put((q + 1) * F + 1 + f, hq ? lmap(row[q][f], f) : 0.0);
      }
    };
    if (c.obs_f64) {
      if (valid) emit([&](int k, double val) { ((double *)obs)[i * (5 * F) + k] = val; });
    } else {
      // float32 rows: stage the wave's rows in LDS, then write them as ONE contiguous run per wave
      // (a lane's 5F floats are 100/120 B apart from its neighbour's: direct stores would touch 64
      // different cache lines per instruction and tripled the measured WRITE_SIZE).
      const int slot = (lane / G) * c.N + a;  // valid lanes of a wave are contiguous in agent index
      if (valid) emit([&](int k, double val) { sw[slot * S + k] = (float)val; });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // first env of this wave (lane l of the launch holds env l / G: a shift, not a 64-bit division by N; rotation layouts:
      // 64 / G whole groups per wave)
      const long long wave_lane0 = (long long)blockIdx.x * blockDim.x + threadIdx.x - lane;
      const long long e0 = kPow2<G> ? wave_lane0 / G : (wave_lane0 >> 6) * (64 / G);
      long long nenv = (long long)c.E - e0;
      nenv = nenv < 0 ? 0 : (nenv > 64 / G ? 64 / G : nenv);
      const int total = (int)nenv * c.N * S;
      float *dst = (float *)obs + e0 * c.N * S;
      if constexpr (S % 2 == 0) {
        for (int t = lane; t < total / 2; t += 64) ((float2 *)dst)[t] = ((const float2 *)sw)[t];
      } else {
        for (int t = lane; t < total; t += 64) dst[t] = sw[t];
      }
    }
  }
  // action mask: with masking on, the reference's `[[0]*n_a]*n` aliases every row (abstract.py:202,475)
  // so each agent gets the OR over all agents; with masking off every action is available.
  unsigned bits = 0x1F;
  if (c.action_masking) {
    unsigned mine = 0;
    if (ctrl) {
      mine = 1u << 1;
      if (v.lane == MM_LANE_BC1 && lane_reachable(MM_LANE_BC0, v.x, v.y)) mine |= 1u << 0;
      if (v.lane == MM_LANE_BC0 && lane_reachable(MM_LANE_BC1, v.x, v.y)) mine |= 1u << 2;
      if (v.sidx < 4) mine |= 1u << 3;
      if (v.sidx > 0) mine |= 1u << 4;
    }
    bits = 0;
#pragma unroll
    for (int k = 0; k < 5; k++)
      if (group_ballot<G>((mine >> k) & 1u, gb)) bits |= 1u << k;
  }
  if (valid && avail) {
#pragma unroll
    for (int k = 0; k < 5; k++) avail[i * 5 + k] = ctrl ? (uint8_t)((bits >> k) & 1u) : (uint8_t)0;
  }
}

// ------------------------------------------------------------------------------------------------
// HDVs: IDM + MOBIL (vehicle/behavior.py:74-266), mixed traffic only (MIXED kernels)
// ------------------------------------------------------------------------------------------------
struct Body {  // what IDM reads of a vehicle / object
  bool present, is_object;
  double x, y, h, v, tspeed;
  int lane;
  // posted once by the vehicle itself instead of being re-evaluated by every (ego, partner) pair: sin / cos of its heading
  // (the step kernel carries them along: the same mmm_sincos of the same value) and its lateral offset of the sine lane kb0,
  // kSineAmp * sin(kSinePuls * (x - 220) + kSinePhase) (only read when the ego's lane is kb0)
  double sh, ch, koff;
};
MM_DEV double desired_gap(const Body &ego, const Body &front) {  // behavior.py:141-156 (projected)
  constexpr double ab = 15.0;  // -COMFORT_ACC_MAX * COMFORT_ACC_MIN
  const double es = ego.sh, ec = ego.ch, fs = front.sh, fc = front.ch;
  const double evx = ego.v * ec, evy = ego.v * es, fvx = front.v * fc, fvy = front.v * fs;
  const double dv = (evx - fvx) * ec + (evy - fvy) * es;
  // (the divisor 2 sqrt(15) is a compile-time constant: Markstein's constant-divisor form, bit-identical to the division)
  constexpr double den = 0x1.efbdeb14f4edap+2;  // 2 * np.sqrt(15.0) = 7.745966692414834 (sqrt correctly rounded, * 2 exact)
  static_assert(den * den > 4 * ab * (1 - 1e-15) && den * den < 4 * ab * (1 + 1e-15), "2 sqrt(ab)");
  return 10.0 + ego.v * 1.5 + div_c(ego.v * dv, den, 1.0 / den);
}
MM_DEV double idm_acceleration(const Body &ego, const Body &front) {  // behavior.py:111-139
  if (!ego.present || ego.is_object) return 0;
  const double target = not_zero(ego.tspeed);
  double q = fmax(ego.v, 0) / target;
  double q2 = q * q;
  double acc = 3.0 * (1 - q2 * q2);  // np.power(., 4) as (x^2)^2 (oracle math mode 1)
  if (front.present) {
    const double sx = lane_sx(ego.lane);
    const double d = (front.x - sx) - (ego.x - sx);  // ego.lane_distance_to(front)
    const double g = desired_gap(ego, front) / not_zero(d);
    acc -= 3.0 * (g * g);
  }
  return acc;
}
// one candidate of road.neighbour_vehicles(lane) (road.py:352-381): keeps the running front / rear by INDEX (creation index;
// obstacle: 99, it is scanned last); ties follow the reference's loop order.  The chosen bodies are gathered from their owners
// after the scan (Body copies under four conditions per candidate were 1.3 k instructions of the mixed-traffic kernels).
struct NbSel {
  double s_front, s_rear;
  int i_front, i_rear;  // -1: none
};
// the scanned lane's frame, evaluated once per ego instead of once per candidate
struct LaneFrame {
  double sx, sy, s_hi;  // lane start, lane_len + LENGTH
  bool sine;            // kb0: the lateral offset follows the sine
};
MM_DEV LaneFrame lane_frame(int lane) { return {lane_sx(lane), lane_sy(lane), lane_len(lane) + kVehLength, lane == MM_LANE_KB0}; }
MM_DEV void neighbour_update(const LaneFrame &lf, double s_me, bool present, double bx, double by, double bkoff, int idx, NbSel &n) {
  double s_v = bx - lf.sx, lat = by - lf.sy;
  if (lf.sine) lat = lat - bkoff;  // (= kSineAmp * sin(kSinePuls * s_v + kSinePhase), posted by the body's owner)
  const bool on = present && fabs(lat) <= kLaneWidth / 2 + 1 && -kVehLength <= s_v && s_v < lf.s_hi;
  const bool fr = on && s_me <= s_v && (n.i_front < 0 || s_v < n.s_front || (s_v == n.s_front && idx > n.i_front));
  const bool rr = on && s_v < s_me && (n.i_rear < 0 || s_v > n.s_rear || (s_v == n.s_rear && idx < n.i_rear));
  n.s_front = fr ? s_v : n.s_front; n.i_front = fr ? idx : n.i_front;
  n.s_rear = rr ? s_v : n.s_rear; n.i_rear = rr ? idx : n.i_rear;
}

// ------------------------------------------------------------------------------------------------
// CBF shield of ONE vehicle given its selected neighbours (safe_action_hss / safe_action_mass,
// decentral_layer.py:290-518 / :521-764; rows of cbf.py:262-322,374-422; exact-KKT QP).
// Scalar form of the reference's 8-vectors: only the x / speed components enter the CBF rows.
// ------------------------------------------------------------------------------------------------
struct Neigh {  // the three slots of multi_agent_state (:85-257) after the obstacle overrides
  bool has_ol, has_oa, has_oar, constrain_adj;
  double ol_x, ol_vx, ol_acc, ol_g;  // leader: state_hist[-2].x / .vx, its action and g.vx (MASS)
  double oa_x, oa_vx, oa_acc, oa_g;  // front adjacent
  double oar_x, oar_vx;              // rear adjacent: current to_dict()
  // literal sweep: the products g2*u2 / g4*u4 of the CBF rows as the PUBLISHING lane computed them from its
  // own record (same operands, same order as below), so a stage fetches 2 values per slot instead of 4
  double ol_gu, oa_gu;
};
struct ShieldOut {
  double acc, us0;  // derived acceleration, u_safe[0]
  bool veto;        // "Avoiding lane change" (:501-506 / :739-744)
  bool lon_safe, lon_invariant;  // CBF_AV.update_status cbf.py:341-351
  bool optimal, bounds;          // sol["status"] != "unknown" (cbf.py:140); check_bounds would raise (cbf.py:87-96)
  double hw_num, hw_den;         // vehicle.set_min_headway = hw_num / hw_den (decentral_layer.py:466,700); the
                                 // division is left to the trace writer (it is wanted only when tracing)
  int flags;
  QpTrace qt;
};
template <bool MASS>
MM_DEV unsigned obstacle_override(Neigh &nb, double x, double y) {  // decentral_layer.py:213-246
  // (a replaced slot has u = max(0, 0 + acc dt) = 0 for acc in {0, -12.5}, so its g*u product is +0)
  unsigned replaced = 0;  // bit0: leader slot, bit1: adjacent slot now hold the obstacle at (420, 4)
  if (!(x > kObstX)) {
    const double ady = fabs(kObstY - y);
    if ((!nb.has_ol || kObstX <= nb.ol_x) && ady <= 2) {
      nb.has_ol = true; nb.ol_x = kObstX; nb.ol_vx = 0.0; nb.ol_gu = 0.0;
      if (MASS) { nb.ol_acc = 0; nb.ol_g = 0; }
      replaced |= 1u;
    }
    if ((!nb.has_oa || kObstX <= nb.oa_x) && (2 < ady && ady <= 4)) {
      nb.has_oa = true; nb.oa_x = kObstX; nb.oa_vx = 0.0; nb.oa_gu = 0.0;
      if (MASS) { nb.oa_acc = 0; nb.oa_g = 0; nb.constrain_adj = false; }
      replaced |= 2u;
    }
  }
  return replaced;
}
// Everything of the shield that does not depend on the neighbours' DECIDED accelerations
// (evaluated once per classification); shield_dyn() finishes it per fixed-point round.
struct ShieldStatic {
  double evx, u0, g0, g2, g4, g6, u6, v_min, v_max, h1, h2;
  double px_lon, px_lona, px_lonr, q_lon, q_lona, q_lonr, hls_lona, hls_lonr;
  double base0, base3;  // h0 / h3 without their g*u terms
  bool cadj, can_abort_lc;
};
template <bool MASS>
MM_DEV ShieldStatic shield_static(const DevCfg &c, const Veh &v, double cpsi, int pk_self, const Neigh &nb) {
  const double dt = c.dt, eta = c.eta;
  ShieldStatic s;
  s.v_min = v.v + kLcMinAcc * dt;
  if (MASS) s.v_min = s.v_min > 0 ? s.v_min : 0;  // :798
  s.v_max = v.v + kLcMaxAcc * dt;
  double evx = v.v * cpsi;
  s.evx = evx > 1 ? evx : 1;  // :307-309
  const double x_e = v.x;
  const double x_ol = nb.has_ol ? nb.ol_x : x_e + kPerception + 1;
  const double x_oa = nb.has_oa ? nb.oa_x : x_e + kPerception + 1;
  const double x_oar = nb.has_oar ? nb.oar_x : x_e - kPerception - 1;
  s.g0 = v.gvx * dt;
  s.g2 = MASS ? nb.ol_g * dt : 1 * dt;
  s.g4 = MASS ? nb.oa_g * dt : 1 * dt;
  s.g6 = 1 * dt;
  double sv_oar = nb.has_oar ? nb.oar_vx : 0;
  sv_oar = sv_oar + kCbfAccHi * dt;
  sv_oar = sv_oar > 1 ? sv_oar : 1;
  const double buffer = (kCbfAccHi + 0.1) * dt * c.tau;
  const double sd0 = s.evx * c.tau + kVehLength + buffer;
  const double sd2 = sv_oar * c.tau + kVehLength + buffer;
  double u0 = s.evx + v.act_acc * dt;  // simplified_control (:60-77)
  s.u0 = u0 > 0 ? u0 : 0;
  s.u6 = 0;
  if (nb.has_oar) { double u6 = nb.oar_vx + kCbfAccHi * dt; s.u6 = u6 > 0 ? u6 : 0; }
  s.cadj = MASS && nb.constrain_adj;
  s.q_lon = -kVehLength - sd0;
  s.q_lona = -kVehLength - sd0;
  s.q_lonr = -kVehLength - sd2;
  if (s.cadj) s.q_lona = -kVehLength - sd0 - kAdjBuffer;
  s.px_lon = x_ol - x_e; s.px_lona = x_oa - x_e; s.px_lonr = x_e - x_oar;
  s.base0 = s.px_lon + (eta - 1) * s.px_lon + eta * s.q_lon;
  s.base3 = s.px_lona + (eta - 1) * s.px_lona + eta * s.q_lona;
  s.h1 = s.v_max - s.u0;
  s.h2 = -s.v_min + s.u0;
  s.hls_lona = s.px_lona + s.q_lona;
  s.hls_lonr = s.px_lonr + s.q_lonr;
  s.can_abort_lc = (pk_self >> 6) == 0;  // neither front corner off the lane (:728-736, pre-step pose)
  return s;
}
// The neighbours' predicted speed u = max(0, vx + acc dt) and its CBF-row product (g dt) u, from one record.
template <bool MASS>
MM_DEV double slot_gu(double vx, double acc, double g, double dt) {
  double u = vx + acc * dt;
  u = u > 0 ? u : 0;
  return (MASS ? g * dt : 1 * dt) * u;
}
// The CBF rows of the QP that depend on the neighbours' decided accelerations: h0 (leader), h3 (front-adjacent, MASS with
// constrain_adj) and the two g*u products they are built from.
// PRE: nb.ol_gu / nb.oa_gu already hold those products (computed by the lanes that own the records)
struct ShieldRows { double g2u2, g4u4, h0, h3; };
template <bool PRE>
MM_DEV ShieldRows shield_rows(const DevCfg &c, const ShieldStatic &s, const Neigh &nb) {
  const double dt = c.dt;
  ShieldRows r;
  if (PRE) {
    r.g2u2 = nb.has_ol ? nb.ol_gu : 0.0;
    r.g4u4 = nb.has_oa ? nb.oa_gu : 0.0;
  } else {
    double u2 = 0, u4 = 0;
    if (nb.has_ol) { u2 = nb.ol_vx + nb.ol_acc * dt; u2 = u2 > 0 ? u2 : 0; }
    if (nb.has_oa) { u4 = nb.oa_vx + nb.oa_acc * dt; u4 = u4 > 0 ? u4 : 0; }
    r.g2u2 = s.g2 * u2;
    r.g4u4 = s.g4 * u4;
  }
  const double g0u0 = s.g0 * s.u0;
  r.h0 = s.base0 + (-g0u0 + r.g2u2);
  r.h3 = __builtin_nan("");
  if (s.cadj) r.h3 = s.base3 + (-g0u0 + r.g4u4);
  return r;
}
// exact KKT point of min 1/2(d^2 + e^2 + 1e18 s^2) s.t. a d - s <= hc, lo <= d <= hi.  (d is clipped into
// [v_min - u0, v_max - u0] and v_min <= v_max, so check_bounds cannot fire in this mode.)
MM_DEV double qp_exact(const ShieldStatic &s, const ShieldRows &r) {
  const double hc = (s.cadj && r.h3 < r.h0) ? r.h3 : r.h0;
  double d;
  if (s.g0 > 0) d = fmin(0.0, hc / s.g0);
  else if (s.g0 < 0) d = fmax(0.0, hc / s.g0);
  else d = 0.0;
  return fmin(fmax(d, -s.h2), s.h1);
}
// Everything after the QP returned d (cbf.py:134-161, decentral_layer.py:493-518 / :721-764): status, is_lc_allowed, the
// veto, the collaboration flags and the derived acceleration.  `optimal`: sol["status"] != "unknown"; `check`: evaluate
// check_bounds (only the interior-point iterate can leave the bounds)
template <bool MASS>
MM_DEV ShieldOut shield_post(const DevCfg &c, const Veh &v, const ShieldStatic &s, const ShieldRows &r, double d, bool optimal, bool check) {
  const double dt = c.dt, eta = c.eta;
  ShieldOut o;
  o.optimal = optimal;
  o.bounds = check && ((s.u0 + d) - 0.001 > s.v_max || (s.u0 + d) + 0.001 < s.v_min);
  double us0 = s.u0 + d;
  {  // update_status (cbf.py:341-351) on u_status = [u_safe (QP), u_ll[2:]]
    const double hls_lon = s.px_lon + s.q_lon;
    const double hlds_lon = s.px_lon + ((-s.g0) * us0 + r.g2u2) + s.q_lon;
    o.lon_safe = hls_lon >= -1e-6;
    o.lon_invariant = (hlds_lon + (eta - 1) * hls_lon) >= -1e-6;
    o.hw_num = s.px_lon - kVehLength; o.hw_den = s.evx;
  }
  // is_lc_allowed (cbf.py:324-339)
  const double hlds_lona = s.px_lona + ((-s.g0) * us0 + r.g4u4) + s.q_lona;
  const double hlds_lonr = s.px_lonr + (s.g0 * us0 + (-s.g6) * s.u6) + s.q_lonr;
  const double inv_lona = hlds_lona + (eta - 1) * s.hls_lona, inv_lonr = hlds_lonr + (eta - 1) * s.hls_lonr;
  const bool lc_allowed = ((s.hls_lona >= 0) && inv_lona >= 0) && ((s.hls_lonr >= 0) && inv_lonr >= 0);
  int fl = v.flags & MM_FLAG_COLLABORATE_ADJ;
  if (s.cadj) fl |= MM_FLAG_IS_COLLABORATING;
  bool veto;
  if (!MASS) {
    veto = !lc_allowed;
  } else {
    veto = s.can_abort_lc && !lc_allowed;
    if (!veto && (v.hl == 2 || v.hl == 0) && v.v < kStoppingSpeed) us0 = s.u0;  // :746-750
    if (inv_lona >= -1e-6) fl |= MM_FLAG_COLLABORATE_ADJ;  // can_collaborate_adj cbf.py:424-430
    else fl &= ~MM_FLAG_COLLABORATE_ADJ;
  }
  if (!veto) fl |= MM_FLAG_IS_LC_SAFE;
  o.acc = div_c(us0 - s.evx, dt, c.inv_dt);  // derived_acceleration :80-82
  o.us0 = us0; o.veto = veto; o.flags = fl;
  o.qt.rows = s.cadj ? 4 : 3; o.qt.a = s.g0; o.qt.h0 = r.h0; o.qt.h1 = s.h1; o.qt.h2 = s.h2; o.qt.h3 = r.h3; o.qt.d = d;
  o.qt.margin = fmin(fmin(s.hls_lona, inv_lona), fmin(s.hls_lonr, inv_lonr));
  return o;
}
// IPM: solve the QP by cvxopt's interior-point algorithm (MM_QP_IPM), here as ONE start-to-stop solve of this lane's QP
// (literal sweep, stand-alone shield kernel; the parallel form of the step kernel runs the wave-wide loop instead);
// `run_qp` = this lane's result is wanted (the literal sweep evaluates every lane at every stage and keeps the ego's
// only: the others must not iterate on garbage)
template <bool MASS, bool PRE = false, bool IPM = false>
MM_DEV ShieldOut shield_dyn(const DevCfg &c, const Veh &v, const ShieldStatic &s, const Neigh &nb, bool run_qp = true) {
  const ShieldRows r = shield_rows<PRE>(c, s, nb);
  double d;
  bool optimal = true;
  if constexpr (IPM) {
    // the iterate cvxopt.solvers.qp stops at (cbf.py:134), its status (:140) and check_bounds (:87-96)
    d = 0.0;
    if (run_qp) {
      double slack;
      int iters;
      optimal = mm_qp_ipm_cbf(s.g0, r.h0, s.h1, s.h2, r.h3, s.cadj ? 4 : 3, &d, &slack, &iters) != 0;
    }
  } else {
    d = qp_exact(s, r);
  }
  return shield_post<MASS>(c, v, s, r, d, optimal, IPM && run_qp);
}
template <bool MASS, bool PRE = false, bool IPM = false>
MM_DEV ShieldOut shield_eval(const DevCfg &c, const Veh &v, double cpsi, int pk_self, const Neigh &nb, bool run_qp = true) {
  const ShieldStatic s = shield_static<MASS>(c, v, cpsi, pk_self, nb);
  return shield_dyn<MASS, PRE, IPM>(c, v, s, nb, run_qp);
}

// Relation of vehicle `o` (as the ego currently sees it) to the ego: the branch conditions of the
// loop in multi_agent_state (decentral_layer.py:104-211).  cls: 0 none, 1 leader, 2 front-adjacent,
// 3 rear-adjacent, 4 on-ramp HDV twin.  cflag: constrain_adj candidate = the relevant front corner of `o` left its lane.
struct Rel {
  double key;  // |ego.lane_distance_to(o)| or inf when o is not within the perception distance
  int cls;
  bool cflag;
};
MM_DEV Rel relate(double ex, double ey, int epk, bool other, double ox, double oy, double oh, int opk, bool o_hdv = false) {
  const int elane = epk & 7, enl = (epk >> 3) & 7, olane = opk & 7, onl = (opk >> 3) & 7;
  // written with non-short-circuit logic on purpose: it runs once per (ego, partner) pair and should
  // compile to compares and selects, not to exec-mask branches
  Rel r;
  const double dx = ox - ex, dy = oy - ey;
  const bool close = other & ((dx * dx + dy * dy) < kT180);  // road.py:259-262 (norm < 180, sqrt-free)
  const double esx = lane_sx(elane);
  const double ld = (ox - esx) - (ex - esx);  // kinematics.py:161-173
  r.key = close ? fabs(ld) : INFINITY;
  // v_a = adj_lane(elane, enl, olane): row of elane, entries it leaves empty filled from the row of enl
  const unsigned rowE = adj_row(elane), rowN = adj_row(enl);
  const unsigned nzE = (rowE | rowE >> 1) & 0x555u;
  const unsigned rowEN = rowE | (rowN & ~(nzE * 3u));
  const unsigned colE = adj_col(elane);  // a_v = adj_lane(olane, onl, elane)
  const unsigned va_c = (rowEN >> (2 * olane)) & 3u;
  const unsigned g1 = (colE >> (2 * olane)) & 3u, g2 = (colE >> (2 * onl)) & 3u;
  const unsigned av_c = g1 != 0 ? g1 : g2;
  const bool head = (dy < 0) ? (oh > 0.037) : (oh < -0.037);
  const bool appr = (!(ld < 0)) & (fabs(dy) <= 3.5) & head;  // :46-57
  const bool adj = (!appr) & ((va_c | av_c) != 0);
  const bool same = (elane == olane) | (olane == enl);  // is_same_lane :15-20
  // :162-184 HDV on the merging lane next to an ego on ab0: "digital twin" slot (class 4)
  const bool twin = o_hdv & (!adj) & (elane == MM_LANE_AB0) & (olane == MM_LANE_KB0) & (ld >= 0);
  const bool lead = (!adj) & (!twin) & (same | appr) & (ld > 0);
  const int cls = adj ? ((ld < 0) ? 3 : 2) : (twin ? 4 : (lead ? 1 : 0));
  r.cls = close ? cls : 0;
  r.cflag = ((opk >> (((va_c == 3u) | (av_c == 1u)) ? 6 : 7)) & 1) != 0;  // (v_a == -1 or a_v == 1; :146-152 which front corner of `o`)
  return r;
}

// trace of the in-step shield call's status dict and min_headway (written where they are produced so that
// they need not stay in registers)
MM_DEV void trace_status(double *t, long long A, const ShieldOut &o) {
  t[MM_T_STATUS * A] = (double)(MM_ST_RAN | (o.optimal ? MM_ST_IS_OPTIMAL : 0u) | (o.lon_safe ? MM_ST_IS_SAFE : 0u) |
                                (o.lon_invariant ? MM_ST_IS_INVARIANT : 0u) | (o.bounds ? MM_ST_QP_BOUNDS : 0u));
  t[MM_T_HEADWAY * A] = o.hw_num / o.hw_den;
}

// ------------------------------------------------------------------------------------------------
// sweep kernel of the split interior-point step: ONE LANE PER ENV
// ------------------------------------------------------------------------------------------------
// The reference steps an env's vehicles front to back and each MDPLCVehicle.step solves its own QP after the vehicles
// ahead of it have moved (road.py:286, safe_controller.py:106-185, decentral_layer.py:126-135): per env the QPs of a
// sub-step form a chain, about one of them is ready at a time, and an interior-point solve is ~6 dependent iterations
// (100 for the 0.4 % that run to cvxopt's iteration cap).  With a lane per VEHICLE (the fused kernels) a wave holds 8 envs
// and iterates with ~8 of its 64 lanes busy.  Here a lane owns a whole ENV and walks its vehicles in sweep order -- the
// literal Gauss-Seidel sweep of the reference, no fixed point, no veto passes: classify the others as the ego sees them
// now (multi_agent_state), build the CBF rows, solve the QP (include/mm_qp.h, resumable: every lane of the wave is at its
// own iteration of its own QP), evaluate status / veto / flags (shield_post), publish the ego's committed post-state for
// the egos behind it.  64 envs per wave, every lane busy until its env is through.
// What the others see of a vehicle ("view": pose, pose code, history record x, its g*u product, longitudinal speed) lives in
// LDS, one column per env: built at the top of the launch from the state planes the phase kernel stored (pre-step view),
// replaced by the post-step view when the vehicle commits.  All lanes of a wave read "vehicle o" of their envs together; the
// ego's own fields and the three selected neighbours are gathers (conflict-free: whole 512-byte rows apart).
// NV: compile-time bound on N (the classification keys live in registers).
#ifndef MM_SWEEP_GATE_T
#define MM_SWEEP_GATE_T 48
#endif
#ifndef MM_SWEEP_GATE_W
#define MM_SWEEP_GATE_W 8
#endif
#ifndef MM_SWEEP_SPEC_AT  // the iteration at which a still-running QP is taken to be on its way to the cap (see "SPECULATION")
#define MM_SWEEP_SPEC_AT 13
#endif
#ifndef MM_SWEEP_GATE_WU  // ... and a lane on the launch's critical path (see `urgent`) at most this many
#define MM_SWEEP_GATE_WU 3
#endif
// MIXED: the env may hold IDM / MOBIL vehicles.  They run no shield; when the sweep passes one it publishes its post-step view
// (Road.step has stepped it), and an ego on ab0 that finds one on the ramp beside it shifts that vehicle's history record in
// place -- the reference's "digital twin" (decentral_layer.py:162-184) -- which the later egos and the next sub-step see.
template <int NV, bool MASS, bool MIXED = false>
__global__ __launch_bounds__(64, 1) void sweep_kernel(DevCfg c, DevState st, SweepBuf sb, int k, double *trace) {
  const long long A = st.A;
  enum { W_X = 0, W_Y, W_H, W_HX, W_GU, W_VX };  // the view planes of s_w
  const int ln = threadIdx.x;
  const long long e = (long long)blockIdx.x * 64 + ln;
  const int N = sb.N;
  const double dt = c.dt;
  // The views of the env's vehicles live in LDS for the whole sweep, one column per lane (= env): every ego reads all of them,
  // and a global round trip per read is ~2 us for a lone wave.  [field][vehicle][lane]: the lanes of a wave read "vehicle o"
  // together and a gather "vehicle j(lane)" differs by whole 512-byte rows -- no bank conflicts either way.
  __shared__ double s_w[6][NV][64];  // W_X .. W_VX
  // (16-bit planes: with NV = 11 the wave's LDS stays under 40 KB = four single-wave workgroups per CU, one per SIMD)
  __shared__ unsigned short s_pk[NV][64], s_meta[NV][64], s_cls[NV][64];
  // SW_META packed into 9 bits: live | shield_on | needB | Vehicle flags (3 bits) << 3 | hl_action (0..4, 7 = None) << 6
  auto pack_meta = [](int m) { const int hl = (m >> 16) & 255; return (unsigned short)((m & 7) | ((m >> 8) & 7) << 3 | (hl > 4 ? 7 : hl) << 6 | ((m >> 3) & 1) << 9 | ((m >> 4) & 1) << 10); };  // (bit 9: HDV, bit 10: the candidate the phase kernel's slot selection assumed is B)
  enum { PH_SETUP = 0, PH_RUN = 1, PH_FIN = 2, PH_DONE = 3 };
  int phase = e < c.E ? PH_SETUP : PH_DONE;
  unsigned long long ord_lo = ~0ull, ord_hi = ~0ull;  // sweep order, one byte per rank (0..7 | 8..15)
  // every vehicle's own shield inputs stay in registers (an ego picks its set by a select chain): a global load at the top
  // of a setup would stall the lone wave for ~2 us
  double own_v[NV], own_gvx[NV], own_acc[NV], own_cpsi[NV];
#pragma unroll
  for (int o = 0; o < NV; o++) own_v[o] = own_gvx[o] = own_acc[o] = own_cpsi[o] = 0.0;
  if (phase != PH_DONE) {
#pragma unroll
    for (int o = 0; o < NV; o++) {
      if (o < N) {
        // the pre-step view the literal sweep starts from (the fused kernel's serial form: wx .. wgu), from the state planes
        const long long i = e * N + o;
        own_v[o] = st.F[MM_F_SPEED * A + i]; own_gvx[o] = st.F[MM_F_G_VX * A + i];
        own_acc[o] = sw_f(sb, SW_ACCN, o, e); own_cpsi[o] = sw_f(sb, SW_CPSI, o, e);
        s_w[W_X][o][ln] = st.F[MM_F_X * A + i]; s_w[W_Y][o][ln] = st.F[MM_F_Y * A + i]; s_w[W_H][o][ln] = st.F[MM_F_HEADING * A + i];
        s_w[W_HX][o][ln] = st.F[MM_F_H2_X * A + i];
        const int meta_o = sw_i(sb, SW_META, o, e);
        const bool hdv_o = MIXED && (meta_o & 8) != 0;  // (an HDV's record: its action is taken as full braking, g.vx as 1)
        s_w[W_GU][o][ln] = slot_gu<MASS>(st.F[MM_F_H2_VX * A + i], (MASS && !hdv_o) ? st.F[MM_F_SAFE_ACC * A + i] : kCbfAccLo, hdv_o ? 1.0 : own_gvx[o], dt);
        s_w[W_VX][o][ln] = own_v[o] * own_cpsi[o];
        s_pk[o][ln] = (unsigned short)sw_i(sb, SW_WPK, o, e);
        s_meta[o][ln] = pack_meta(meta_o);
        s_cls[o][ln] = (unsigned short)sw_i(sb, SW_CLS, o, e);
        const unsigned long long ob = sb.order[(long long)o * sb.Ep + e];
        if (o < 8) ord_lo = (ord_lo & ~(0xFFull << (8 * o))) | ob << (8 * o);
        else ord_hi = (ord_hi & ~(0xFFull << (8 * (o - 8)))) | ob << (8 * (o - 8));
      } else {
        s_meta[o][ln] = 0;
      }
    }
  }
  // the views as the phase kernel left them, once more (a failed speculation: the env is swept again; what the sweep never
  // writes -- own_*, s_meta, s_cls, the order -- is still in place)
  auto reload_views = [&]() {
    for (int o = 0; o < N; o++) {
      const long long i = e * N + o;
      const double gvx = st.F[MM_F_G_VX * A + i];
      s_w[W_X][o][ln] = st.F[MM_F_X * A + i]; s_w[W_Y][o][ln] = st.F[MM_F_Y * A + i]; s_w[W_H][o][ln] = st.F[MM_F_HEADING * A + i];
      s_w[W_HX][o][ln] = st.F[MM_F_H2_X * A + i];
      s_w[W_GU][o][ln] = slot_gu<MASS>(st.F[MM_F_H2_VX * A + i], MASS ? st.F[MM_F_SAFE_ACC * A + i] : kCbfAccLo, gvx, dt);
      s_w[W_VX][o][ln] = st.F[MM_F_SPEED * A + i] * sw_f(sb, SW_CPSI, o, e);
      s_pk[o][ln] = (unsigned short)sw_i(sb, SW_WPK, o, e);
    }
  };
  int r = 0;         // next sweep rank to look at
  int ego = 0;       // creation index of the vehicle whose QP this lane is solving
  int ego_meta = 0;
  bool sing = false, opt = false;
  // the phase kernel's slot selection (SW_CLS) holds for an ego as long as every vehicle ahead of it committed candidate A
  // (general kernels carry no parallel-form classification: the sweep always classifies itself there)
  bool dirty = MIXED || (c.debug_flags & 1) != 0;  // (debug_flags bit0: classify here always -- the validation form of this kernel)
  unsigned hdv_stepped = 0, hdv_shifted = 0;  // per vehicle: the HDV's published view is the post-step one / its record was edited since
  bool hss_collab = false;                    // HSS: vehicle.is_collaborating = cbf.constrain_adj of a twin (:181)
  double e_v = 0;  // the ego's speed (shield_post and the published view read it again after the QP)
  // what the ego publishes when it commits: fetched while its QP iterates
  double p_h1x = 0, p_h1vx = 0, p_ax = 0, p_ay = 0, p_ah = 0, p_ag = 0, p_ac = 0;
  int p_apk = 0;
  ShieldStatic ss;
  ShieldRows rw = {0, 0, 0, 0};
  MMQpState q;
  MMQpRes rs = {0, 0, 0, 0, 0, 0};
  {  // (every field is written before it is read: by the first setup of the lane; zeros keep the compiler's dataflow simple)
    ss.evx = ss.u0 = ss.g0 = ss.g2 = ss.g4 = ss.g6 = ss.u6 = ss.v_min = ss.v_max = ss.h1 = ss.h2 = 0;
    ss.px_lon = ss.px_lona = ss.px_lonr = ss.q_lon = ss.q_lona = ss.q_lonr = ss.hls_lona = ss.hls_lonr = ss.base0 = ss.base3 = 0;
    ss.cadj = ss.can_abort_lc = false;
    q.a = q.h0 = q.h1 = q.h2 = q.h3 = q.resz0 = q.x0 = q.x2 = q.gap = 0;
    q.s0 = q.s1 = q.s2 = q.s3 = q.z0 = q.z1 = q.z2 = q.z3 = 0;
    q.d0 = q.d1 = q.d2 = q.d3 = q.di0 = q.di1 = q.di2 = q.di3 = q.l0 = q.l1 = q.l2 = q.l3 = 1.0;
    q.ax0 = q.ax2 = 0; q.m4 = 0; q.iters = 0;
  }
#ifdef MM_STAMPS
  unsigned long long _t_last = __builtin_amdgcn_s_memtime(), _t_acc = 0;
#endif
  int since = 0;       // trips since lanes were last served (wave-uniform)
  bool serve = false;
  // An env with a QP on its way to the iteration cap (past 12 iterations: none of the QPs that converge takes that long) is on
  // the launch's critical path -- the launch ends with its slowest env -- so (HSS) its lane is served after at most
  // MM_SWEEP_GATE_WU trips, and with it whoever else is waiting.  Such envs tend to stay that way for several sub-steps (the
  // same vehicle keeps braking at its bound): the flag of the last sweep is the prior.  A scheduling hint: results do not
  // depend on it.  Measured at 65 536 x 8 (ms per step, without / with): HSS 1.70 / 1.48 - 1.52 for WU = 1 .. 6; MASS 1.93 /
  // 1.96 - 2.00 (more envs are urgent at a time and the extra services cost the wave more than the slowest lane gains): HSS only.
  bool urgent = phase != PH_DONE && sb.urgent[e] != 0, slow_now = false;
  // SPECULATION on the QPs that run towards the iteration cap.  Every QP of this path that converges does so within 11
  // iterations; one still running at iteration MM_SWEEP_SPEC_AT (13) is infeasible without the slack and ends -- whether the
  // frozen-iterate certificate stops it at ~29 or it runs to 100 -- at d = -h2 with status "unknown" (include/mm_qp.h).  Waiting for
  // that is what the launch's slowest envs do (chains of such QPs: a hard-braking leader makes its follower's row infeasible
  // too).  So the lane ASSUMES that outcome, goes on with the env, and puts the QP itself -- state and residuals -- into the
  // wave's verification queue in LDS.  When every env of the wave is through, the queued QPs are iterated to their real stop
  // side by side, one per lane (top / bottom only: ~16 of the kernel's cheapest trips instead of ~16 per QP on its env's
  // critical path).  A stop that is not ("unknown", x0 == the assumed value bit for bit) marks the owner's env: it is swept
  // again from the state planes without speculation (everything it wrote is written again).  Results are therefore those of
  // the literal loop by construction; debug_flags bit4 makes every assumption wrong (the rollback under test), bit5 switches
  // speculation off.  (Built first with idle lanes verifying DURING the sweep: fewer trips, but the merge of a second QP
  // state into the main loop cost more than it saved -- profiles/r04/variants.jsonl.)
  constexpr int VQ_CAP = NV <= 8 ? 16 : 1;  // (a full queue: the lane simply iterates on; LDS: 40 KB per wave in all)
  __shared__ double s_vq[VQ_CAP][40];
  __shared__ int s_vq_own[VQ_CAP], s_redo[64];
  s_redo[ln] = 0;
  int vq_head = 0, vq_tail = 0;  // ring indices, wave-uniform
  // (MASS with up to 8 vehicles only.  HSS has no chains -- a vehicle's QP reads nobody's decision -- and its sweep lost more to
  // the extra code than the shorter waits gained: 1.49 -> 1.60 ms at 65 536 x 8; with NV = 11 the LDS has room for four queue
  // entries only: 3.06 -> 3.12 ms at traffic_density 3)
  constexpr bool kSpec = MASS && NV <= 8 && !MIXED;
  bool nospec = !kSpec || (c.debug_flags & 32) != 0;
  int help_owner = 0;
  double help_pred = 0.0;
  for (int trip = 0;; trip++) {
    SCOUNT(4);
    if (trip > 3 * (MM_QP_MAXITERS + 3 + MM_SWEEP_GATE_W) * (NV + 1)) { atomicOr(c.err, MM_LATCH_INTERNAL); break; }  // cannot happen: every trip advances a QP or the rank, or counts towards a service
    // (1) cvxopt's stopping test for the running QPs
    bool push = false;
    if (phase == PH_RUN) {
      const int stop = mm_qp_top(&q, &rs);
      {
        if (stop || sing) { phase = PH_FIN; opt = stop == 1 && !sing; }
        else if (kSpec && q.iters == MM_SWEEP_SPEC_AT && !nospec && q.a > 0.0) {
          // Assume only where the outcome is all but certain (a wrong assumption costs the wave a second sweep of the env):
          // the lower bound is not degenerate (h2 = 0: a stopped vehicle -- that QP converges, slowly), the CBF row is clearly
          // infeasible (predicted slack s* = -a h2 - h_c > 1e-4: a smaller one converges, "optimal", after 20 - 90
          // iterations), and -h2 keeps clear of the lattice on which the dual residual could vanish (mm_qp_frozen's condition
          // (4), from the predicted multiplier 1e18 s*).  20 M random QPs of the shield's shapes: 10.1 M assumed, 6 wrong at
          // s* > 1e-6 (all below 2e-5).
          const double hc = (q.m4 && q.h3 < q.h0) ? q.h3 : q.h0, sstar = (-q.a * q.h2) - hc;
          if (q.h2 > 1e-3 && sstar > 1e-4) {
            const int ex = ilogb(q.a * 1e18 * sstar) - 2;
            const double y = ldexp(q.h2, 53 - ex);
            push = fabs(y) < 0x1p+51 && ldexp(fabs(y - rint(y)), ex - 53) > 1e-6;
          }
        }
        if (q.iters > 12) { urgent = true; slow_now = true; }
      }
    }
#ifndef MM_SWEEP_NOSPEC_CODE
    if constexpr (kSpec) {
      const unsigned long long pm = __ballot(push);
      if (pm) {  // hand the QPs over (as many as the queue has room for; the others simply iterate on)
        const int room = VQ_CAP - (vq_tail - vq_head), idx = __popcll(pm & ((1ull << ln) - 1ull));
        if (push && idx < room) {
          double *t = s_vq[(vq_tail + idx) % VQ_CAP];
          double pred = -q.h2;
          if (c.debug_flags & 16) pred = pred + 0x1p-30;
          t[0] = q.a; t[1] = q.h0; t[2] = q.h1; t[3] = q.h2; t[4] = q.h3; t[5] = q.resz0; t[6] = q.x0; t[7] = q.x2; t[8] = q.gap;
          t[9] = q.s0; t[10] = q.s1; t[11] = q.s2; t[12] = q.s3; t[13] = q.z0; t[14] = q.z1; t[15] = q.z2; t[16] = q.z3;
          t[17] = q.d0; t[18] = q.d1; t[19] = q.d2; t[20] = q.d3; t[21] = q.di0; t[22] = q.di1; t[23] = q.di2; t[24] = q.di3;
          t[25] = q.l0; t[26] = q.l1; t[27] = q.l2; t[28] = q.l3; t[29] = q.ax0; t[30] = q.ax2;
          t[31] = rs.rx0; t[32] = rs.rx2; t[33] = rs.rz0; t[34] = rs.rz1; t[35] = rs.rz2; t[36] = rs.rz3;
          t[37] = pred; t[38] = (double)(q.m4 | q.iters << 1);
          s_vq_own[(vq_tail + idx) % VQ_CAP] = ln;
          q.x0 = pred; opt = false; phase = PH_FIN;  // the assumed outcome: solvers.qp returned (d = -h2, "unknown")
#ifdef MM_STAMPS
          atomicAdd(&g_stamps_s[blockIdx.x % 4096 * 8 + 7], 1ull);
#endif
        }
        const int n = __popcll(pm);
        vq_tail += n < room ? n : room;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
#endif
    SSTAMP(0);
    // Lanes whose QP stopped (or that have none yet) are SERVED -- (2) post / publish, (3) set up the next ego -- together:
    // the wave executes those ~1 k instructions whenever one lane needs them, so a lone finisher waits up to
    // MM_SWEEP_GATE_W trips for company unless MM_SWEEP_GATE_T lanes are waiting or nothing else is running.
    {
      // (n_run: lanes iterating a QP of their OWN env -- a lane that only has helpers for company is served at once)
      const int n_wait = __popcll(__ballot(phase == PH_FIN || phase == PH_SETUP)), n_run = __popcll(__ballot(phase == PH_RUN));
      since += 1;
      serve = n_wait > 0 && (n_wait >= MM_SWEEP_GATE_T || n_run == 0 || since >= MM_SWEEP_GATE_W ||
                             (!MASS && since >= MM_SWEEP_GATE_WU && __any(urgent && (phase == PH_FIN || phase == PH_SETUP))));
      if (serve) since = 0;
    }
    // (2) a QP stopped: everything after solvers.qp returned (cbf.py:134-161, decentral_layer.py:493-518 / :721-764), then
    //     Vehicle.step commits and the egos behind see this vehicle's post-step state
    if (serve && __any(phase == PH_FIN)) {
      if (phase == PH_FIN) {
        Veh v;
        v.v = e_v; v.flags = (ego_meta >> 3) & 7; v.hl = ((ego_meta >> 6) & 7) == 7 ? (int)MM_HL_NONE : (ego_meta >> 6) & 7;  // (all shield_post reads of the vehicle)
        ShieldOut so = shield_post<MASS>(c, v, ss, rw, q.x0, opt, true);
        if (MIXED && hss_collab) so.flags |= MM_FLAG_IS_COLLABORATING;
        if (so.bounds) atomicOr(c.err, MM_LATCH_QP_BOUNDS);
        const bool needB = (ego_meta & 4) != 0, use_B = so.veto && needB;
        if (use_B) {  // (rare: the LC-veto candidate is fetched on demand)
          p_ax = sw_f(sb, SW_BX, ego, e); p_ay = sw_f(sb, SW_BY, ego, e); p_ah = sw_f(sb, SW_BH, ego, e);
          p_ag = sw_f(sb, SW_BGVX, ego, e); p_ac = sw_f(sb, SW_BCPSI, ego, e); p_apk = sw_i(sb, SW_BPK, ego, e);
        }
        double nv = e_v + so.acc * dt;
        nv = nv > 0 ? nv : 0;
        s_w[W_X][ego][ln] = p_ax; s_w[W_Y][ego][ln] = p_ay; s_w[W_H][ego][ln] = p_ah; s_pk[ego][ln] = (unsigned short)p_apk;
        s_w[W_HX][ego][ln] = p_h1x;
        s_w[W_GU][ego][ln] = slot_gu<MASS>(p_h1vx, MASS ? so.acc : kCbfAccLo, p_ag, dt);
        s_w[W_VX][ego][ln] = nv * p_ac;
        sw_f(sb, SW_ACC, ego, e) = so.acc;
        sw_i(sb, SW_RES, ego, e) = 1 | (so.veto ? 2 : 0) | (use_B ? 4 : 0) | (so.flags & 255) << 8;
        if (trace) {
          double *t = trace + (long long)k * MM_T_COUNT * A + e * N + ego;
          trace_status(t, A, so);
          t[MM_T_QP_ROWS * A] = so.qt.rows; t[MM_T_QP_A * A] = so.qt.a;
          t[MM_T_QP_H0 * A] = so.qt.h0; t[MM_T_QP_H1 * A] = so.qt.h1; t[MM_T_QP_H2 * A] = so.qt.h2;
          t[MM_T_QP_H3 * A] = so.qt.h3; t[MM_T_QP_D * A] = so.qt.d; t[MM_T_LC_MARGIN * A] = so.qt.margin;
        }
        dirty = dirty || use_B != (((ego_meta >> 10) & 1) != 0);  // committed the other candidate than the phase kernel's slot selection assumed
        r += 1;
        phase = PH_SETUP;
      }
    }
    SSTAMP(1);
    // (3) next ego of the env: multi_agent_state (decentral_layer.py:85-257) on the views as they stand, the CBF rows, the
    //     initial point of its QP
    if (serve && __any(phase == PH_SETUP)) {
      SCOUNT(5);
      if (phase == PH_SETUP) {
        ego = 0xFF;
        for (; r < N; r++) {  // the next vehicle in sweep order that runs a shield (gate: safe_controller.py:232-239)
          const int o = (int)(((r < 8 ? ord_lo : ord_hi) >> (8 * (r & 7))) & 255);
          if (o == 0xFF) { r = N; break; }
          ego_meta = s_meta[o][ln];
          if (ego_meta & 2) { ego = o; break; }
          if (MIXED && (ego_meta & 0x201) == 0x201) {
            // an HDV steps when the sweep reaches it (no shield): from here on the egos see its post-step pose, and its
            // state_hist[-2] is the record it held as [-1] before (IDMVehicleHist, behavior.py:505-521); its speed entry
            // stays the pre-step one, like the fused kernel's view
            s_w[W_X][o][ln] = sw_f(sb, SW_AX, o, e); s_w[W_Y][o][ln] = sw_f(sb, SW_AY, o, e); s_w[W_H][o][ln] = sw_f(sb, SW_AH, o, e);
            s_pk[o][ln] = (unsigned short)sw_i(sb, SW_APK, o, e);
            s_w[W_HX][o][ln] = st.F[MM_F_H1_X * A + e * N + o];
            s_w[W_GU][o][ln] = slot_gu<MASS>(st.F[MM_F_H1_VX * A + e * N + o], kCbfAccLo, 1.0, dt);
            hdv_stepped |= 1u << o; hdv_shifted &= ~(1u << o);
          }
        }
        if (ego == 0xFF) {
          if constexpr (MIXED) {  // what the egos did to the HDVs' records goes back to the phase kernel
            for (int o = 0; o < N; o++) {
              const int m = s_meta[o][ln];
              if ((m & 0x201) != 0x201) continue;
              const bool ed = ((hdv_stepped & hdv_shifted) >> o) & 1u;
              if (ed) sw_f(sb, SW_ACC, o, e) = s_w[W_HX][o][ln];
              sw_i(sb, SW_RES, o, e) = ed ? 16 : 0;
            }
          }
          phase = PH_DONE;
        } else {
          // what the ego will publish: global loads issued now, consumed when its QP has stopped
          double e_gvx = 0, e_acc = 0, cpsi = 1;
#pragma unroll
          for (int o = 0; o < NV; o++) {
            if (o == ego) { e_v = own_v[o]; e_gvx = own_gvx[o]; e_acc = own_acc[o]; cpsi = own_cpsi[o]; }
          }
          p_h1x = st.F[MM_F_H1_X * A + e * N + ego]; p_h1vx = st.F[MM_F_H1_VX * A + e * N + ego];
          p_ax = sw_f(sb, SW_AX, ego, e); p_ay = sw_f(sb, SW_AY, ego, e); p_ah = sw_f(sb, SW_AH, ego, e);
          p_ag = sw_f(sb, SW_AGVX, ego, e); p_ac = sw_f(sb, SW_ACPSI, ego, e); p_apk = sw_i(sb, SW_APK, ego, e);
          const double ex = s_w[W_X][ego][ln], ey = s_w[W_Y][ego][ln];
          const int epk = s_pk[ego][ln];
          // slot selection: the phase kernel's (all egos of the batch classified at once, lane per vehicle), or -- once a vehicle
          // ahead committed candidate B, so that this ego sees a pose the phase kernel did not assume -- classified here
          const int cw = s_cls[ego][ln];
          int j_ol = cw & 15, j_oa = (cw >> 4) & 15, j_oar = (cw >> 8) & 15;
          Neigh nb;
          nb.has_ol = (cw >> 12) & 1; nb.has_oa = (cw >> 13) & 1; nb.has_oar = (cw >> 14) & 1; nb.constrain_adj = (cw >> 15) & 1;
          if (__any(dirty)) {
            if (dirty) {
              double key[NV];
              double k_ol = INFINITY, k_oa = INFINITY, k_oar = INFINITY;
              bool cadj = false;
              j_ol = j_oa = j_oar = -1;
              unsigned twins = 0;  // vehicles in the digital-twin class (general kernels)
#pragma unroll
              for (int o = 0; o < NV; o++) {
                const bool other = (s_meta[o][ln] & 1) != 0 && o != ego;  // (o >= N: meta 0)
                const Rel rl = relate(ex, ey, epk, other, s_w[W_X][o][ln], s_w[W_Y][o][ln], s_w[W_H][o][ln], s_pk[o][ln],
                                      MIXED && (s_meta[o][ln] & 0x200) != 0);
                key[o] = rl.key;
                if (MIXED && rl.cls == 4) twins |= 1u << o;
                // first in sorted order per class: smaller key, ties by creation index (ascending o: a strict < keeps the earlier one)
                if (rl.cls == 1 && rl.key < k_ol) { k_ol = rl.key; j_ol = o; }
                if (rl.cls == 2 && rl.key < k_oa) { k_oa = rl.key; j_oa = o; cadj = rl.cflag; }
                if (rl.cls == 3 && rl.key < k_oar) { k_oar = rl.key; j_oar = o; }
              }
              // count = 5 of close_vehicles_to: a slot exists only if its vehicle is among the 5 nearest
              int pos_ol = 0, pos_oa = 0, pos_oar = 0;
#pragma unroll
              for (int o = 0; o < NV; o++) {
                pos_ol += (key[o] < k_ol || (key[o] == k_ol && o < j_ol)) ? 1 : 0;
                pos_oa += (key[o] < k_oa || (key[o] == k_oa && o < j_oa)) ? 1 : 0;
                pos_oar += (key[o] < k_oar || (key[o] == k_oar && o < j_oar)) ? 1 : 0;
              }
              nb.has_ol = j_ol >= 0 && pos_ol < 5; nb.has_oa = j_oa >= 0 && pos_oa < 5; nb.has_oar = j_oar >= 0 && pos_oar < 5;
              nb.constrain_adj = MASS && nb.has_oa && cadj;
              if constexpr (MIXED) {
                // digital twins (decentral_layer.py:162-184): EVERY HDV of that class among the 5 nearest gets its record
                // shifted by half the ego's longitudinal speed, in place; the LAST one in sorted order ends up in s_oa (the
                // branch has no `is None` guard) with cbf.constrain_adj = True
                hss_collab = false;
                if (twins) {
                  const double e_vx = s_w[W_VX][ego][ln];  // vehicle.velocity[0] of the ego
                  int best = -1, best_pos = -1;
#pragma unroll
                  for (int o = 0; o < NV; o++) {
                    if ((twins >> o) & 1u) {
                      int pos = 0;
#pragma unroll
                      for (int o2 = 0; o2 < NV; o2++) pos += (key[o2] < key[o] || (key[o2] == key[o] && o2 < o)) ? 1 : 0;
                      if (pos < 5) {
                        s_w[W_HX][o][ln] = s_w[W_HX][o][ln] + 0.5 * e_vx;
                        hdv_shifted |= 1u << o;
                        if (pos > best_pos) { best_pos = pos; best = o; }
                      }
                    }
                  }
                  if (best >= 0) { nb.has_oa = true; j_oa = best; nb.constrain_adj = MASS; hss_collab = !MASS; }
                }
              }
            }
          }
          const int s_ol = nb.has_ol ? j_ol : 0, s_oa = nb.has_oa ? j_oa : 0, s_oar = nb.has_oar ? j_oar : 0;
          nb.ol_x = s_w[W_HX][s_ol][ln]; nb.ol_gu = s_w[W_GU][s_ol][ln];
          nb.oa_x = s_w[W_HX][s_oa][ln]; nb.oa_gu = s_w[W_GU][s_oa][ln];
          nb.oar_x = s_w[W_X][s_oar][ln]; nb.oar_vx = s_w[W_VX][s_oar][ln];
          nb.ol_vx = nb.oa_vx = nb.ol_acc = nb.ol_g = nb.oa_acc = nb.oa_g = 0;  // folded into ol_gu / oa_gu
          obstacle_override<MASS>(nb, ex, ey);
          Veh v;
          v.x = ex; v.v = e_v; v.gvx = e_gvx; v.act_acc = e_acc;  // (all shield_static reads of the vehicle)
          ss = shield_static<MASS>(c, v, cpsi, epk, nb);
          rw = shield_rows<true>(c, ss, nb);
          sing = false;
          // (a singular initial KKT system -- NaN input -- makes cvxopt raise: iterate NaN, "unknown")
          int stop = mm_qp_start(&q, ss.g0, rw.h0, ss.h1, ss.h2, rw.h3, ss.cadj ? 4 : 3) != 0 ? 0 : 2;
          if (!stop) stop = mm_qp_top(&q, &rs);
          phase = stop == 0 ? PH_RUN : PH_FIN;  // (stopped at the initial point: posted in the next trip)
          opt = stop == 1;
        }
      }
    }
    SSTAMP(2);
    // (4) one interior-point iteration of every running QP
    if (!__any(phase != PH_DONE)) {
      // every env is through: the wave now VERIFIES what its lanes assumed -- lane j takes the j-th queued QP and iterates it
      // to its real stop (top / bottom only: the cheapest trips of the kernel, all queued QPs side by side)
      const int nq = kSpec ? vq_tail - vq_head : 0;
      if (kSpec && nq > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bool helping = ln < nq;
        if (helping) {
          const double *t = s_vq[(vq_head + ln) % VQ_CAP];
          q.a = t[0]; q.h0 = t[1]; q.h1 = t[2]; q.h2 = t[3]; q.h3 = t[4]; q.resz0 = t[5]; q.x0 = t[6]; q.x2 = t[7]; q.gap = t[8];
          q.s0 = t[9]; q.s1 = t[10]; q.s2 = t[11]; q.s3 = t[12]; q.z0 = t[13]; q.z1 = t[14]; q.z2 = t[15]; q.z3 = t[16];
          q.d0 = t[17]; q.d1 = t[18]; q.d2 = t[19]; q.d3 = t[20]; q.di0 = t[21]; q.di1 = t[22]; q.di2 = t[23]; q.di3 = t[24];
          q.l0 = t[25]; q.l1 = t[26]; q.l2 = t[27]; q.l3 = t[28]; q.ax0 = t[29]; q.ax2 = t[30];
          rs.rx0 = t[31]; rs.rx2 = t[32]; rs.rz0 = t[33]; rs.rz1 = t[34]; rs.rz2 = t[35]; rs.rz3 = t[36];
          help_pred = t[37];
          const int mi = (int)t[38];
          q.m4 = mi & 1; q.iters = mi >> 1;
          help_owner = s_vq_own[(vq_head + ln) % VQ_CAP];
        }
        vq_head = vq_tail;
        while (__any(helping)) {
          if (helping) {
            const bool ok = mm_qp_bottom(&q, &rs) != 0;  // (the entry holds the state after a stopping test that said "go on")
            const int stop = ok ? mm_qp_top(&q, &rs) : 2;
            if (stop) {  // did it end where its owner assumed?
              if (stop == 1 || __double_as_longlong(q.x0) != __double_as_longlong(help_pred)) s_redo[help_owner] = 1;
              helping = false;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      // nothing left to do -- unless an assumption failed: those envs are swept again, literally
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const bool redo = kSpec && s_redo[ln] != 0;
      if (!kSpec || !__any(redo)) break;
      if (redo) {
#ifdef MM_STAMPS
        atomicAdd(&g_stamps_s[blockIdx.x % 4096 * 8 + 6], 1ull);
#endif
        s_redo[ln] = 0;
        reload_views();
        r = 0; dirty = (c.debug_flags & 1) != 0; nospec = true;
        phase = PH_SETUP;
      }
      continue;
    }
    if (__any(phase == PH_RUN)) {
      if (phase == PH_RUN) sing = mm_qp_bottom(&q, &rs) == 0;  // singular KKT matrix: the iterate stands, status "unknown"
    }
    SSTAMP(3);
  }
  if (e < c.E) sb.urgent[e] = slow_now ? 1 : 0;
#ifdef MM_STAMPS
  if ((threadIdx.x & 63) < 8 && blockIdx.x < 4096) g_stamps_s[blockIdx.x * 8 + (threadIdx.x & 63)] += _t_acc;
#endif
}

// ------------------------------------------------------------------------------------------------
// the fused step kernel
// ------------------------------------------------------------------------------------------------
#ifndef MM_MIN_WAVES
#define MM_MIN_WAVES 2  // 2 waves/SIMD: measured 0.62 ms vs 0.98 (1) / 0.92 (3, spills) at 65536x8 MASS
#endif
// Unshielded kernels are small (7.7 KB LDS per wave): 3 waves/SIMD (168 VGPRs).  Measured at 65536 x 8, round 2:
// 0.157 (2 waves) / 0.133 (3) / 0.135 (4: 128 VGPRs, ~80 B/lane of spills) / 0.165 ms (5).
#ifndef MM_NONE_WAVES
#define MM_NONE_WAVES 3
#endif
// unshielded kernels in groups of at least this many lanes run at 2 waves / SIMD (256 registers): at 168 the 12- / 16-lane
// instantiations spill 56 - 116 VGPRs
#ifndef MM_NONE_WIDE_G
#define MM_NONE_WIDE_G 99
#endif
#ifndef MM_IPM_WAVES
#define MM_IPM_WAVES 2
#endif
// the fused interior-point kernels of CAV-only batches run only up to two fused waves per SIMD (larger batches take the split
// step, steps_split below)
#ifndef MM_IPM_CAV_WAVES
#define MM_IPM_CAV_WAVES 2
#endif
// ... and in groups of 6 lanes and more only up to ONE (steps_split): one wave per SIMD = 512 registers, 6 spilled VGPRs instead
// of 287 in <8,1,2,false,true,false> (8 192 x 8 MASS: 1.117 -> 1.067 ms per step)
#ifndef MM_IPM_WIDE_WAVES
#define MM_IPM_WIDE_WAVES 1
#endif
#ifndef MM_SPLIT_WAVES
#define MM_SPLIT_WAVES 2  // phase form of the step kernel (split interior-point step): 168 registers (3 waves) spill 85 of them
#endif
template <int G, int SHIELD, bool MIXED>
#ifndef MM_GENERAL_NONE_WAVES
#define MM_GENERAL_NONE_WAVES 3  // mixed-traffic unshielded: 0.49 (2 waves) / 0.445 (3) / 0.51 ms (4)
#endif
constexpr int step_min_waves(bool ipm = false) { return SHIELD == MM_SHIELD_NONE ? (G >= MM_NONE_WIDE_G ? 2 : (MIXED ? MM_GENERAL_NONE_WAVES : MM_NONE_WAVES)) : (ipm ? (G >= 6 ? MM_IPM_WIDE_WAVES : (MIXED ? MM_IPM_WAVES : MM_IPM_CAV_WAVES)) : MM_MIN_WAVES); }
// IPM: the MM_QP_IPM fidelity mode (the shield's QP by cvxopt's interior-point algorithm, include/mm_qp.h); carried by
// general (MIXED) instantiations only, which run the literal sweep -- one QP per vehicle per sub-step, as the reference
// TRACE: the per-sub-step trace planes (MMStepOut.trace, tests / profile export) are a compile-time property: the
// production instantiation carries neither the stores nor the eight QP-trace registers per lane
// SPLIT: the "phase" form of the kernel for the split interior-point step (SweepBuf above).  Launch kb = 0 .. nsub runs the
// COMMIT half of sub-step kb - 1 (Vehicle.step with the accelerations / vetoes the sweep kernel decided, collisions,
// terminal test) and then the ACT half of sub-step kb (act, predict candidates A and B, hand the shield's inputs to the
// sweep kernel) -- or, for kb = nsub, the epilogue (rewards, re-spawn, observation).  The state planes carry the vehicle
// state from launch to launch; everything else a sub-step keeps in registers / LDS between its two halves goes through
// the SweepBuf planes.  Same code as the fused form, statement for statement: only the shield sweep between the halves
// is replaced by the hand-off.
template <int G, int KIND, int SHIELD, bool MIXED, bool IPM = false, bool TRACE = false, bool SPLIT = false>
#ifndef MM_STEP_BLOCK
#define MM_STEP_BLOCK 64  // one wave per workgroup: waves of a 256-thread block drifted ~12 % apart and the block held its LDS until the slowest was done (0.355 -> 0.333 ms)
#endif
__global__ __launch_bounds__(MM_STEP_BLOCK, (SPLIT ? MM_SPLIT_WAVES : step_min_waves<G, SHIELD, MIXED>(IPM))) void step_kernel(DevCfg c, DevState st, const int32_t *__restrict__ actions,
                                                   MMStepOut out, double *metrics, SweepBuf sb, int kb) {
  constexpr bool LC = (KIND == MM_ENV_V1);
  constexpr bool SHIELDED = LC && (SHIELD != MM_SHIELD_NONE);
  constexpr bool MASS = (SHIELD == MM_SHIELD_MASS);
  static_assert(!IPM || SHIELDED, "the IPM mode lives in shielded kernels");
  static_assert(!SPLIT || IPM, "the split form exists for the interior-point kernels");
  // Form of the shield sweep.  Every CAV-only shielded kernel (HSS and MASS) runs the parallel fixed-point form with the
  // literal front-to-back sweep compiled in as fallback (a vehicle moving backwards in x) and as the validation form
  // (debug_flags bit0); the general kernels (HDVs / steer_vel) and the MASS IPM kernels carry the literal sweep ONLY
  // (kSerialOnly): the HDV "digital twin" branch is only expressible there.  See DESIGN.md section 2.
#ifdef MM_SERIAL_ALL  // tuning switch: every shielded kernel carries the literal sweep only
  constexpr bool kSerialOnly = true || IPM;
#else
#ifdef MM_SERIAL_MASS  // tuning switch: MASS in the literal form at every size
  constexpr bool kSerialOnly = MIXED || MASS || IPM;
#else
  // (IPM: the parallel form runs the dependency-driven wave-wide interior-point loop below instead of the rounds)
  constexpr bool kSerialOnly = MIXED;
#endif
#endif
  static_assert(kPow2<G> || MM_STEP_BLOCK == 64, "rotation layouts: one wave per workgroup");
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // power-of-two groups tile the launch seamlessly; 6- / 12-lane groups: 64 / G whole groups per wave, its last 4 lanes idle
  // (the idle tail lanes address the wave's last group with a >= G: whatever they read is a real lane's, they own no env and
  // no vehicle slot -- e = E, a >= N -- and no group's ballot window contains them)
  constexpr int kTail = kPow2<G> ? 64 : (64 / G) * G;
  const bool tail = lane_id() >= kTail;
  const int gb = kPow2<G> ? (lane_id() & ~(G - 1)) : (tail ? kTail - G : lane_id() - lane_id() % G);
  const int a = kPow2<G> ? (int)(gtid % G) : lane_id() - gb;
  const long long e = kPow2<G> ? gtid / G : (tail ? (long long)st.E : (gtid >> 6) * (64 / G) + lane_id() / G);
  const bool valid = e < st.E && a < st.N;
  const long long i = e * st.N + a;
  const long long A = st.A;
  const double dt = c.dt;

#ifdef MM_STAMPS
  unsigned long long _t_last = __builtin_amdgcn_s_memtime(), _t_acc = 0;
#endif
  Veh v;
  const bool sv = MIXED && KIND == MM_ENV_V1 && c.steer_vel != 0;
  load_veh(st, i, valid, v, sv);
  int steps = 0, time = 0, n_merge = 0, episode = 0;
  if (e < st.E) {
    steps = st.I[MM_E_STEPS * st.E + e]; time = st.I[MM_E_TIME * st.E + e];
    // n_merge / episode are only needed by the epilogue: loaded there, not held across the sub-steps
  }
  if (!MIXED && v.kind == 2) v.kind = 1;  // CAV-only kernels: the host guarantees there are no HDVs
  const bool hdv = MIXED && v.kind == 2;
  const bool ctrl = v.present && !hdv;  // controlled vehicle (MDPVehicle / MDPLCVehicle)
  int action = (valid && ctrl) ? actions[i] : 1;
  if ((unsigned)action > 4u) {  // self.actions[action]: KeyError in the reference (action.py:194-196) -> latched, acts as IDLE
    if (!SPLIT || kb == 0) atomicOr(c.err, MM_LATCH_BAD_ACTION);
    action = 1;
  }
  const unsigned present_bits = group_ballot<G>(v.present, gb);
  const unsigned ctrl_bits = group_ballot<G>(ctrl, gb);
  const int n_veh = __popc(present_bits), n_ctrl = __popc(ctrl_bits);
  if (!SPLIT || kb == 0) steps += 1;  // abstract.py:457 (split form: the first launch of the policy step counts it)

  // derived per-vehicle registers the shield keeps current across sub-steps
  double cpsi = 1.0, spsi = 0.0;  // cos / sin of my current heading (every kernel: the bicycle step is built on them)
  double st_t = __builtin_nan("");  // 1/2 tan(v.act_steer) once steering_control has produced it (NaN: not known -- an IDM action persisting from the last launch -- and predict() runs the general sincos)
  if (v.present) mmm_sincos(v.h, &spsi, &cpsi);
  int pk_self = v.lane;  // pose code of my current (pre-step) pose
  if (SHIELDED && v.present) pk_self = pose_code(v.lane, next_lane(v.lane, v.x, v.y), MASS ? corner_flags(v.x, v.y, spsi, cpsi, v.lane) : 0);

  // Register relief: lane-private values that are written once and read rarely live in LDS ("cold"
  // slots, one column per thread) instead of being spilled to scratch by the compiler (measured: each
  // 100 B/lane of scratch costs ~8 % of the kernel): the LC-veto candidate B, the history records,
  // the previous safe action, the target speed, and the 7..15 sort keys of the classification pass.
  // (C_GU0 / C_GU1: parallel form only -- the g*u product of my pre- / post-step record for the askers' gather)
  // (C_PRE, C_MSGW: parallel form only -- my pre-step pose (x, y, h, pose code) and my rank word, the mailbox the
  // partners' classification reads instead of a DPP exchange per field)
  enum { C_B = 0, C_H1X = 8, C_H1VX, C_H2X, C_H2VX, C_SSTEER, C_SACC, C_TSPEED, C_A = 15, C_GU0 = 23, C_GU1 = 24, C_PRE = 25, C_MSGW = 29 };
  // (16-lane groups keep the DPP exchange: 5 more slots would cost them a wave per CU -- measured 0.456 -> 0.504 ms at N = 12)
  constexpr bool kMailbox = G <= 8 || !kPow2<G>;  // (the DPP form below exchanges reader-specific messages: power-of-two groups only)
  // (rotation layouts: the rank word rides in the unused high half of the pose-code slot -- 40 slots = 20 KB at G = 12, the
  // eighth wave of a CU)
  constexpr bool kPackMsg = !kPow2<G>;
  constexpr int kColdB = kMailbox ? (kPackMsg ? 29 : 30) : 25;
  // candidates occupy 8 slots each (x, y, h, pose code [int], cos h, steering, g.vx, sin h); unshielded kernels use only slots
  // 8..14 (+ the obs staging: 15 slots); the sort keys are a parallel-form temporary
  constexpr bool kRoomy = false;  // (round 1 parked cos(heading) / g.vx in LDS as well: no longer measurable, 0.332 ms either way)
  constexpr int C_CPSI = kColdB + G - 1, C_GVX = C_CPSI + 1;
  constexpr int kColdN = !SHIELDED ? 15 : (kSerialOnly ? 23 : kColdB + G - 1 + (kRoomy ? 2 : 0));
  static_assert(kColdN * MM_STEP_BLOCK * 8 >= (MM_STEP_BLOCK / 64) * 64 * 30 * 4, "the obs staging must fit in the cold slots");
  __shared__ double s_cold[kColdN][MM_STEP_BLOCK];
  const int tid = threadIdx.x;
  auto cold_i = [&](int slot, int col) -> int & { return ((int *)&s_cold[slot][col])[0]; };  // an int kept in a slot's low word
  if (LC) {
    s_cold[C_H1X][tid] = v.h1x; s_cold[C_H1VX][tid] = v.h1vx; s_cold[C_H2X][tid] = v.h2x; s_cold[C_H2VX][tid] = v.h2vx;
    s_cold[C_SSTEER][tid] = v.safe_steer; s_cold[C_SACC][tid] = v.safe_acc;
  }
  s_cold[C_TSPEED][tid] = v.tspeed;
  // (form / slot selection is `if constexpr` throughout: an instantiation contains no code for slots it does not own)
  if constexpr (kRoomy) { s_cold[C_CPSI][tid] = cpsi; s_cold[C_GVX][tid] = v.gvx; }
  auto CPSI = [&]() -> double { if constexpr (kRoomy) return s_cold[C_CPSI][tid]; else return cpsi; };
  auto GVX = [&]() -> double { if constexpr (kRoomy) return s_cold[C_GVX][tid]; else return v.gvx; };
  bool env_active = n_ctrl > 0;
  if constexpr (SPLIT) {
    if (kb > 0) env_active = e < st.E && sb.envf[e] != 0;
  }
  STAMP(0);  // load + setup
  // split form: launch kb runs the commit half ("tail") of sub-step kb - 1 and the act half ("head") of sub-step kb
  const int k_lo = SPLIT ? (kb > 0 ? kb - 1 : 0) : 0, k_hi = SPLIT ? (kb < c.nsub ? kb : c.nsub - 1) : c.nsub - 1;
  for (int k = k_lo; k <= k_hi; k++) {
    const bool head = !SPLIT || k == kb, tail = !SPLIT || k != kb;
    const bool live = env_active && v.present;
    QpTrace qt = {0, __builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan(""),
                  __builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
    // Road.act / Road.step order: sorted by x descending, stable (road.py:277,286) -> rank
    int rank = 0;
    if ((SHIELDED || MIXED) && head) {
      for_partners<G>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        double px = px_d<m, G>(v.x, a);
        bool pp = px_i<m, G>((int)live, a) != 0;
        rank += (pp && (px > v.x || (px == v.x && pidx<m, G>(a) < a))) ? 1 : 0;
      });
    }
    if (head) v.tspeed = s_cold[C_TSPEED][tid];
    if (live && !hdv && head) {
      if (time % c.nsub == 0) hl_act<KIND>(v, action);  // action_type.act abstract.py:516-519
    }
    const int tl_pre = v.tlane;  // what an HDV acting before this vehicle still sees
    if (live && !hdv && head) controlled_act<!MIXED>(v, -1, sv, st_t);  // road.act road.py:269-278 (general kernels: steering below)
    if (head) s_cold[C_TSPEED][tid] = v.tspeed;
    if constexpr (MIXED) { STAMP(1); }  // (general kernels: "act" = the CAVs' part up to here; slots 3 / 4 / 5 split the HDVs' part)
    if constexpr (MIXED) {
      // ---------------- IDMVehicle.act for the HDVs (behavior.py:74-100) -------------------------
      // Positions do not move during Road.act, so each HDV scans its neighbours in parallel.  The one
      // thing acted-upon state an HDV reads is other vehicles' target_lane_index (ongoing-LC abort
      // test, :193-206): it sees the post-act value of vehicles ahead of it in the sweep, the pre-act
      // value of the others; HDV decisions ahead feed HDVs behind -> small fixed point on tl_post.
      if (head && __any(hdv && live)) {
        const bool acting = hdv && live && !v.crashed;
        // the sine-lane offset of my position, for the HDVs that scan lane kb0 (an ego's own lane only: the side lane of a
        // lane change is bc0 / bc1); one sine per vehicle instead of one per (ego, partner) pair
        double koff = 0.0, koff_obst = 0.0;
        if (__any(acting && v.lane == MM_LANE_KB0)) {
          koff = kSineAmp * mmm_sin(kSinePuls * (v.x - lane_sx(MM_LANE_KB0)) + kSinePhase);
          koff_obst = kSineAmp * mmm_sin(kSinePuls * (kObstX - lane_sx(MM_LANE_KB0)) + kSinePhase);
        }
        Body self = {true, false, v.x, v.y, v.h, v.v, v.tspeed, v.lane, spsi, cpsi, koff};
        const double s_me = v.x - lane_sx(v.lane);
        NbSel n0 = {0, 0, -1, -1}, n1 = {0, 0, -1, -1};  // own lane; the lane MOBIL considers
        const LaneFrame lf0 = lane_frame(v.lane);
        if (acting && lane_after_end(v.tlane, v.x)) v.tlane = next_lane(v.tlane, v.x, v.y);  // follow_road
        const int my_tl = v.tlane;
        const bool lc_branch = v.lane != my_tl;  // a lane change is under way (:191)
        const bool same_road = lane_road(v.lane) == lane_road(my_tl);
        const bool timer_due = !lc_branch && (1.0 < v.gvx);  // utils.do_every(LANE_CHANGE_DELAY, timer)
        const int side = v.lane == MM_LANE_BC0 ? MM_LANE_BC1 : (v.lane == MM_LANE_BC1 ? MM_LANE_BC0 : -1);
        const bool try_mobil = acting && timer_due && side >= 0 && lane_reachable(side < 0 ? 0 : side, v.x, v.y);
        const int sl = side < 0 ? 0 : side;
        const double s_side = v.x - lane_sx(sl);
        const LaneFrame lf1 = lane_frame(sl);
        unsigned abort_mask = 0;  // partners that would make me abort IF they target my target lane
        for_partners<G>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const int p = pidx<m, G>(a);
          const bool bp = px_i<m, G>((int)live, a) != 0;
          const double bx = px_d<m, G>(v.x, a), by = px_d<m, G>(v.y, a), bk = px_d<m, G>(koff, a);
          if (acting) {
            neighbour_update(lf0, s_me, bp, bx, by, bk, p, n0);
            if (try_mobil) neighbour_update(lf1, s_side, bp, bx, by, bk, p, n1);
          }
          if (__any(acting && lc_branch && same_road)) {  // the ongoing-lane-change abort test (:193-206) needs the partner as a Body
            Body b = {bp, false, bx, by, 0.0, px_d<m, G>(v.v, a), 0.0, px_i<m, G>(v.lane, a), px_d<m, G>(spsi, a), px_d<m, G>(cpsi, a), bk};
            if (acting && lc_branch && same_road && b.present && b.lane != my_tl) {
              const double sx = lane_sx(v.lane);
              const double d = (b.x - sx) - (v.x - sx);
              if (0 < d && d < desired_gap(self, b)) abort_mask |= 1u << p;
            }
          }
        });
        if constexpr (MIXED) { STAMP(3); }  // HDV neighbour scan
        if (acting) {  // road.objects come after the vehicles (road.py:369)
          neighbour_update(lf0, s_me, true, kObstX, kObstY, koff_obst, 99, n0);
          if (try_mobil) neighbour_update(lf1, s_side, true, kObstX, kObstY, koff_obst, 99, n1);
        }
        // gather what IDM reads of the chosen bodies from their owners (all lanes: the shuffles sit in uniform control flow)
        auto fetch = [&](int idx, bool full) {
          Body r = {};
          const int src = gb + ((idx >= 0 && idx < G) ? idx : 0);
          r.x = shfl_d(v.x, src); r.v = shfl_d(v.v, src); r.sh = shfl_d(spsi, src); r.ch = shfl_d(cpsi, src);
          if (full) { r.tspeed = shfl_d(v.tspeed, src); r.lane = shfl_i(v.lane, src); }
          r.present = idx >= 0;
          if (idx == 99) { r.is_object = true; r.x = kObstX; r.v = 0.0; r.sh = 0.0; r.ch = 1.0; r.tspeed = 0.0; r.lane = 0; }  // heading 0: sin 0, cos 1
          return r;
        };
        const Body front0 = fetch(acting ? n0.i_front : -1, false);
        Body front1 = {}, rear1 = {};
        if (__any(try_mobil)) { front1 = fetch(try_mobil ? n1.i_front : -1, false); rear1 = fetch(try_mobil ? n1.i_rear : -1, true); }
        const double acc_f0 = acting ? idm_acceleration(self, front0) : 0.0;  // (MOBIL's current-lane term and the HDV's own action)
        bool mobil_go = false;
        if (try_mobil) {  // mobil() behavior.py:225-266 (no route, POLITENESS = 0)
          const double nf_pred = idm_acceleration(rear1, self);
          if (!(nf_pred < -9.0)) {
            const double jerk = idm_acceleration(self, front1) - acc_f0 + 0.0;
            mobil_go = !(jerk < 0.1);
          }
        }
        if constexpr (MIXED) { STAMP(4); }  // gather + IDM / MOBIL
        if (acting && timer_due) v.gvx = 0;  // self.timer = 0 (:212)
        // fixed point on the post-act target lanes
        int tl_post = acting ? (lc_branch ? my_tl : (mobil_go ? side : my_tl)) : v.tlane;
        for (int round = 0; round <= st.N; round++) {
          bool hit = false;
          for_partners<G>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const int p = pidx<m, G>(a);
            const int pk = px_i<m, G>(rank | tl_pre << 4 | tl_post << 8, a);
            const int p_tl = ((pk & 15) < rank) ? ((pk >> 8) & 7) : ((pk >> 4) & 7);  // acted before me?
            hit = hit || (((abort_mask >> p) & 1u) && p_tl == my_tl);
          });
          const int nxt = (acting && lc_branch) ? ((same_road && hit) ? v.lane : my_tl) : tl_post;
          const bool changed = nxt != tl_post;
          tl_post = nxt;
          if (!__any(changed)) break;
        }
        if constexpr (MIXED) { STAMP(5); }  // target-lane fixed point
        if (acting) {
          v.tlane = tl_post;
          v.act_acc = clipd(acc_f0, -6.0, 6.0);
        }
      }
      // one steering_control for every vehicle that steers this sub-step: the CAVs, and the HDVs that acted (a crashed HDV
      // keeps its last action)
      if (head && live && (!hdv || !v.crashed)) steer_lane(v, sv && !hdv, st_t);
      if (head && hdv && live) v.gvx += dt;  // IDMVehicle.step: self.timer += dt (behavior.py:102-109)
    }
    if (live && head) clip_actions(v, LC && !hdv, st_t);
    STAMP(1);  // act
    // predicted post-state for the nominal steering (the only one when nothing vetoes)
    Cand cA;
    memset(&cA, 0, sizeof cA);
    const bool shield_on = SHIELDED && live && !hdv && v.hist_len >= 2;  // gate safe_controller.py:232-239
    if (live && head) cA = predict<KIND, SHIELDED, MASS, MIXED>(v, v.act_steer, st_t, spsi, cpsi, dt, sv && !hdv);
    auto park = [&](auto base_c, const Cand &cc, double steer) {  // a candidate's LDS image (shielded kernels only)
      constexpr int base = decltype(base_c)::value;
      if constexpr (SHIELDED) {
        s_cold[base + 0][tid] = cc.x; s_cold[base + 1][tid] = cc.y; s_cold[base + 2][tid] = cc.h;
        cold_i(base + 3, tid) = cc.pk; s_cold[base + 4][tid] = cc.cpsi; s_cold[base + 5][tid] = steer;
        s_cold[base + 6][tid] = cc.gvx; s_cold[base + 7][tid] = cc.spsi;
      }
    };
    if (head) park(std::integral_constant<int, C_A>{}, cA, v.act_steer);
    STAMP(2);  // predict A
    // LC veto re-steers to the CURRENT lane (decentral_layer.py:501-506,739-744); identical to the
    // nominal command unless a lane change / lane hand-over is under way or the car crashed.
    // Candidate B is only predicted when a veto actually fires (lazily, below).
    // ("steer_vel": the veto's velocity command is not clipped to +-pi/3 like the nominal one, so it can
    // differ even without a lane change)
    const bool needB = SHIELDED && shield_on && (v.tlane != v.lane || v.crashed || sv);
    bool haveB = false;
    auto make_B = [&]() {
      if constexpr (SHIELDED) {
        if (needB && !haveB) {
          double tB;
          double steerB = steering_control(v.x, v.y, v.h, v.v, v.lane, tB);
          if (sv) steerB = steer_vel_command(steerB, v.sang);
          park(std::integral_constant<int, C_B>{}, predict<KIND, true, MASS, MIXED>(v, steerB, tB, spsi, CPSI(), dt, sv), steerB);
          haveB = true;
        }
      }
    };
    // the candidate a vehicle commits / shows to later vehicles: A, or B (from LDS) after a veto
    auto chosen = [&](bool useB) {
      if constexpr (!SHIELDED) {
        return cA;
      } else {
        const int base = useB ? C_B : C_A;  // both candidates sit in LDS; A's registers are free meanwhile
        Cand cc;
        cc.x = s_cold[base + 0][tid]; cc.y = s_cold[base + 1][tid]; cc.h = s_cold[base + 2][tid];
        cc.gvx = s_cold[base + 6][tid]; cc.cpsi = s_cold[base + 4][tid]; cc.spsi = s_cold[base + 7][tid];
        const int pk = cold_i(base + 3, tid);
        cc.pk = pk; cc.lane = pk & 7;
        return cc;
      }
    };
    if constexpr (SPLIT) {
      if (!head && live && valid) {
        // commit half of sub-step k: what its act half (previous launch) left in the hand-off planes -- the clipped nominal
        // acceleration and the candidate images -- goes back where the fused form keeps it
        v.act_acc = sw_f(sb, SW_ACCN, a, e);
        if (MIXED && hdv) {  // an HDV's own action (store_veh keeps it in the safe_* planes) and the twin edit of its record
          v.act_steer = sw_f(sb, SW_ASTEER, a, e);
          if (sw_i(sb, SW_RES, a, e) & 16) s_cold[C_H1X][tid] = sw_f(sb, SW_ACC, a, e);  // (decentral_layer.py:164-184: state_hist[-2] edited in place)
        }
        s_cold[C_A + 0][tid] = sw_f(sb, SW_AX, a, e); s_cold[C_A + 1][tid] = sw_f(sb, SW_AY, a, e); s_cold[C_A + 2][tid] = sw_f(sb, SW_AH, a, e);
        cold_i(C_A + 3, tid) = sw_i(sb, SW_APK, a, e); s_cold[C_A + 4][tid] = sw_f(sb, SW_ACPSI, a, e); s_cold[C_A + 5][tid] = sw_f(sb, SW_ASTEER, a, e);
        s_cold[C_A + 6][tid] = sw_f(sb, SW_AGVX, a, e); s_cold[C_A + 7][tid] = sw_f(sb, SW_ASPSI, a, e);
        if (needB) {
          s_cold[C_B + 0][tid] = sw_f(sb, SW_BX, a, e); s_cold[C_B + 1][tid] = sw_f(sb, SW_BY, a, e); s_cold[C_B + 2][tid] = sw_f(sb, SW_BH, a, e);
          cold_i(C_B + 3, tid) = sw_i(sb, SW_BPK, a, e); s_cold[C_B + 4][tid] = sw_f(sb, SW_BCPSI, a, e); s_cold[C_B + 5][tid] = sw_f(sb, SW_BSTEER, a, e);
          s_cold[C_B + 6][tid] = sw_f(sb, SW_BGVX, a, e); s_cold[C_B + 7][tid] = sw_f(sb, SW_BSPSI, a, e);
          haveB = true;
        }
      }
    }
    double new_acc = v.act_acc;
    bool use_B = false, veto = false;
    int new_flags = v.flags;
    if constexpr (SHIELDED && !SPLIT && !kSerialOnly && kVetoPrior) {
      // First guess of the veto passes below: what this vehicle's shield said one sub-step ago (a vetoed lane change usually
      // stays vetoed: starting from "no veto" cost such a wave a second classification + solve pass in every sub-step).  The
      // passes converge to the sequential answer from ANY first guess (rank r is final after pass r); a wrong one costs
      // the pass the right one saves.
      if ((c.debug_flags & 1) == 0 && shield_on && needB && !(v.flags & MM_FLAG_IS_LC_SAFE)) { make_B(); use_B = true; }
    }
    if constexpr (SPLIT) {
      if (head) {
        // act half of sub-step k: hand the shield's inputs to the sweep kernel (one lane per env there)
        make_B();  // (eagerly, like the literal sweep: the sweep kernel decides the vetoes)
        // the classification pass below assumes every vehicle commits the candidate its LAST veto decision points to (bit 4 of
        // SW_META); the sweep kernel classifies an env itself only from the first vehicle on that decides otherwise
        if constexpr (!kSerialOnly && kVetoPrior) use_B = (c.debug_flags & 1) == 0 && shield_on && needB && !(v.flags & MM_FLAG_IS_LC_SAFE);
        const int n_live = __popc(group_ballot<G>(live, gb));
        if (valid) {
          sw_i(sb, SW_META, a, e) = (live ? 1 : 0) | (shield_on ? 2 : 0) | (needB ? 4 : 0) | (hdv ? 8 : 0) | (use_B ? 16 : 0) | (v.flags & 255) << 8 | (v.hl & 255) << 16;
          if (live) {
            // (the pre-step view the literal sweep starts from -- serial form below: wx .. wgu -- is built by the sweep kernel from
            // the state planes this launch stores; only what is not state goes through the hand-off planes)
            sw_i(sb, SW_WPK, a, e) = pk_self;
            sw_f(sb, SW_CPSI, a, e) = cpsi; sw_f(sb, SW_ACCN, a, e) = v.act_acc;
            sw_f(sb, SW_AX, a, e) = cA.x; sw_f(sb, SW_AY, a, e) = cA.y; sw_f(sb, SW_AH, a, e) = cA.h; sw_i(sb, SW_APK, a, e) = cA.pk;
            sw_f(sb, SW_ACPSI, a, e) = cA.cpsi; sw_f(sb, SW_AGVX, a, e) = cA.gvx; sw_f(sb, SW_ASTEER, a, e) = v.act_steer; sw_f(sb, SW_ASPSI, a, e) = cA.spsi;
            if (needB) {
              sw_f(sb, SW_BX, a, e) = s_cold[C_B + 0][tid]; sw_f(sb, SW_BY, a, e) = s_cold[C_B + 1][tid]; sw_f(sb, SW_BH, a, e) = s_cold[C_B + 2][tid];
              sw_i(sb, SW_BPK, a, e) = cold_i(C_B + 3, tid); sw_f(sb, SW_BCPSI, a, e) = s_cold[C_B + 4][tid]; sw_f(sb, SW_BSTEER, a, e) = s_cold[C_B + 5][tid];
              sw_f(sb, SW_BGVX, a, e) = s_cold[C_B + 6][tid]; sw_f(sb, SW_BSPSI, a, e) = s_cold[C_B + 7][tid];
            }
            sb.order[(long long)rank * sb.Ep + e] = (uint8_t)a;  // sweep order: ranks 0 .. n_live - 1 are taken by the live vehicles
          }
          if (a >= n_live) sb.order[(long long)a * sb.Ep + e] = 0xFF;
          if (a == 0) sb.envf[e] = env_active ? 1 : 0;
        }
      } else if (shield_on && valid) {
        const int res = sw_i(sb, SW_RES, a, e);  // what the sweep kernel decided for this vehicle
        new_acc = sw_f(sb, SW_ACC, a, e); veto = (res & 2) != 0; use_B = (res & 4) != 0; new_flags = (res >> 8) & 255;
      }
    }

    // (split form: only the classification pass of the parallel form runs, in the act half -- its slot selection is handed
    // to the sweep kernel, which does everything from the rows on; the rest of this block is dead code there)
    if constexpr (SHIELDED) {
     if (__any(shield_on) && (!SPLIT || head)) {
      // Which form runs is a compile-time property of the instantiation (kSerialOnly, above); debug_flags
      // bit0 forces the literal sweep in the kernels that carry both.
      bool serial = !SPLIT && (kSerialOnly || (c.debug_flags & 1) != 0);  // (split form: bit0 makes the sweep kernel classify itself)
      ShieldOut so;
      memset(&so, 0, sizeof so);
      if constexpr (!kSerialOnly) {
       if (!serial) {
        // ------------- parallel form of the front-to-back sweep --------------------------------
        // When vehicle i runs its shield, vehicle j is in its committed post-state if it steps
        // earlier (rank_j < rank_i), else in its pre-step state.  Post-states are the predicted
        // candidates (A, or B where j's own shield vetoes), so every lane can classify its
        // neighbours at once; what remains serial is (1) MASS: a follower needs its leader's
        // DECIDED acceleration -> fixed point over the leader chain, (2) a veto changes what
        // later vehicles see -> outer fixed point.  Both iterate to the unique sequential answer
        // (induction on rank: the rank-r vehicle is final after r+1 rounds / passes).
        // Register discipline: the partner loop only classifies and selects (key, index, flags) per slot;
        // the chosen neighbours' records are gathered afterwards from their owners' LDS columns (the history
        // records already live there, the g*u products are parked at the top of the pass), so the loop does
        // not drag 10 neighbour fields and their select chains through 7 (15) partners.
        static_assert(!MIXED || kSerialOnly, "the parallel form is CAV-only (the twin branch needs the literal sweep)");
        bool irregular = false;
        // can a vehicle have more than 5 others around it at all?  (compile-time only: making the 8-lane kernels
        // test st.N > 6 as well cost them 3 % through 20 B/lane more scratch)
        constexpr bool count5 = G > 4;
        if constexpr (kMailbox) { s_cold[C_PRE + 0][tid] = v.x; s_cold[C_PRE + 1][tid] = v.y; s_cold[C_PRE + 2][tid] = v.h; cold_i(C_PRE + 3, tid) = pk_self; }
        // (interior-point mode) the QP this lane solved in the previous veto pass of this sub-step: rows that did not change
        // need no second solve.  a, h1, h2 depend on the lane's own pre-step state only, so (rows, h0, h3) identify the QP.
        double qc_h0 = 0.0, qc_h3 = 0.0, qc_d = 0.0;
        int qc_meta = 0;  // rows (3 | 4; 0: nothing cached) | 8: status "optimal"
        (void)qc_h0; (void)qc_h3; (void)qc_d; (void)qc_meta;
        for (int pass = 0; pass <= st.N; pass++) {
#ifdef MM_STAMPS
#ifdef MM_COUNT_ROUNDS  // (one-off: count the MASS rounds instead of the passes)
          if ((threadIdx.x & 63) == 13) _t_acc += (pass == 0 ? 1ull << 32 : 0ull);
#else
          if ((threadIdx.x & 63) == 13) _t_acc += 1 + (pass == 0 ? 1ull << 32 : 0ull);  // (stamps builds: veto passes | wave-sub-steps << 32)
#endif
#endif
          const double mine_gvx = s_cold[(use_B ? C_B : C_A) + 6][tid];  // g.vx of the candidate I commit
          const double h1vx_mine = s_cold[C_H1VX][tid];
          s_cold[C_GU0][tid] = slot_gu<MASS>(s_cold[C_H2VX][tid], MASS ? s_cold[C_SACC][tid] : kCbfAccLo, GVX(), dt);
          s_cold[C_GU1][tid] = slot_gu<MASS>(h1vx_mine, kCbfAccLo, mine_gvx, dt);  // (MASS: the rounds exchange the live value)
          // my rank word: rank (99: not live) | bit 8: the candidate I commit is B.  With the pre-step pose posted above and
          // both candidate images in LDS, a partner picks what it sees of me by address -- my committed post-state if I step
          // before it, else my pre-step pose -- instead of receiving 3 doubles + 2 ints through DPP and select chains
          if constexpr (kMailbox) {
            const int my_word = (live ? rank : 99) | (use_B ? 256 : 0);
            if constexpr (kPackMsg) ((int *)&s_cold[C_PRE + 3][tid])[1] = my_word;
            else cold_i(C_MSGW, tid) = my_word;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
          Cand mine;  // (DPP exchange only)
          if constexpr (!kMailbox) mine = chosen(use_B);
          double k_ol = INFINITY, k_oa = INFINITY, k_oar = INFINITY;
          int j_ol = -1, j_oa = -1, j_oar = -1;
          for_partners<G>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const int p = pidx<m, G>(a);
            int p_rank, opk;
            double ox, oy, oh;
            bool i_first;
            if constexpr (kMailbox) {
              const int pl = plane<m, G>(tid, a);
              int pw;
              if constexpr (kPackMsg) pw = ((const int *)&s_cold[C_PRE + 3][pl])[1];
              else pw = cold_i(C_MSGW, pl);
              p_rank = pw & 255;
              i_first = live && rank < p_rank;                      // I step before this partner
              // its pose as I see it: the committed post-state if it steps before me, else the pre-step pose
              const int pb = (!i_first && p_rank < 99) ? ((pw & 256) ? (int)C_B : (int)C_A) : (int)C_PRE;
              ox = s_cold[pb + 0][pl]; oy = s_cold[pb + 1][pl]; oh = s_cold[pb + 2][pl];
              opk = cold_i(pb + 3, pl) | (p_rank < 99 ? 256 : 0);
            } else {
              p_rank = dppx_i<m>(live ? rank : 99);
              i_first = live && rank < p_rank;
              // message: my pose as that partner sees it (post-state if I step before it, else pre-state)
              const double sx_ = i_first ? mine.x : v.x, sy_ = i_first ? mine.y : v.y, sh_ = i_first ? mine.h : v.h;
              const int spk = i_first ? mine.pk : pk_self;
              ox = dppx_d<m>(sx_); oy = dppx_d<m>(sy_); oh = dppx_d<m>(sh_);
              opk = dppx_i<m>(spk | (int)live << 8);
            }
            const bool o_first = !i_first && p_rank < 99;  // partner steps before me (ranks are distinct)
            const Rel r = relate(v.x, v.y, pk_self, ((opk >> 8) & 1) != 0, ox, oy, oh, opk);
            if constexpr (count5) s_cold[kColdB + m - 1][tid] = r.key;
            // running "first in sorted order" per class: smaller key, ties by creation index (selects, no branches)
            // (the slot's flags ride in the index word -- bit 4: that partner steps before me, bit 5: its corner
            // flag -- instead of living as lane masks through the loop: 84 fewer SGPR spills)
            const bool b_ol = (r.cls == 1) & ((r.key < k_ol) | ((r.key == k_ol) & (p < (j_ol & 15))));
            const bool b_oa = (r.cls == 2) & ((r.key < k_oa) | ((r.key == k_oa) & (p < (j_oa & 15))));
            const bool b_oar = (r.cls == 3) & ((r.key < k_oar) | ((r.key == k_oar) & (p < (j_oar & 15))));
            const int pf = p | (o_first ? 16 : 0) | (r.cflag ? 32 : 0);
            k_ol = b_ol ? r.key : k_ol; j_ol = b_ol ? pf : j_ol;
            k_oa = b_oa ? r.key : k_oa; j_oa = b_oa ? pf : j_oa;
            k_oar = b_oar ? r.key : k_oar; j_oar = b_oar ? pf : j_oar;
          });
          const bool ol_first = j_ol >= 0 && (j_ol & 16), oa_first = j_oa >= 0 && (j_oa & 16), oar_stepped = j_oar >= 0 && (j_oar & 16);
          const bool cadj = j_oa >= 0 && (j_oa & 32);
          j_ol = j_ol < 0 ? -1 : (j_ol & 15); j_oa = j_oa < 0 ? -1 : (j_oa & 15); j_oar = j_oar < 0 ? -1 : (j_oar & 15);
          STAMP(3);  // S1 partner classification
          Neigh nb;
          memset(&nb, 0, sizeof nb);
          // count = 5 of close_vehicles_to: a slot exists only if its vehicle is among the 5 nearest
          // (with at most 6 vehicles per env every other vehicle is: the position count is skipped)
          nb.has_ol = j_ol >= 0; nb.has_oa = j_oa >= 0; nb.has_oar = j_oar >= 0;
          if constexpr (count5) {
            int pos_ol = 0, pos_oa = 0, pos_oar = 0;
#pragma unroll
            for (int m = 1; m < G; m++) {
              const int p = pidx_rt<G>(a, m);
              const double km = s_cold[kColdB + m - 1][tid];
              pos_ol += (km < k_ol || (km == k_ol && p < j_ol)) ? 1 : 0;
              pos_oa += (km < k_oa || (km == k_oa && p < j_oa)) ? 1 : 0;
              pos_oar += (km < k_oar || (km == k_oar && p < j_oar)) ? 1 : 0;
            }
            nb.has_ol = nb.has_ol && pos_ol < 5;
            nb.has_oa = nb.has_oa && pos_oa < 5;
            nb.has_oar = nb.has_oar && pos_oar < 5;
          }
          nb.constrain_adj = MASS && nb.has_oa && cadj;
          if constexpr (SPLIT) {
            if (valid && shield_on)
              sw_i(sb, SW_CLS, a, e) = (j_ol & 15) | (j_oa & 15) << 4 | (j_oar & 15) << 8 | (nb.has_ol ? 1 << 12 : 0) | (nb.has_oa ? 1 << 13 : 0) |
                                       (nb.has_oar ? 1 << 14 : 0) | (nb.constrain_adj ? 1 << 15 : 0);
            break;
          }
          // gather the chosen neighbours' records from their owners' columns (same wave: program order + a
          // wave-level fence make the parked values visible)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int t_ol = tid - a + (j_ol < 0 ? a : j_ol), t_oa = tid - a + (j_oa < 0 ? a : j_oa);
          nb.ol_x = s_cold[ol_first ? C_H1X : C_H2X][t_ol]; nb.ol_gu = s_cold[ol_first ? C_GU1 : C_GU0][t_ol];
          nb.oa_x = s_cold[oa_first ? C_H1X : C_H2X][t_oa]; nb.oa_gu = s_cold[oa_first ? C_GU1 : C_GU0][t_oa];
          const int src_ol = gb + (j_ol < 0 ? 0 : j_ol), src_oa = gb + (j_oa < 0 ? 0 : j_oa), src_oar = gb + (j_oar < 0 ? 0 : j_oar);
          const double cpsi_now = CPSI();
          nb.oar_x = shfl_d(v.x, src_oar); nb.oar_vx = shfl_d(v.v * cpsi_now, src_oar);  // rear adjacent: its current to_dict()
          // MASS: a neighbour that has stepped shows the acceleration it DECIDED this sub-step -> rounds below
          bool ol_dyn = MASS && nb.has_ol && ol_first, oa_dyn = MASS && nb.has_oa && oa_first;
          {
            unsigned rep = obstacle_override<MASS>(nb, v.x, v.y);  // the obstacle's action is static
            if (rep & 1u) ol_dyn = false;
            if (rep & 2u) oa_dyn = false;
          }
          // a stepped vehicle seen as REAR-adjacent would need its post-step speed: only possible when
          // a vehicle moves backwards in x; handled by the serial form below
          irregular = irregular || (shield_on && nb.has_oar && oar_stepped);
          // ---- MASS: fixed point over the decided accelerations (HSS: one evaluation) ----------
          double acc_cur = shield_on ? 0.0 : v.act_acc;  // vehicles without a shield keep their command
          if constexpr (kRoomy) v.gvx = GVX();
          const ShieldStatic ss = shield_static<MASS>(c, v, cpsi_now, pk_self, nb);
          if constexpr (IPM) {
            // ---- interior-point mode: ONE wave-wide loop, every lane at its own iteration of its own QP ----------------
            // A follower's QP reads the acceleration its leader / front-adjacent vehicle DECIDED this sub-step, so a QP can
            // start only when those decisions are final (`fin`).  Each trip of the loop: (1) every running lane evaluates
            // cvxopt's stopping test (mm_qp_top); a lane that stops derives its acceleration and becomes final; (2) lanes
            // whose inputs are final now -- including the ones that became final in (1) -- set up their rows and start
            // (mm_qp_start + the stopping test of the initial point), or take the answer of the previous veto pass when the
            // rows did not change; (3) every running lane does one interior-point iteration (mm_qp_bottom).  A QP with n
            // iterations occupies n trips and its follower starts in the trip it stopped in, so the wave leaves after its
            // longest dependency chain of iterations -- not after N sweep stages, and a QP that runs to the iteration cap
            // holds up its own followers only.  tools/ipm_sched_model.py replays this schedule on the CPU oracle's QP log.
            // Only what the loop itself reads stays live in it: the rows' ingredients and, per finished lane, the decided
            // acceleration its followers wait for; status / veto / flags are evaluated once after the loop (shield_post).
            bool fin = !shield_on, run = false, sing = false;
            bool late = false, late_opt = false;  // finished outside step (1): posted in the next trip
            const bool slow_lc = MASS && (v.hl == 2 || v.hl == 0) && v.v < kStoppingSpeed;  // decentral_layer.py:746-750: the decision needs the veto
            MMQpState q;
            q.x0 = 0.0; q.m4 = 0; q.iters = 0;
            bool more = true;  // wave-uniform
            for (int trip = 0; more; trip++) {
              if (trip > (MM_QP_MAXITERS + 2) * (G + 1)) { atomicOr(c.err, MM_LATCH_INTERNAL); break; }  // cannot happen: every trip starts or advances a QP
              MMQpRes rs;
              bool done_now = late, opt = late_opt;
              late = false;
              if (run) {
                const int stop = mm_qp_top(&q, &rs);
                if (stop || sing) { run = false; done_now = true; opt = stop == 1 && !sing; }
              }
              if (__any(done_now)) {
                if (done_now) {
                  double us0 = ss.u0 + q.x0;
                  if (MASS && slow_lc) us0 = shield_post<MASS>(c, v, ss, shield_rows<true>(c, ss, nb), q.x0, opt, true).us0;
                  acc_cur = div_c(us0 - ss.evx, dt, c.inv_dt);  // derived_acceleration :80-82 (== shield_post's)
                  fin = true;
                  qc_d = q.x0; qc_meta = (ss.cadj ? 4 : 3) | (opt ? 8 : 0);
                }
              }
              bool ready = shield_on && !fin && !run;
              double da = 0.0, db = 0.0;
              if (MASS) {
                const double gu_cur = slot_gu<true>(h1vx_mine, acc_cur, mine_gvx, dt);  // my post-step record under my decision
                da = shfl_d(gu_cur, src_ol); db = shfl_d(gu_cur, src_oa);
                const unsigned long long fm = __ballot(fin);
                ready = ready && (!ol_dyn || ((fm >> src_ol) & 1ull)) && (!oa_dyn || ((fm >> src_oa) & 1ull));
              }
              if (__any(ready)) {
                if (ready) {
                  if (MASS) { if (ol_dyn) nb.ol_gu = da; if (oa_dyn) nb.oa_gu = db; }
                  const ShieldRows rw = shield_rows<true>(c, ss, nb);
                  const int rows = ss.cadj ? 4 : 3;
                  if ((qc_meta & 7) == rows && __double_as_longlong(qc_h0) == __double_as_longlong(rw.h0) &&
                      __double_as_longlong(qc_h3) == __double_as_longlong(rw.h3)) {
                    q.x0 = qc_d; late = true; late_opt = (qc_meta & 8) != 0;  // same QP as in the previous pass: same answer
                  } else {
                    qc_h0 = rw.h0; qc_h3 = rw.h3;
                    sing = false;
                    // (a singular initial KKT system -- NaN input -- makes cvxopt raise: iterate NaN, "unknown")
                    int stop = mm_qp_start(&q, ss.g0, rw.h0, ss.h1, ss.h2, rw.h3, rows) != 0 ? 0 : 2;
                    if (!stop) stop = mm_qp_top(&q, &rs);
                    run = stop == 0;
                    if (stop) { late = true; late_opt = stop == 1; }
                  }
                }
              }
              const bool any_run = __any(run);
              if (any_run) {
                if (run) sing = mm_qp_bottom(&q, &rs) == 0;  // singular KKT matrix: the iterate stands, status "unknown"
              }
              more = any_run || __any(!fin);  // (nothing iterating but lanes not final: they waited for this trip's decisions)
            }
            if (shield_on) so = shield_post<MASS>(c, v, ss, shield_rows<true>(c, ss, nb), qc_d, (qc_meta & 8) != 0, true);
          } else {
          if constexpr (MASS && kRoundsLite) {
            // The rounds need the decided ACCELERATION only: rows, the QP and derived_acceleration.  Status, is_lc_allowed, veto and
            // flags (shield_post) are evaluated once, on the rows the fixed point ended with -- except for a lane in the slow-LC
            // bypass (decentral_layer.py:746-750), whose acceleration depends on its veto.
            const bool slow_lc = (v.hl == 2 || v.hl == 0) && v.v < kStoppingSpeed;
            ShieldRows rr;
            double d_cur = 0.0;
            for (int round = 0; round <= st.N; round++) {
#if defined(MM_STAMPS) && defined(MM_COUNT_ROUNDS)
              if ((threadIdx.x & 63) == 13) _t_acc += 1;
#endif
              const double gu_cur = slot_gu<true>(h1vx_mine, acc_cur, mine_gvx, dt);  // my post-step record under my current decision
              const double da = shfl_d(gu_cur, src_ol), db = shfl_d(gu_cur, src_oa);
              if (ol_dyn) nb.ol_gu = da;
              if (oa_dyn) nb.oa_gu = db;
              rr = shield_rows<true>(c, ss, nb);
              d_cur = qp_exact(ss, rr);
              double acc_next = v.act_acc;
              if (shield_on) {
                if (slow_lc) acc_next = shield_post<true>(c, v, ss, rr, d_cur, true, false).acc;
                else acc_next = div_c((ss.u0 + d_cur) - ss.evx, dt, c.inv_dt);  // derived_acceleration :80-82 (shield_post's expression)
              }
              const bool changed = __double_as_longlong(acc_next) != __double_as_longlong(acc_cur);
              acc_cur = acc_next;
              // (4.3 rounds per pass on the headline batch.  Leaving without the confirming round -- on "no decision some lane
              // reads changed", by ballot, or on "no lane received anything new" at the next exchange -- saves an evaluation and
              // measured 2 - 3 % SLOWER both ways: profiles/r04/variants.jsonl)
              if (!__any(changed)) break;
            }
            so = shield_post<true>(c, v, ss, rr, d_cur, true, false);
          } else {
          for (int round = 0; round <= st.N; round++) {
            if (MASS) {
              const double gu_cur = slot_gu<true>(h1vx_mine, acc_cur, mine_gvx, dt);  // my post-step record under my current decision
              const double da = shfl_d(gu_cur, src_ol), db = shfl_d(gu_cur, src_oa);
              if (ol_dyn) nb.ol_gu = da;
              if (oa_dyn) nb.oa_gu = db;
            }
            so = shield_dyn<MASS, true, false>(c, v, ss, nb, shield_on);
            const double acc_next = shield_on ? so.acc : v.act_acc;
            const bool changed = __double_as_longlong(acc_next) != __double_as_longlong(acc_cur);
            acc_cur = acc_next;
            if (!MASS || !__any(changed)) break;
          }
          }
          }
          if (TRACE && shield_on) trace_status(out.trace + (long long)k * MM_T_COUNT * A + i, A, so);
          STAMP(4);  // selection + fixed-point rounds
          const bool want_B = shield_on && so.veto && needB;
          if (want_B) make_B();
          STAMP(5);  // lazy candidate B
          const bool flip = want_B != use_B;
          use_B = want_B;
          if (!__any(flip)) break;
        }
        serial = __any(irregular);
        if (!serial && shield_on) {
          new_acc = so.acc; veto = so.veto; new_flags = so.flags; qt = so.qt;
          if (IPM && so.bounds) atomicOr(c.err, MM_LATCH_QP_BOUNDS);  // check_bounds on the QP this vehicle finally solved
        }
       }
      }
      if (!SPLIT && serial) {
        // ------------- literal front-to-back sweep (fallback / validation form) -----------------
        make_B();
        use_B = false; veto = false; new_acc = v.act_acc; new_flags = v.flags;
        // working copy of what the others see of me; committed stage by stage
        if constexpr (kRoomy) { v.gvx = GVX(); cpsi = CPSI(); }
        double wx = v.x, wy = v.y, wh = v.h, wg = hdv ? 1.0 : v.gvx, wacc = hdv ? kCbfAccLo : s_cold[C_SACC][tid], wvx = v.v * cpsi;
        double whx = s_cold[C_H2X][tid], whvx = s_cold[C_H2VX][tid];
        int wpk = pk_self;
        if (MIXED && hdv && live) {
          // HDVs take no part in the shield: publish their stepped view when the sweep passes them
          // (done below by rank), here only note that their record [-2] after stepping is the old [-1]
        }
        // the g*u product an ego's CBF rows take from my record (slot_gu), recomputed whenever the record changes
        double wgu = slot_gu<MASS>(whvx, MASS ? wacc : kCbfAccLo, wg, dt);
        bool w_stepped = false;  // my published view is already the post-step one
        double twin_shift = 0;   // in-place edits egos made to my history record this sub-step
        // One stage per SHIELDED vehicle of the group, in sweep order -- not one per rank: a rank held by an HDV (or by no
        // vehicle) has no shield to run, and with 4 CAVs + 4 HDVs that was half of the stages.  srank = my position among
        // the group's shielded vehicles; the wave leaves when no group has a j-th one.
        int srank = 0;
        for_partners<G>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          srank += (px_i<m, G>(shield_on ? rank : 99, a) < rank) ? 1 : 0;
        });
        for (int j = 0; j < st.N; j++) {
          const unsigned sel = group_ballot<G>(shield_on && srank == j, gb);
          if (!__any(sel != 0)) break;
          const bool has = sel != 0;
          const int ai = has ? (__ffs((int)sel) - 1) : 0;
          const int src = gb + ai;
          if constexpr (MIXED) {
            // an HDV that comes before this ego in the sweep has stepped by now (no shield): the ego sees its post-step
            // pose, and its state_hist[-2] is then the record it held as [-1] before
            const int r_ego = shfl_i(rank, src);
            if (live && hdv && has && rank < r_ego && !w_stepped) {
              const Cand ca = chosen(false);
              wx = ca.x; wy = ca.y; wh = ca.h; wpk = ca.pk;
              whx = s_cold[C_H1X][tid]; whvx = s_cold[C_H1VX][tid]; w_stepped = true; twin_shift = 0;
              wgu = slot_gu<MASS>(whvx, MASS ? wacc : kCbfAccLo, wg, dt);
            }
          }
          const double ex = shfl_d(v.x, src), ey = shfl_d(v.y, src);
          const int epk = shfl_i(pk_self, src);
          const double e_vx = shfl_d(v.v * cpsi, src);  // vehicle.velocity[0] of the ego
          const Rel rl = relate(ex, ey, epk, live && has && a != ai, wx, wy, wh, wpk, hdv);
          int pos = 0;
          for_partners<G>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            double kp = px_d<m, G>(rl.key, a);
            pos += (kp < rl.key || (kp == rl.key && pidx<m, G>(a) < a)) ? 1 : 0;
          });
          const bool in5 = rl.key < INFINITY && pos < 5;
          const unsigned none = 0x3FFu;
          unsigned f_ol = (in5 && rl.cls == 1) ? ((unsigned)pos << 5 | (unsigned)a << 1) : none;
          unsigned f_oa = (in5 && rl.cls == 2) ? ((unsigned)pos << 5 | (unsigned)a << 1 | (rl.cflag ? 1u : 0u)) : none;
          unsigned f_oar = (in5 && rl.cls == 3) ? ((unsigned)pos << 5 | (unsigned)a << 1) : none;
          // digital-twin slot: every such HDV gets its record shifted, the LAST one in sorted order
          // ends up in s_oa (the branch has no `is None` guard, :162-184) -> min over (15 - pos)
          const bool is_twin = MIXED && in5 && rl.cls == 4;
          if (is_twin) { whx = whx + 0.5 * e_vx; twin_shift = 1; }
          unsigned f_tw = is_twin ? ((unsigned)(15 - pos) << 5 | (unsigned)a << 1) : none;
          unsigned w = f_ol | f_oa << 10 | f_oar << 20;
          // butterfly min over the group by DPP (one VALU-rate op per exchange; after xor 1, 2 a quad is
          // uniform, so row_half_mirror serves as xor 4, then row_ror:8 as xor 8)
          auto red = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const unsigned o = (unsigned)dppx_i<m>((int)w);
            const unsigned m0 = min(w & 0x3FFu, o & 0x3FFu), m1 = min((w >> 10) & 0x3FFu, (o >> 10) & 0x3FFu),
                           m2 = min((w >> 20) & 0x3FFu, (o >> 20) & 0x3FFu);
            w = m0 | m1 << 10 | m2 << 20;
            if (MIXED) f_tw = min(f_tw, (unsigned)dppx_i<m>((int)f_tw));
          };
          if constexpr (kPow2<G>) {
            if constexpr (G >= 2) red(std::integral_constant<int, 1>{});
            if constexpr (G >= 4) red(std::integral_constant<int, 2>{});
            if constexpr (G >= 8) red(std::integral_constant<int, 7>{});
            if constexpr (G >= 16) red(std::integral_constant<int, 8>{});
          } else {
            // rotation layout: doubling steps 1, 2, 4, 8 -- after them a lane holds the minimum over itself and the next 15
            // lanes of its ring of G <= 16, i.e. over the whole group (windows overlap: harmless for a minimum)
            auto ring = [&](auto mc) {
              constexpr int off = decltype(mc)::value;
              if constexpr (off < G) {
                const unsigned o = (unsigned)px_i<off, G>((int)w, a);
                const unsigned m0 = min(w & 0x3FFu, o & 0x3FFu), m1 = min((w >> 10) & 0x3FFu, (o >> 10) & 0x3FFu),
                               m2 = min((w >> 20) & 0x3FFu, (o >> 20) & 0x3FFu);
                w = m0 | m1 << 10 | m2 << 20;
                if (MIXED) f_tw = min(f_tw, (unsigned)px_i<off, G>((int)f_tw, a));
              }
            };
            ring(std::integral_constant<int, 1>{}); ring(std::integral_constant<int, 2>{});
            ring(std::integral_constant<int, 4>{}); ring(std::integral_constant<int, 8>{});
          }
          f_ol = w & 0x3FFu; f_oa = (w >> 10) & 0x3FFu; f_oar = (w >> 20) & 0x3FFu;
          const bool has_tw = MIXED && f_tw != none;
          if (has_tw) f_oa = f_tw & ~1u;  // s_oa = that HDV's (shifted) record; constrain_adj set below
          Neigh nb;
          nb.has_ol = f_ol != none; nb.has_oa = f_oa != none; nb.has_oar = f_oar != none;
          const int s_ol = gb + (nb.has_ol ? (int)((f_ol >> 1) & 15u) : 0);
          const int s_oa = gb + (nb.has_oa ? (int)((f_oa >> 1) & 15u) : 0);
          const int s_oar = gb + (nb.has_oar ? (int)((f_oar >> 1) & 15u) : 0);
          nb.ol_x = shfl_d(whx, s_ol); nb.ol_gu = shfl_d(wgu, s_ol);
          nb.oa_x = shfl_d(whx, s_oa); nb.oa_gu = shfl_d(wgu, s_oa);
          nb.ol_vx = nb.oa_vx = nb.ol_acc = nb.ol_g = nb.oa_acc = nb.oa_g = 0;  // folded into ol_gu / oa_gu
          nb.oar_x = shfl_d(wx, s_oar); nb.oar_vx = shfl_d(wvx, s_oar);
          nb.constrain_adj = MASS && nb.has_oa && (f_oa & 1u);
          bool hss_collab = false;
          if (has_tw) { nb.constrain_adj = MASS; hss_collab = !MASS; }  // cbf.constrain_adj = True (:181)
          // (absent slots contribute g*u = 0, an obstacle +0: handled in shield_dyn<.., PRE> / obstacle_override)
          obstacle_override<MASS>(nb, v.x, v.y);
          const bool mine = has && a == ai && shield_on;  // this stage's ego
          ShieldOut s1 = shield_eval<MASS, true, IPM>(c, v, cpsi, pk_self, nb, mine);
          if (hss_collab) s1.flags |= MM_FLAG_IS_COLLABORATING;  // vehicle.is_collaborating = cbf.constrain_adj
          if (IPM && mine && s1.bounds) atomicOr(c.err, MM_LATCH_QP_BOUNDS);
          if (mine) {
            new_acc = s1.acc; veto = s1.veto; new_flags = s1.flags; qt = s1.qt;
            if (TRACE) trace_status(out.trace + (long long)k * MM_T_COUNT * A + i, A, s1);
            use_B = veto && needB;
            const Cand cc = chosen(use_B);
            double nv = v.v + new_acc * dt;
            nv = nv > 0 ? nv : 0;
            // publish my post-step view (Vehicle.step committed) for the later stages
            wx = cc.x; wy = cc.y; wh = cc.h; wg = cc.gvx; wacc = new_acc; wvx = nv * cc.cpsi;
            whx = s_cold[C_H1X][tid]; whvx = s_cold[C_H1VX][tid]; wpk = cc.pk;
            wgu = slot_gu<MASS>(whvx, MASS ? wacc : kCbfAccLo, wg, dt);
          }
        }
        // the in-place history edits persist in the record that becomes state_hist[-2] (stepped HDV)
        if (MIXED && hdv && twin_shift != 0 && w_stepped) s_cold[C_H1X][tid] = whx;
      }
     }
    }
    STAMP(6);  // serial fallback (if taken) + sweep exit
    if (SPLIT && SHIELDED && out.trace && live && tail && !shield_on) {  // (the sweep kernel wrote the QP planes of the vehicles it ran)
      out.trace[(long long)k * MM_T_COUNT * A + i + MM_T_QP_ROWS * A] = 0;
    }
    if (!SPLIT && SHIELDED && TRACE && live) {  // QP internals go out now so they need not stay in registers
      double *t = out.trace + (long long)k * MM_T_COUNT * A + i;
      t[MM_T_QP_ROWS * A] = qt.rows; t[MM_T_QP_A * A] = qt.a;
      t[MM_T_QP_H0 * A] = qt.h0; t[MM_T_QP_H1 * A] = qt.h1; t[MM_T_QP_H2 * A] = qt.h2;
      t[MM_T_QP_H3 * A] = qt.h3; t[MM_T_QP_D * A] = qt.d; t[MM_T_LC_MARGIN * A] = qt.margin;
    }
    if (tail) {  // (split form: the commit half)
    // ---------------- commit Vehicle.step / MDPLCVehicle.step for every vehicle -------------------
    if (live) {
      const Cand cc = chosen(use_B);
      const double acc = (SHIELDED && shield_on) ? new_acc : v.act_acc;
      double nv = v.v + acc * dt;
      nv = nv > 0 ? nv : 0;  // max(0, speed)
      if (SHIELDED && shield_on) {
        if (veto) v.tlane = v.lane;
        v.flags = new_flags;
      }
      v.x = cc.x; v.y = cc.y; v.h = cc.h; v.v = nv; v.lane = cc.lane;
      if (LC) {
        if (!hdv) {
          // a veto re-steers to the current lane; identical to the nominal command unless B was needed
          // (shielded kernels: the nominal command was parked with candidate A -- its register is free since then)
          double steer_nom = v.act_steer;
          if constexpr (SHIELDED) steer_nom = s_cold[C_A + 5][tid];
          s_cold[C_SSTEER][tid] = (SHIELDED && shield_on && veto && haveB) ? s_cold[C_B + 5][tid] : steer_nom;
          s_cold[C_SACC][tid] = acc; v.gvx = cc.gvx;
          if constexpr (kRoomy) s_cold[C_GVX][tid] = cc.gvx;
          if (sv) v.sang += s_cold[C_SSTEER][tid] * dt;  // steering_angle += safe steering velocity * dt (:139)
        }
        s_cold[C_H2X][tid] = s_cold[C_H1X][tid]; s_cold[C_H2VX][tid] = s_cold[C_H1VX][tid];  // log_step :187-201 (IDMVehicleHist: behavior.py:505-521)
        s_cold[C_H1X][tid] = v.x; s_cold[C_H1VX][tid] = v.v * cc.cpsi;
        if (v.hist_len < 2) v.hist_len++;
      }
      if constexpr (kRoomy) s_cold[C_CPSI][tid] = cc.cpsi;
      else cpsi = cc.cpsi;
      spsi = cc.spsi;
      if (SHIELDED) pk_self = cc.pk;
    }

    STAMP(7);  // commit
    // ---------------- collisions (road.py:288-292, kinematics.py:175-209) ----------------------
    // Which pairs need the 9-point rotated-rectangle test at all?  The reference's own pre-check (norm <= LENGTH) lets
    // through every pair of vehicles side by side on bc0 / bc1 and every vehicle passing the obstacle; boxes_may_touch is
    // an exact early-out on top of it (a pair it rejects would test false), so the full test -- ONE copy of the code, in a
    // loop over the flagged partners -- runs only on near-contacts.
    unsigned need = 0;  // bit p: partner with creation index p; bit G: the obstacle
    for_partners<G>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      double px = px_d<m, G>(v.x, a), py = px_d<m, G>(v.y, a), ph = px_d<m, G>(v.h, a);
      bool pp = px_i<m, G>((int)live, a) != 0;
      double dx = px - v.x, dy = py - v.y;
      const bool near = live & pp & !((dx * dx + dy * dy) > kU5);  // norm > LENGTH pre-check, sqrt-free
      need |= (near & boxes_may_touch(dx, dy, v.h, ph, 0.9 * kVehLength / 2, 0.9 * kVehWidth / 2)) ? 1u << pidx<m, G>(a) : 0u;
    });
    {
      double dx = kObstX - v.x, dy = kObstY - v.y;
      const bool near = live & !((dx * dx + dy * dy) > kU5);
      need |= (near & boxes_may_touch(dx, dy, v.h, 0.0, 0.9 * 2.0 / 2, 0.9 * 2.0 / 2)) ? 1u << G : 0u;
    }
    unsigned hits = 0;
    bool obst_hit = false;
    if (__any(need != 0)) {
      for (int p = 0; p <= G; p++) {
        const bool t = ((need >> p) & 1u) != 0;
        if (!__any(t)) continue;
        const bool is_obst = p == G;
        const int src = gb + (is_obst ? 0 : p);
        double px = shfl_d(v.x, src), py = shfl_d(v.y, src), ph = shfl_d(v.h, src);
        if (is_obst) { px = kObstX; py = kObstY; ph = 0.0; }
        const double ol = is_obst ? 2.0 : kVehLength, ow = is_obst ? 2.0 : kVehWidth;
        if (t && rects_intersect(v.x, v.y, v.h, px, py, ol, ow, ph)) {
          if (is_obst) obst_hit = true;
          else hits |= 1u << p;
        }
      }
    }
    if (__any(hits != 0 || obst_hit)) {
      // order-dependent part: creation-order double loop, min-|speed| hand-down (kinematics.py:187-196)
      unsigned cb = group_ballot<G>(v.crashed != 0, gb);
      const unsigned ob = group_ballot<G>(obst_hit, gb);
      for (int ii = 0; ii < st.N; ii++) {
        const unsigned row = (unsigned)shfl_i((int)hits, gb + ii);
        for (int jj = 0; jj < st.N; jj++) {
          const double si = shfl_d(v.v, gb + ii), sj = shfl_d(v.v, gb + jj);
          const bool fire = !((cb >> ii) & 1u) && ii != jj && ((row >> jj) & 1u);
          if (fire) {
            const double s = fabs(si) <= fabs(sj) ? si : sj;
            if (a == ii || a == jj) { v.v = s; v.crashed = 1; }
            cb |= (1u << ii) | (1u << jj);
          }
        }
        if (!((cb >> ii) & 1u) && ((ob >> ii) & 1u)) {
          if (a == ii) { v.v = fabs(v.v) <= 0 ? v.v : 0.0; v.crashed = 1; }
          cb |= 1u << ii;
        }
      }
    }
    STAMP(8);  // collisions
    if (env_active) time += 1;
    if (TRACE && live && (!SPLIT || out.trace)) {  // (the split form is instantiated once: the trace is a run-time choice there)
      double *t = out.trace + (long long)k * MM_T_COUNT * A + i;
      t[MM_T_X * A] = v.x; t[MM_T_Y * A] = v.y; t[MM_T_HEADING * A] = v.h; t[MM_T_SPEED * A] = v.v;
      double steer_tr = v.act_steer;
      if constexpr (SHIELDED) steer_tr = s_cold[C_A + 5][tid];
      t[MM_T_ACT_STEER * A] = steer_tr; t[MM_T_ACT_ACC * A] = v.act_acc;
      t[MM_T_SAFE_STEER * A] = (LC && !hdv) ? s_cold[C_SSTEER][tid] : steer_tr;
      t[MM_T_SAFE_ACC * A] = (LC && !hdv) ? s_cold[C_SACC][tid] : v.act_acc;
      t[MM_T_LANE * A] = v.lane; t[MM_T_TARGET_LANE * A] = v.tlane; t[MM_T_CRASHED * A] = v.crashed;
      t[MM_T_FLAGS * A] = v.flags;
      if (!SHIELDED) t[MM_T_QP_ROWS * A] = 0;  // the other QP planes keep the NaN fill
    }
    // _is_terminal (merge_env_v1.py:168-172) breaks the sub-step loop (abstract.py:530)
    const bool term = group_ballot<G>(ctrl && (v.crashed || v.x < 0), gb) != 0 || steps >= c.T;
    if (term) env_active = false;
    }  // tail
  }

  STAMP(9);  // trace + terminal
  if constexpr (SPLIT) {
    if (kb < c.nsub) {  // more sub-steps to come: the state planes carry the vehicles to the next launch
      if (valid && v.present) {
        if (kb == 0 && !MIXED) {  // (nothing committed yet: only what the act half changes; general kernels: an HDV's action and timer too)
          st.F[MM_F_TARGET_SPEED * A + i] = s_cold[C_TSPEED][tid];
          st.B[MM_B_TARGET_LANE * A + i] = (uint8_t)v.tlane; st.B[MM_B_SPEED_INDEX * A + i] = (uint8_t)v.sidx;
          st.B[MM_B_HL_ACTION * A + i] = (uint8_t)v.hl;
        } else {
          v.h1x = s_cold[C_H1X][tid]; v.h1vx = s_cold[C_H1VX][tid]; v.h2x = s_cold[C_H2X][tid]; v.h2vx = s_cold[C_H2VX][tid];
          v.safe_steer = s_cold[C_SSTEER][tid]; v.safe_acc = s_cold[C_SACC][tid];
          v.tspeed = s_cold[C_TSPEED][tid];
          store_veh(st, i, v, sv);
        }
      }
      if (e < st.E && a == 0) { st.I[MM_E_STEPS * st.E + e] = steps; st.I[MM_E_TIME * st.E + e] = time; }
      return;
    }
  }
  // ---------------- rewards / info (merge_env_v1.py:59-166, abstract.py:469-498) -----------------
  const bool env_ok = n_ctrl > 0;
  const unsigned crashed_bits = group_ballot<G>(ctrl && v.crashed, gb);
  const bool done = env_ok && (crashed_bits != 0 || steps >= c.T ||
                               group_ballot<G>(ctrl && v.x < 0, gb) != 0);
  const int nl = v.present ? next_lane(v.lane, v.x, v.y) : 0;
  // surrounding_vehicles lane sets (road.py:315-342), as bit sets over lane ids
  const unsigned allow_tbl[6] = {
      (1u << MM_LANE_AB0) | (1u << MM_LANE_BC0), (1u << MM_LANE_AB0) | (1u << MM_LANE_BC0) | (1u << MM_LANE_CD0),
      (1u << MM_LANE_KB0) | (1u << MM_LANE_BC1), (1u << MM_LANE_BC0) | (1u << MM_LANE_CD0),
      (1u << MM_LANE_JK0) | (1u << MM_LANE_KB0), (1u << MM_LANE_JK0) | (1u << MM_LANE_KB0) | (1u << MM_LANE_BC1)};
  unsigned allow_own = 0, allow_side = 0;
  {
    int side = -1;  // _regional_reward side / ramp lane choice (merge_env_v1.py:95-118)
    const int l = v.lane;
    if (l == MM_LANE_BC0) side = MM_LANE_BC1;
    else if (l == MM_LANE_AB0 && v.x > 220) side = MM_LANE_KB0;
    else if (l == MM_LANE_BC1) side = MM_LANE_BC0;
    else if (l == MM_LANE_KB0) side = MM_LANE_AB0;
#pragma unroll
    for (int q = 0; q < 6; q++) {
      if (l == q) allow_own = allow_tbl[q];
      if (side == q) allow_side = allow_tbl[q];
    }
  }
  double hd = 60;  // _compute_headway_distance abstract.py:620-635
  double of_s = 0, or_s = 0, sf_s = 0, sr_s = 0;
  int of_i = -1, or_i = -1, sf_i = -1, sr_i = -1;  // own/side chain front & rear
  unsigned hdv_bits = 0;                           // which vehicles of the env are HDVs
  for_partners<G>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    const int p = pidx<m, G>(a);
    double px = px_d<m, G>(v.x, a);
    int pk = px_i<m, G>(v.present ? v.lane : 15, a);
    bool pp = pk != 15;
    if (MIXED) hdv_bits |= (px_i<m, G>((int)hdv, a) != 0) ? (1u << p) : 0u;
    if (pp && pk == v.lane && px > v.x) { double dd = px - v.x; if (dd < hd) hd = dd; }
    if (pp && v.lane != MM_LANE_BC1 && pk == nl && px > v.x) { double dd = px - v.x; if (dd < hd) hd = dd; }
    const bool in_own = pp && ((allow_own >> pk) & 1u), in_side = pp && ((allow_side >> pk) & 1u);
    // road.py:344-349: front = min s_v >= s (ties: later index), rear = max s_v < s (ties: earlier)
    if (in_own && v.x <= px && (of_i < 0 || px < of_s || (px == of_s && p > of_i))) { of_s = px; of_i = p; }
    if (in_own && px < v.x && (or_i < 0 || px > or_s || (px == or_s && p < or_i))) { or_s = px; or_i = p; }
    if (in_side && v.x <= px && (sf_i < 0 || px < sf_s || (px == sf_s && p > sf_i))) { sf_s = px; sf_i = p; }
    if (in_side && px < v.x && (sr_i < 0 || px > sr_s || (px == sr_s && p < sr_i))) { sr_s = px; sr_i = p; }
  });
  // _agent_reward merge_env_v1.py:64-89
  double local = 0;
  if (ctrl) {
    // default reward, or MergeEnvLCMARL's "srew" / "mrew" variants (merge_env_v1.py:439-474)
    const bool xrew = LC && c.agent_reward != 0, is_mrew = c.agent_reward == 2;
    const bool half = xrew && is_mrew && (v.flags & MM_FLAG_IS_COLLABORATING);  // halved speed range
    double scaled = 0 + (half ? div_c((v.v - c.rs_lo) * (1 - 0), c.rs_span2, c.rs_ispan2)
                              : div_c((v.v - c.rs_lo) * (1 - 0), c.rs_span, c.rs_ispan));
    double merging = 0;
    if (v.lane == MM_LANE_BC1 && (!(xrew && is_mrew) || (v.flags & MM_FLAG_IS_LC_SAFE))) {
      double t = v.x - 420;
      merging = -mmm_exp(MM_DIVC(-(t * t), 1000.0));
    }
    double hc = v.v > 0 ? mmm_log(hd / (c.headway_time * v.v)) : 0;
    if (xrew) hc = -1 * hc;  // srew / mrew flip the sign of the headway term (:465-466)
    local = c.collision_reward * (-1 * v.crashed) + (c.high_speed_reward * clipd(scaled, 0, 1)) +
            c.merging_lane_cost * merging + c.headway_cost * (hc < 0 ? hc : 0);
  }
  // regional reward: mean over [v_fl, v_fr, self, v_rl, v_rr] (merge_env_v1.py:119-124)
  double regional = 0;
  {
    const bool on_main = v.lane == MM_LANE_AB0 || v.lane == MM_LANE_BC0 || v.lane == MM_LANE_CD0;
    const int i_fl = on_main ? of_i : sf_i, i_rl = on_main ? or_i : sr_i;
    const int i_fr = on_main ? sf_i : of_i, i_rr = on_main ? sr_i : or_i;
    const double r_fl = shfl_d(local, gb + (i_fl < 0 ? 0 : i_fl)), r_fr = shfl_d(local, gb + (i_fr < 0 ? 0 : i_fr));
    const double r_rl = shfl_d(local, gb + (i_rl < 0 ? 0 : i_rl)), r_rr = shfl_d(local, gb + (i_rr < 0 ? 0 : i_rr));
    auto is_mdp = [&](int j) { return j >= 0 && !((hdv_bits >> j) & 1u); };  // isinstance(v, MDPVehicle) :121
    double sum = 0;
    int cnt = 0;
    if (is_mdp(i_fl)) { sum += r_fl; cnt++; }
    if (is_mdp(i_fr)) { sum += r_fr; cnt++; }
    sum += local; cnt++;
    if (is_mdp(i_rl)) { sum += r_rl; cnt++; }
    if (is_mdp(i_rr)) { sum += r_rr; cnt++; }
    regional = sum / cnt;
  }
  // env-level sums in creation order (Python sum / += order)
  double rsum = 0, ssum = 0, tsum = 0;
  for (int q = 0; q < st.N; q++) {
    const double lr = shfl_d(local, gb + q), sp = shfl_d(v.v, gb + q);
    if ((ctrl_bits >> q) & 1u) { rsum += lr; ssum += sp; }
    if ((present_bits >> q) & 1u) tsum += sp;  // traffic_speed over road.vehicles (:147-151)
  }
  const double reward = env_ok ? rsum / n_ctrl : 0, avg_speed = env_ok ? ssum / n_ctrl : 0;
  const double traffic_speed = env_ok ? tsum / n_veh : 0;
  // _compute_min_time_headway merge_env_v1.py:373-386
  double th = INFINITY;
  if (ctrl) {
    double h2d = hd;
    if (fabs(kObstY - v.y) <= 2 && kObstX > v.x) { double dd = kObstX - v.x; if (dd < h2d) h2d = dd; }
    h2d = h2d - kVehLength;
    double vx = v.v * cpsi;  // cos(heading), carried along
    th = h2d / (vx > 1 ? vx : 1);
  }
  const double min_headway = group_min_d<G>(th, a);
  if (e < st.E) { n_merge = st.I[MM_E_N_MERGE * st.E + e]; episode = st.I[MM_E_EPISODE * st.E + e]; }
  double merge_pct = __builtin_nan("");
  if (done) {
    const int n_rem = __popc(group_ballot<G>(
        ctrl && (v.lane == MM_LANE_BC1 || v.lane == MM_LANE_KB0 || v.lane == MM_LANE_JK0), gb));
    merge_pct = n_merge > 0 ? (double)(n_merge - n_rem) / n_merge * 100 : 100.0;
  }
  if (valid) {
    if (out.agents_rewards) out.agents_rewards[i] = ctrl ? local : 0;
    if (out.regional_rewards) out.regional_rewards[i] = ctrl ? regional : 0;
    if (out.agents_dones) out.agents_dones[i] = ctrl ? (uint8_t)(v.crashed || steps >= c.T || v.x < 0) : 1;
    if (out.crashed) out.crashed[i] = v.present ? (uint8_t)v.crashed : 0;
    if (out.agents_info) {
      out.agents_info[i * 3 + 0] = v.present ? v.x : 0;
      out.agents_info[i * 3 + 1] = v.present ? v.y : 0;
      out.agents_info[i * 3 + 2] = v.present ? v.v : 0;
    }
  }
  if (e < st.E && a == 0 && env_ok) {
    if (out.reward) out.reward[e] = reward;
    if (out.done) out.done[e] = (uint8_t)done;
    if (out.average_speed) out.average_speed[e] = avg_speed;
    if (out.traffic_speed) out.traffic_speed[e] = traffic_speed;
    if (out.min_headway) out.min_headway[e] = min_headway;
    if (out.merge_percent) out.merge_percent[e] = merge_pct;
  }
  STAMP(10);  // rewards + outputs
  // ---------------- rollout metrics: one 64-byte partial per WAVE, no block barrier, no contended atomics -
  // Each env's leader lane parks its 8 contributions in its own (now dead) candidate columns of the cold slots; lanes
  // 0..7 of the wave then each sum one metric over the wave's env leaders and store (or, with mm_defer_metrics, add to)
  // the wave's slot of the partial buffer; metrics_flush_kernel -- launched right behind this kernel by mm_step, or once
  // per rollout by mm_flush_metrics when the metrics are deferred -- folds the partials into the caller's 8 doubles.
  // (Round 1 used LDS atomics between two __syncthreads() + 8 global atomics per block: 15 % of a wave's lifetime parked
  // at the barriers.)
  if (metrics) {
    static_assert(MM_STEP_BLOCK % 64 == 0, "whole waves per workgroup");
    const bool lead = e < st.E && a == 0 && env_ok;
    const double mv[8] = {lead ? reward : 0.0, (lead && done && crashed_bits) ? 1.0 : 0.0, lead ? avg_speed : 0.0,
                          lead ? traffic_speed : 0.0, lead ? 1.0 : 0.0, (lead && done) ? merge_pct : 0.0,
                          (lead && done) ? 1.0 : 0.0, lead ? min_headway : INFINITY};
    if (a == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) s_cold[C_B + k][tid] = mv[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ln = lane_id();
    if (ln < 8) {
      const int slot = C_B + ln, base = tid - ln;  // (tid - lane = first column of this wave, for any MM_STEP_BLOCK)
      double acc = ln == 7 ? INFINITY : 0.0;
#pragma unroll
      for (int j = 0; j < 64 / G; j++) {
        const double x = s_cold[slot][base + j * G];
        acc = ln == 7 ? fmin(acc, x) : acc + x;
      }
      double *row = metrics + (gtid >> 6) * 8;  // this wave's row of the partial buffer
      if ((c.debug_flags >> kMetricsAccBit) & 1) {  // deferred folding (mm_defer_metrics): the row accumulates across launches; fire and forget
        if (ln == 7) (void)unsafeAtomicMin(&row[7], acc);
        else (void)unsafeAtomicAdd(&row[ln], acc);
      } else {
        row[ln] = acc;
      }
    }
  }
  STAMP(15);  // metrics
  // ---------------- optional re-spawn (marl/mappo.py:133-135 `if done: env.reset()`) --------------
  if (c.auto_reset && done) {
    const uint64_t seed = st.seeds[e];
    Veh nv;
    memset(&nv, 0, sizeof nv);
    int nm = 0;
    int nc = n_ctrl, nh = n_veh - n_ctrl;  // fixed counts: the env keeps its own composition
    episode_counts(c, seed, (uint32_t)episode, nc, nh);
    const bool spawn = valid && (c.traffic_density > 0 ? a < nc + nh : v.present);
    if (spawn) nm = spawn_vehicle(nv, a, nc, nh, seed, (uint32_t)episode);
    if (c.traffic_density > 0 && valid) {  // ragged batch: the slot's occupancy changes with the episode
      if (!spawn) {
        if (v.present) {  // the slot empties: its planes read zero from now on (not whichever store happened to be the last one)
          Veh z;
          memset(&z, 0, sizeof z);
          store_veh(st, i, z, true);
        }
        v.present = false; v.kind = 0;
      }
      st.B[MM_B_KIND * A + i] = spawn ? (uint8_t)nv.kind : (uint8_t)0;
    }
    if (spawn) {
      v = nv;
      spsi = 0.0; cpsi = 1.0;  // heading 0
      s_cold[C_H1X][tid] = v.h1x; s_cold[C_H1VX][tid] = v.h1vx; s_cold[C_H2X][tid] = v.h2x; s_cold[C_H2VX][tid] = v.h2vx;
      s_cold[C_SSTEER][tid] = v.safe_steer; s_cold[C_SACC][tid] = v.safe_acc; s_cold[C_TSPEED][tid] = v.tspeed;
    }
    n_merge = shfl_i(nm, gb);
    steps = 0; time = 0; episode += 1;
  }
  if (valid && v.present) {
    if (LC) {
      v.h1x = s_cold[C_H1X][tid]; v.h1vx = s_cold[C_H1VX][tid]; v.h2x = s_cold[C_H2X][tid]; v.h2vx = s_cold[C_H2VX][tid];
      if (!hdv) { v.safe_steer = s_cold[C_SSTEER][tid]; v.safe_acc = s_cold[C_SACC][tid]; }
    }
    v.tspeed = s_cold[C_TSPEED][tid];
    store_veh(st, i, v, sv);
  }
  if (e < st.E && a == 0) {
    st.I[MM_E_STEPS * st.E + e] = steps; st.I[MM_E_TIME * st.E + e] = time;
    st.I[MM_E_N_MERGE * st.E + e] = n_merge; st.I[MM_E_EPISODE * st.E + e] = episode;
  }
  STAMP(11);  // re-spawn + state store
  __syncthreads();  // every wave is done with its cold slots: the obs staging below reuses that LDS
  STAMP(14);  // barrier (+ drain of the stores issued before it)
  const double sc_h[2] = {spsi, cpsi};
  observe<G, KIND, true>(c, v, a, gb, i, valid, out.obs, out.action_mask, (float *)&s_cold[0][0], sc_h);
  STAMP(12);  // observation
#ifdef MM_STAMPS
  {
    const long long wv = gtid >> 6;
    if ((threadIdx.x & 63) < 16 && wv < MM_STAMP_WAVES) g_stamps_w[wv * 16 + (threadIdx.x & 63)] += _t_acc;
  }
#endif
}

// reset / init / observe --------------------------------------------------------------------------
// mode 0: device-RNG spawn (mm_reset), 1: finish a host-provided spawn (mm_init_from_kinematics),
// 2: observe only (mm_observe)
template <int G, int KIND>
__global__ __launch_bounds__(256) void reset_kernel(DevCfg c, DevState st, int mode, const uint8_t *env_mask,
                                                    const uint64_t *seeds_in, void *obs, uint8_t *avail) {
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e = gtid / G;
  const int a = (int)(gtid % G);
  const int gb = lane_id() & ~(G - 1);
  const bool valid = e < st.E && a < st.N;
  const long long i = e * st.N + a;
  const bool selected = e < st.E && mode != 2 && (!env_mask || env_mask[e]);
  Veh v;
  load_veh(st, i, valid, v);
  if (selected) {
    int steps = 0, time = 0, n_merge = 0, episode = st.I[MM_E_EPISODE * st.E + e];
    if (mode == 0) {
      if (seeds_in && a == 0) st.seeds[e] = seeds_in[e];
      const uint64_t seed = seeds_in ? seeds_in[e] : st.seeds[e];
      int nm = 0;
      int nc = st.N - c.n_hdv, nh = c.n_hdv;
      episode_counts(c, seed, (uint32_t)episode, nc, nh);
      if (valid) {
        if (a < nc + nh) nm = spawn_vehicle(v, a, nc, nh, seed, (uint32_t)episode);
        else {  // unused slot of a ragged batch (a slot that empties reads zero in every plane)
          const bool was = v.present;
          memset(&v, 0, sizeof v);
          if (was) store_veh(st, i, v, true);
          v.hl = MM_HL_NONE;
        }
        st.B[MM_B_KIND * st.A + i] = (uint8_t)v.kind;
      }
      n_merge = shfl_i(nm, gb);
      episode += 1;
    } else {
      if (valid && v.present) init_vehicle(v);
      n_merge = __popc(group_ballot<G>(valid && v.kind == 1 && (v.lane == MM_LANE_JK0 || v.lane == MM_LANE_KB0), gb));
    }
    if (valid && v.present) store_veh(st, i, v, true);
    if (a == 0) {
      st.I[MM_E_STEPS * st.E + e] = steps; st.I[MM_E_TIME * st.E + e] = time;
      st.I[MM_E_N_MERGE * st.E + e] = n_merge; st.I[MM_E_EPISODE * st.E + e] = episode;
    }
  }
  observe<G, KIND>(c, v, a, gb, i, valid, obs, avail);
}


// ------------------------------------------------------------------------------------------------
// stand-alone safety_layer(...) for every controlled vehicle on the current state (mm_shield_actions)
// ------------------------------------------------------------------------------------------------
template <int G, int SHIELD, bool IPM = false>
__global__ __launch_bounds__(256) void shield_kernel(DevCfg c, DevState st, const double *__restrict__ act_steer,
                                                     const double *__restrict__ act_acc, double *__restrict__ safe_steer,
                                                     double *__restrict__ safe_acc, uint8_t *__restrict__ status,
                                                     double *__restrict__ margin, double *__restrict__ headway) {
  constexpr bool MASS = (SHIELD == MM_SHIELD_MASS);
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e = gtid / G;
  const int a = (int)(gtid % G);
  const bool valid = e < st.E && a < st.N;
  const long long i = e * st.N + a;
  Veh v;
  load_veh(st, i, valid, v, c.steer_vel != 0);
  const bool hdv = v.kind == 2;
  const bool ctrl = v.present && !hdv;
  if (valid) { v.act_steer = act_steer[i]; v.act_acc = act_acc[i]; }
  const bool on = ctrl && SHIELD != MM_SHIELD_NONE && v.hist_len >= 2;  // gate safe_controller.py:229-239
  double cpsi = 1.0, spsi = 0.0;
  if (v.present) mmm_sincos(v.h, &spsi, &cpsi);
  int pk_self = v.lane;
  if (v.present) pk_self = pose_code(v.lane, next_lane(v.lane, v.x, v.y), MASS ? corner_flags(v.x, v.y, spsi, cpsi, v.lane) : 0);
  // every other vehicle is seen in its CURRENT state: records [-2], last safe_action, g.vx as stored
  double k_ol = INFINITY, k_oa = INFINITY, k_oar = INFINITY;
  int j_ol = -1, j_oa = -1, j_oar = -1;
  Neigh nb;
  memset(&nb, 0, sizeof nb);
  unsigned tw_mask = 0;  // partners (by xor mask m) in the on-ramp HDV twin class
  double keys[G];
  keys[0] = INFINITY;
  const double evx_raw = v.v * cpsi;  // vehicle.velocity[0]
  for_partners<G>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    const int p = a ^ m;
    const double ox = dppx_d<m>(v.x), oy = dppx_d<m>(v.y), oh = dppx_d<m>(v.h);
    const int opk = dppx_i<m>(pk_self | (int)v.present << 8 | (int)hdv << 9);
    const double ohx = dppx_d<m>(v.h2x), ohvx = dppx_d<m>(v.h2vx);
    const double og = dppx_d<m>(hdv ? 1.0 : v.gvx), oacc = dppx_d<m>(hdv ? kCbfAccLo : v.safe_acc);
    const double ovx = dppx_d<m>(v.v * cpsi);
    const bool o_hdv = ((opk >> 9) & 1) != 0;
    const Rel r = relate(v.x, v.y, pk_self, ((opk >> 8) & 1) != 0, ox, oy, oh, opk, o_hdv);
    keys[m] = r.key;
    if (r.cls == 1 && (r.key < k_ol || (r.key == k_ol && p < j_ol))) { k_ol = r.key; j_ol = p; nb.ol_x = ohx; nb.ol_vx = ohvx; nb.ol_g = og; nb.ol_acc = oacc; }
    if (r.cls == 2 && (r.key < k_oa || (r.key == k_oa && p < j_oa))) { k_oa = r.key; j_oa = p; nb.oa_x = ohx; nb.oa_vx = ohvx; nb.oa_g = og; nb.oa_acc = oacc; nb.constrain_adj = r.cflag; }
    if (r.cls == 3 && (r.key < k_oar || (r.key == k_oar && p < j_oar))) { k_oar = r.key; j_oar = p; nb.oar_x = ox; nb.oar_vx = ovx; }
    if (r.cls == 4) tw_mask |= 1u << m;
  });
  int pos_ol = 0, pos_oa = 0, pos_oar = 0, n_close = 0;
#pragma unroll
  for (int m = 1; m < G; m++) {
    const int p = a ^ m;
    pos_ol += (keys[m] < k_ol || (keys[m] == k_ol && p < j_ol)) ? 1 : 0;
    pos_oa += (keys[m] < k_oa || (keys[m] == k_oa && p < j_oa)) ? 1 : 0;
    pos_oar += (keys[m] < k_oar || (keys[m] == k_oar && p < j_oar)) ? 1 : 0;
    n_close += keys[m] < INFINITY ? 1 : 0;
  }
  nb.has_ol = j_ol >= 0 && pos_ol < 5;
  nb.has_oa = j_oa >= 0 && pos_oa < 5;
  nb.has_oar = j_oar >= 0 && pos_oar < 5;
  nb.constrain_adj = MASS && nb.has_oa && nb.constrain_adj;
  bool hss_collab = false;
  {
    // on-ramp HDV twin (:162-184): among the twins within the 5 nearest, the LAST one in sorted order
    // takes the adjacent slot (the branch has no `is None` guard); its record is read shifted
    int m_best = 0, pos_best = -1;
#pragma unroll
    for (int m = 1; m < G; m++) {
      if ((tw_mask >> m) & 1u) {
        int pos = 0;
#pragma unroll
        for (int m2 = 1; m2 < G; m2++)
          if (m2 != m) pos += (keys[m2] < keys[m] || (keys[m2] == keys[m] && (a ^ m2) < (a ^ m))) ? 1 : 0;
        if (pos < 5 && pos > pos_best) { pos_best = pos; m_best = m; }
      }
    }
    double tw_x = 0, tw_vx = 0;
    for_partners<G>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const double ohx = dppx_d<m>(v.h2x), ohvx = dppx_d<m>(v.h2vx);
      if (m == m_best) { tw_x = ohx + 0.5 * evx_raw; tw_vx = ohvx; }
    });
    if (pos_best >= 0) {
      nb.has_oa = true; nb.oa_x = tw_x; nb.oa_vx = tw_vx; nb.oa_acc = kCbfAccLo; nb.oa_g = 1.0;
      nb.constrain_adj = MASS; hss_collab = !MASS;
    }
  }
  if (!nb.has_ol) { nb.ol_acc = 0; nb.ol_g = 0; }
  if (!nb.has_oa) { nb.oa_acc = 0; nb.oa_g = 0; }
  if (!MASS) { nb.ol_acc = kCbfAccLo; nb.oa_acc = kCbfAccLo; }
  obstacle_override<MASS>(nb, v.x, v.y);
  ShieldOut so = shield_eval<MASS, false, IPM>(c, v, cpsi, pk_self, nb, valid && on);
  if (hss_collab) so.flags |= MM_FLAG_IS_COLLABORATING;
  if (IPM && valid && on && so.bounds) atomicOr(c.err, MM_LATCH_QP_BOUNDS);
  if (valid) {
    double ss = v.act_steer, sa = v.act_acc;
    unsigned stt = 0;
    if (on) {
      sa = so.acc;
      if (so.veto) {  // target_lane_index = lane_index (:501-506)
        ss = steering_control(v.x, v.y, v.h, v.v, v.lane);
        if (c.steer_vel) ss = steer_vel_command(ss, v.sang);
      }
      stt = MM_ST_RAN | (so.optimal ? MM_ST_IS_OPTIMAL : 0u) | (so.lon_safe ? MM_ST_IS_SAFE : 0u) | (so.lon_invariant ? MM_ST_IS_INVARIANT : 0u) |
            (so.bounds ? MM_ST_QP_BOUNDS : 0u) | ((so.flags & MM_FLAG_IS_LC_SAFE) ? MM_ST_IS_LC_SAFE : 0u) |
            ((so.flags & MM_FLAG_IS_COLLABORATING) ? MM_ST_IS_COLLABORATING : 0u) |
            ((so.flags & MM_FLAG_COLLABORATE_ADJ) ? MM_ST_COLLABORATE_ADJ : 0u);
    }
    safe_steer[i] = ss; safe_acc[i] = sa;
    if (status) status[i] = (uint8_t)stt;
    if (margin) margin[i] = on ? so.qt.margin : __builtin_nan("");
    if (headway) headway[i] = on ? so.hw_num / so.hw_den : __builtin_nan("");  // vehicle.set_min_headway (decentral_layer.py:466,700)
  }
}

// stand-alone batched shield QP (cbf.py:110-161), one thread (lane) per QP: the exact KKT point, or (solver ==
// MM_QP_IPM) the iterate of cvxopt's interior-point algorithm with its status and iteration count (include/mm_qp.h)
#if MM_TU <= 1
__global__ void qp_kernel(int n, const double *__restrict__ G, const double *__restrict__ h,
                          const int32_t *__restrict__ rows, int solver, double *__restrict__ u, uint8_t *status,
                          int32_t *iters, int *err) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double *g = G + (long long)k * 12, *hh = h + (long long)k * 4;
  const int m = rows[k];
  // only the G of get_G (cbf.py:288-304,386-403): [a 0 -1; 1 0 0; -1 0 0] (+ [a 0 -1])
  bool ok = (m == 3 || m == 4) && g[1] == 0.0 && g[2] == -1.0 && g[3] == 1.0 && g[4] == 0.0 && g[5] == 0.0 && g[6] == -1.0 &&
            g[7] == 0.0 && g[8] == 0.0;
  if (ok && m == 4) ok = g[9] == g[0] && g[10] == 0.0 && g[11] == -1.0;
  if (iters) iters[k] = 0;
  if (!ok) {
    u[k * 3 + 0] = u[k * 3 + 1] = u[k * 3 + 2] = __builtin_nan("");
    if (status) status[k] = MM_QPS_BAD_STRUCTURE;
    atomicOr(err, MM_LATCH_BAD_QP);
    return;
  }
  const double a = g[0];
  if (solver == MM_QP_IPM) {
    double d, sl;
    int it;
    const int opt = mm_qp_ipm_cbf(a, hh[0], hh[1], hh[2], m == 4 ? hh[3] : 0.0, m, &d, &sl, &it);
    u[k * 3 + 0] = d; u[k * 3 + 1] = 0.0; u[k * 3 + 2] = sl;
    if (status) status[k] = opt ? MM_QPS_OPTIMAL : MM_QPS_UNKNOWN;
    if (iters) iters[k] = it;
    return;
  }
  double hc = hh[0];
  if (m == 4 && hh[3] < hc) hc = hh[3];
  double d;
  if (a > 0) d = fmin(0.0, hc / a);
  else if (a < 0) d = fmax(0.0, hc / a);
  else d = 0.0;
  d = fmin(fmax(d, -hh[2]), hh[1]);
  const double s = a * d - hc;
  u[k * 3 + 0] = d; u[k * 3 + 1] = 0.0; u[k * 3 + 2] = s > 0 ? s : 0.0;
  if (status) status[k] = MM_QPS_OPTIMAL;
}

#endif  // MM_TU <= 1

// folds the per-wave metric partials of one step launch into the caller's accumulator (7 sums + 1 min): each block takes
// 256 waves' partials (8 independent loads per thread in flight), reduces them in LDS and issues 8 atomics
#if MM_TU <= 1
constexpr int kFlushWaves = 256;
// reset: the partials were accumulated over several launches (mm_defer_metrics) -- put every row back to the identity
__global__ __launch_bounds__(256) void metrics_flush_kernel(double *partial, long long waves, double *metrics, int reset) {
  __shared__ double s_p[32][8];
  const int k = threadIdx.x & 7, r = threadIdx.x >> 3;
  const long long w0 = (long long)blockIdx.x * kFlushWaves;
  double x[kFlushWaves / 32];
#pragma unroll
  for (int j = 0; j < kFlushWaves / 32; j++) {
    const long long w = w0 + r + 32 * j;
    x[j] = w < waves ? partial[w * 8 + k] : (k == 7 ? INFINITY : 0.0);
    if (reset && w < waves) partial[w * 8 + k] = k == 7 ? INFINITY : 0.0;
  }
  double acc = k == 7 ? INFINITY : 0.0;
#pragma unroll
  for (int j = 0; j < kFlushWaves / 32; j++) acc = k == 7 ? fmin(acc, x[j]) : acc + x[j];
  if (reset == 2) return;  // (mm_defer_metrics: rows to the identity, nothing folded -- block-uniform)
  s_p[r][k] = acc;
  __syncthreads();
  if (threadIdx.x < 8) {
    double t = k == 7 ? INFINITY : 0.0;
    for (int j = 0; j < 32; j++) t = k == 7 ? fmin(t, s_p[j][k]) : t + s_p[j][k];
    if (k == 7) atomic_min_d(&metrics[7], t);
    else atomicAdd(&metrics[k], t);
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------------
static int group_size(int N) { return N <= 2 ? 2 : (N <= 4 ? 4 : (N <= 8 ? 8 : 16)); }

struct MMHandle_ {
  MMConfig cfg;
  int E, N, device;
  unsigned char *state;
  MMStateLayout lay;
  long long first_env;
  double *metrics;          // caller's 8 doubles (mm_set_metrics_buffer) or NULL
  double *metrics_partial;  // [waves of a step launch][8], device, owned by the handle
  int metrics_deferred;     // mm_defer_metrics: the partials accumulate across launches, folded by mm_flush_metrics only
  // device error latches (MM_LATCH_* bits), hipMalloc'd at create.  One word per entry point that reports synchronously, so
  // that a condition latched by an un-polled mm_step cannot fail a later, valid mm_shield_qp / mm_shield_actions call (or be
  // consumed by it): [MM_LW_STEP] mm_step -> mm_poll_errors, [MM_LW_QP] mm_shield_qp, [MM_LW_SHIELD] mm_shield_actions
  int *dev_err;
  // hand-off planes of the split interior-point step (SweepBuf): allocated by mm_create / mm_set_config when the
  // configuration steps that way (shielded v1, qp_solver = MM_QP_IPM, CAV-only), never inside mm_step
  SweepBuf sweep;
  void *sweep_mem;
  int n_simd;  // SIMDs of the handle's device (4 per CU), read at mm_create
  char err[256];
};
static bool needs_general(const MMHandle_ *h) {  // HDVs can appear, or steer_vel lateral control: the kernels that carry IDM / MOBIL
  return h->cfg.n_hdv > 0 || (h->cfg.traffic_density > 0 && h->cfg.mixed_traffic != 0) ||
         (h->cfg.env_kind == MM_ENV_V1 && h->cfg.lateral_control == MM_LATERAL_STEER_VEL);
}
// The interior-point mode of a CAV-only shielded batch steps as phase kernels + sweep kernels (SweepBuf above) once the
// batch is large enough for it.  The split step takes as long as its slowest env needs for its QP chain, almost independent of
// the batch (one lane per env, one wave per SIMD up to 65 536 envs); the fused kernel iterates with ~8 of 64 lanes busy but a
// small batch leaves the chip's other lanes idle anyway.  Measured at N = 8 MASS (ms per step, fused / split; tools/split_crossover.sh):
// 8 192 envs 1.12 / 1.31, 12 288 envs 1.43 / 1.33, 16 384 envs 1.46 / 1.36, 32 768 envs 2.46 / 1.54; N = 4 HSS: 4 096 envs 0.30 /
// 0.64, 16 384 envs 0.32 / 0.73.  Crossover = more than one fused wave per SIMD for the 8- and 16-lane groups (long chains per
// env), more than two for the 2- and 4-lane groups.  debug_flags bit2 keeps the fused kernel, bit3 forces the split step at any size
// (validation / A-B timing: same results either way).
static bool steps_split(const MMHandle_ *h) {
  if (!(h->cfg.env_kind == MM_ENV_V1 && h->cfg.shield != MM_SHIELD_NONE && h->cfg.qp_solver == MM_QP_IPM)) return false;
  if (h->cfg.debug_flags & 4) return false;
  if (h->cfg.debug_flags & 8) return true;
  const int g = h->N <= 2 ? 2 : (h->N <= 4 ? 4 : (h->N <= 8 ? 8 : 16));
  return (long long)h->E * g / 64 > (g >= 8 ? 1ll : 2ll) * (h->n_simd > 0 ? h->n_simd : 1024);
}

// waves one step launch starts (launch_step_t rounds the grid up to whole MM_STEP_BLOCK-thread blocks): every one of them
// stores its 64-byte slot of the metrics partial buffer
// waves of a step launch in the power-of-two layout: the size of the metrics partial buffer (a launch in a rotation layout
// has fewer waves -- more envs per wave -- never more)
static long long step_launch_waves(const MMHandle h) {
  const long long threads = (long long)h->E * group_size(h->N);
  return (threads + MM_STEP_BLOCK - 1) / MM_STEP_BLOCK * (MM_STEP_BLOCK / 64);
}
// Lanes per env group of the step launch: batches of 5..6 / 9..12 vehicles run the 6- / 12-lane rotation layouts (kPow2
// above: 10 / 5 envs per wave instead of 8 / 4, 5 / 11 partners per loop instead of 7 / 15), the others power-of-two groups.
static int step_group(const MMHandle h) {
  const int g = group_size(h->N);
#if defined(MM_ONLY_G)  // tuning builds: the one group size that was compiled
  (void)g;
  return MM_ONLY_G;
#elif defined(MM_NO_LANES)
  return g;
#else
  if (h->cfg.debug_flags & 2) return g;  // (debug_flags bit1: validation / A-B timing against the power-of-two groups)
  if (h->N == 5 || h->N == 6) return 6;
  if (h->N >= 9 && h->N <= 12) return 12;
  return g;
#endif
}
static long long step_launch_waves_now(const MMHandle h) {
  const int g = step_group(h);
  if ((g & (g - 1)) == 0) return step_launch_waves(h);
  const long long epw = 64 / g;
  return (h->E + epw - 1) / epw;
}
#if MM_TU <= 1
static uint64_t align256(uint64_t x) { return (x + 255u) & ~(uint64_t)255u; }

extern "C" int32_t mm_abi_version(void) { return MM_ABI_VERSION; }

extern "C" int32_t mm_state_layout(int32_t E, int32_t N, MMStateLayout *out) {
  if (!out || E <= 0 || N <= 0 || N > MM_MAX_AGENTS) return MM_ERR_INVALID_ARG;
  uint64_t A = (uint64_t)E * (uint64_t)N, off = 0;
  out->f64_offset = off; off = align256(off + A * 8u * MM_F_COUNT);
  out->u8_offset = off; off = align256(off + A * MM_B_COUNT);
  out->env_offset = off; off = align256(off + (uint64_t)E * 4u * MM_E_COUNT);
  out->seed_offset = off; off = align256(off + (uint64_t)E * 8u);
  out->total_bytes = off;
  return MM_OK;
}

static int check_cfg(const MMConfig *c, int N, char *err) {
  if (!c || c->abi_version != MM_ABI_VERSION) { snprintf(err, 256, "ABI version mismatch"); return MM_ERR_INVALID_ARG; }
  if (c->env_kind != MM_ENV_V0 && c->env_kind != MM_ENV_V1) { snprintf(err, 256, "unknown env_kind %d", c->env_kind); return MM_ERR_INVALID_ARG; }
  if (c->shield < MM_SHIELD_NONE || c->shield > MM_SHIELD_MASS) { snprintf(err, 256, "Undefined safety_type:%d", c->shield); return MM_ERR_INVALID_ARG; }
  if (c->policy_frequency <= 0 || c->simulation_frequency < c->policy_frequency ||
      c->simulation_frequency / c->policy_frequency > 3) { snprintf(err, 256, "unsupported frequencies"); return MM_ERR_INVALID_ARG; }
  if (N > 12) { snprintf(err, 256, "N=%d exceeds the 6+6 spawn slots", N); return MM_ERR_INVALID_ARG; }
  if (c->n_hdv < 0 || c->n_hdv >= N) { snprintf(err, 256, "n_hdv=%d must leave at least one controlled vehicle of N=%d", c->n_hdv, N); return MM_ERR_INVALID_ARG; }
  if (c->qp_solver != MM_QP_EXACT && c->qp_solver != MM_QP_IPM) { snprintf(err, 256, "unknown qp_solver %d", c->qp_solver); return MM_ERR_INVALID_ARG; }
  // every vehicle composition this configuration can produce -- fixed counts, or any per-episode draw incl. the
  // reset(num_CAV=k) override -- must fit the N slots and the six spawn points per road (the reference raises from
  // np.random.choice(replace=False), merge_env_v1.py:284-320); checked here so that mm_create / mm_set_config refuse it
  // before any (auto-)reset can meet it
  if (mm_counts_check(c, N, 0, err, 256)) return MM_ERR_INVALID_ARG;
  return MM_OK;
}

// device memory of the hand-off planes (idempotent; the caller holds a DeviceGuard)
static hipError_t ensure_sweep(MMHandle_ *h) {
  if (h->sweep_mem || !steps_split(h)) return hipSuccess;
  const uint64_t Ep = ((uint64_t)h->E + 63u) & ~(uint64_t)63u, N = (uint64_t)h->N;
  const uint64_t bF = align256(Ep * N * 8u * SW_F_COUNT), bI = align256(Ep * N * 4u * SW_I_COUNT), bO = align256(Ep * N), bE = 2 * align256(Ep);
  unsigned char *m = nullptr;
  hipError_t rc = hipMalloc((void **)&m, bF + bI + bO + bE);
  if (rc != hipSuccess) return rc;
  rc = hipMemset(m, 0, bF + bI + bO + bE);
  if (rc != hipSuccess) { (void)hipFree(m); return rc; }
  h->sweep_mem = m;
  h->sweep.F = (double *)m; h->sweep.I = (int *)(m + bF); h->sweep.order = m + bF + bI; h->sweep.envf = m + bF + bI + bO;
  h->sweep.urgent = h->sweep.envf + align256(Ep);
  h->sweep.Ep = (long long)Ep; h->sweep.N = h->N;
  return hipSuccess;
}

static thread_local char g_create_err[256] = "null handle";  // why the last mm_create of this thread refused (there is no handle to ask)
// The host-side entries that touch the runtime outside a stream (allocation, latch poll, drain) must address the handle's
// device, but the caller's current device is the caller's: it is put back on exit (a two-GPU process polling the env of the
// other GPU would otherwise find torch's current device changed under it).
struct DeviceGuard {
  int prev = -1;
  hipError_t rc;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    rc = hipSetDevice(device);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
static int hip_fail(MMHandle h, hipError_t e, const char *what) {
  snprintf(h->err, sizeof h->err, "%s: %s", what, hipGetErrorString(e));
  return MM_ERR_DEVICE;
}

extern "C" int32_t mm_create(const MMConfig *cfg, int32_t E, int32_t N, int32_t device, void *state,
                             uint64_t state_bytes, int64_t first_env, MMHandle *out) {
  if (!out || !state) return MM_ERR_INVALID_ARG;
  MMHandle h = (MMHandle)calloc(1, sizeof(struct MMHandle_));
  snprintf(h->err, sizeof h->err, "mm_create: E, N, the state buffer (size, 256-byte alignment) or the configuration is invalid");
  if (mm_state_layout(E, N, &h->lay) != MM_OK || state_bytes < h->lay.total_bytes ||
      ((uintptr_t)state & 255u) || check_cfg(cfg, N, h->err) != MM_OK) {
    snprintf(g_create_err, sizeof g_create_err, "%s", h->err);
    free(h);
    return MM_ERR_INVALID_ARG;
  }
  h->err[0] = 0;
  h->cfg = *cfg; h->E = E; h->N = N; h->device = device; h->state = (unsigned char *)state;
  h->first_env = first_env;
  // per-env seed plane: cfg.seed + global env index
  uint64_t *tmp = (uint64_t *)malloc((size_t)E * 8u);
  for (int64_t e = 0; e < E; e++) tmp[e] = cfg->seed + (uint64_t)(first_env + e);
  DeviceGuard dg(device);
  hipError_t rc = dg.rc;
  if (rc == hipSuccess) {
    int cus = 0;
    rc = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    h->n_simd = 4 * cus;
  }
  if (rc == hipSuccess) rc = hipMemcpy(h->state + h->lay.seed_offset, tmp, (size_t)E * 8u, hipMemcpyHostToDevice);
  free(tmp);
  if (rc == hipSuccess) rc = hipMalloc((void **)&h->dev_err, MM_LW_COUNT * sizeof(int));
  if (rc == hipSuccess) rc = hipMemset(h->dev_err, 0, MM_LW_COUNT * sizeof(int));
  if (rc == hipSuccess) rc = ensure_sweep(h);
  if (rc != hipSuccess) { if (h->dev_err) (void)hipFree(h->dev_err); free(h); return MM_ERR_DEVICE; }
  *out = h;
  return MM_OK;
}
extern "C" int32_t mm_destroy(MMHandle h) {
  if (!h) return MM_OK;
  // launches of this handle may still be in flight and they write its error latch: drain the device first
  DeviceGuard dg(h->device);
  hipError_t rc = dg.rc;
  if (rc == hipSuccess) rc = hipDeviceSynchronize();
  if (h->dev_err) (void)hipFree(h->dev_err);
  if (h->metrics_partial) (void)hipFree(h->metrics_partial);
  if (h->sweep_mem) (void)hipFree(h->sweep_mem);
  free(h);
  return rc == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}
// include/mm_abi.h: conditions the reference raises inside step(), latched by the kernels
static int poll_latch(MMHandle h, int word, MMStream stream) {
  int bits = 0;
  DeviceGuard dg(h->device);  // (a multi-GPU process may have another device current)
  hipError_t rc = dg.rc;
  if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)stream);
  if (rc == hipSuccess) rc = hipMemcpy(&bits, h->dev_err + word, sizeof bits, hipMemcpyDeviceToHost);
  if (rc == hipSuccess && bits) rc = hipMemset(h->dev_err + word, 0, sizeof(int));
  if (rc != hipSuccess) return hip_fail(h, rc, "error poll");
  if (!bits) return MM_OK;
  // every latched condition goes into the message; the return code is the most specific one
  snprintf(h->err, sizeof h->err, "%s%s%s%s",
           (bits & MM_LATCH_QP_BOUNDS) ? "Error in QP. Invalid accceleration; " : "",
           (bits & MM_LATCH_BAD_ACTION) ? "an action is outside 0..4; " : "",
           (bits & MM_LATCH_BAD_QP) ? "mm_shield_qp: G is not of the form get_G builds (cbf.py:288-304,386-403); " : "",
           (bits & MM_LATCH_INTERNAL) ? "internal: a kernel loop guard fired; " : "");
  if (bits & MM_LATCH_INTERNAL) return MM_ERR_DEVICE;
  if (bits & MM_LATCH_QP_BOUNDS) return MM_ERR_QP_BOUNDS;
  return MM_ERR_INVALID_ARG;
}
static void launch_metrics_flush(MMHandle h, hipStream_t s, int reset) {
  // per-step fold: the rows the launch just wrote; deferred fold: the whole buffer (rows nothing added to hold the identity)
  const long long waves = reset ? step_launch_waves(h) : step_launch_waves_now(h);
  hipLaunchKernelGGL(metrics_flush_kernel, dim3((unsigned)((waves + kFlushWaves - 1) / kFlushWaves)), dim3(256), 0, s,
                     h->metrics_partial, waves, h->metrics, reset);
}
extern "C" int32_t mm_flush_metrics(MMHandle h, MMStream stream) {
  if (!h) return MM_ERR_INVALID_ARG;
  if (!h->metrics_deferred || !h->metrics) return MM_OK;  // (not deferred: the caller's buffer is current after every step)
  DeviceGuard dg(h->device);  // (a multi-GPU process may have another device current: the launch belongs to the handle's)
  if (dg.rc != hipSuccess) return hip_fail(h, dg.rc, "metrics flush");
  launch_metrics_flush(h, (hipStream_t)stream, 1);
  const hipError_t rc = hipGetLastError();
  return rc == hipSuccess ? MM_OK : hip_fail(h, rc, "metrics flush");
}
extern "C" int32_t mm_defer_metrics(MMHandle h, int32_t deferred, MMStream stream) {
  if (!h) return MM_ERR_INVALID_ARG;
  if (deferred && !h->metrics) {
    snprintf(h->err, sizeof h->err, "mm_defer_metrics: no metrics buffer (mm_set_metrics_buffer first)");
    return MM_ERR_INVALID_ARG;
  }
  if ((deferred != 0) == (h->metrics_deferred != 0)) return MM_OK;
  DeviceGuard dg(h->device);
  if (dg.rc != hipSuccess) return hip_fail(h, dg.rc, "mm_defer_metrics");
  if (deferred) {  // every row to the identity (reset = 2: nothing is folded), then the step launches accumulate
    const long long waves = step_launch_waves(h);
    hipLaunchKernelGGL(metrics_flush_kernel, dim3((unsigned)((waves + kFlushWaves - 1) / kFlushWaves)), dim3(256), 0,
                       (hipStream_t)stream, h->metrics_partial, waves, (double *)nullptr, 2);
    h->metrics_deferred = 1;
  } else {
    launch_metrics_flush(h, (hipStream_t)stream, 1);
    h->metrics_deferred = 0;
  }
  const hipError_t rc = hipGetLastError();
  return rc == hipSuccess ? MM_OK : hip_fail(h, rc, "mm_defer_metrics");
}
extern "C" int32_t mm_poll_errors(MMHandle h, MMStream stream) {
  if (!h) return MM_ERR_INVALID_ARG;
  const int frc = mm_flush_metrics(h, stream);  // deferred metrics: a poll is where a rollout loop synchronises anyway
  if (frc != MM_OK) return frc;
  return poll_latch(h, MM_LW_STEP, stream);
}
extern "C" int32_t mm_set_config(MMHandle h, const MMConfig *cfg) {
  if (!h) return MM_ERR_INVALID_ARG;
  int rc = check_cfg(cfg, h->N, h->err);
  if (rc != MM_OK) return rc;
  h->cfg = *cfg;
  DeviceGuard dg(h->device);  // (the new configuration may step in the split form: its hand-off planes are allocated here)
  hipError_t hrc = dg.rc;
  if (hrc == hipSuccess) hrc = ensure_sweep(h);
  return hrc == hipSuccess ? MM_OK : hip_fail(h, hrc, "hand-off planes of the split interior-point step");
}
extern "C" int32_t mm_set_metrics_buffer(MMHandle h, double *metrics) {
  if (!h) return MM_ERR_INVALID_ARG;
  if (metrics && !h->metrics_partial) {  // per-wave partials of one step launch (allocated here, never inside step)
    const long long waves = step_launch_waves(h);
    DeviceGuard dg(h->device);
    hipError_t rc = dg.rc;
    if (rc == hipSuccess) rc = hipMalloc((void **)&h->metrics_partial, (size_t)waves * 8 * sizeof(double));
    if (rc != hipSuccess) return hip_fail(h, rc, "metrics partial buffer");
  }
  if (h->metrics_deferred && h->metrics && metrics != h->metrics) {
    // the partials were collected for the buffer that is being replaced: fold them into THAT one before the switch (there is
    // no stream argument here: the null stream, synchronised)
    DeviceGuard dg(h->device);
    hipError_t rc = dg.rc;
    if (rc == hipSuccess) { launch_metrics_flush(h, (hipStream_t)0, 1); rc = hipGetLastError(); }
    if (rc == hipSuccess) rc = hipStreamSynchronize((hipStream_t)0);
    if (rc != hipSuccess) return hip_fail(h, rc, "metrics flush before the buffer switch");
  }
  h->metrics = metrics;
  if (!metrics) h->metrics_deferred = 0;  // (no buffer, nothing to defer; what the partials held is dropped)
  return MM_OK;
}
extern "C" const char *mm_last_error(MMHandle h) { return h ? h->err : g_create_err; }
#endif  // MM_TU <= 1

static DevCfg dev_cfg(const MMHandle h) {
  const MMConfig &c = h->cfg;
  DevCfg d;
  d.env_kind = c.env_kind; d.shield = c.env_kind == MM_ENV_V1 ? c.shield : MM_SHIELD_NONE;
  d.nsub = c.simulation_frequency / c.policy_frequency; d.T = c.duration * c.policy_frequency;
  d.action_masking = c.action_masking; d.auto_reset = c.auto_reset; d.obs_f64 = c.obs_f64; d.N = h->N; d.E = h->E;
  d.debug_flags = (c.debug_flags & ~(1 << kMetricsAccBit)) | (h->metrics_deferred ? (1 << kMetricsAccBit) : 0); d.n_hdv = c.n_hdv;
  d.dt = 1.0 / c.simulation_frequency;
  d.collision_reward = c.collision_reward; d.high_speed_reward = c.high_speed_reward;
  d.headway_cost = c.headway_cost; d.headway_time = c.headway_time; d.merging_lane_cost = c.merging_lane_cost;
  d.rs_lo = c.reward_speed_lo; d.rs_hi = c.reward_speed_hi; d.eta = c.cbf_eta; d.tau = c.cbf_tau;
  d.inv_dt = 1.0 / d.dt; d.rs_span = d.rs_hi - d.rs_lo; d.rs_ispan = 1.0 / d.rs_span;
  d.rs_span2 = (d.rs_lo + (d.rs_hi - d.rs_lo) / 2) - d.rs_lo; d.rs_ispan2 = 1.0 / d.rs_span2;  // mrew, collaborating
  d.agent_reward = c.env_kind == MM_ENV_V1 ? c.agent_reward : 0;
  d.steer_vel = (c.env_kind == MM_ENV_V1 && c.lateral_control == MM_LATERAL_STEER_VEL) ? 1 : 0;
  d.err = h->dev_err + MM_LW_STEP;
  d.traffic_density = c.traffic_density; d.mixed_traffic = c.mixed_traffic; d.num_cav = c.num_cav;
  return d;
}
static DevState dev_state(const MMHandle h) {
  DevState s;
  s.F = (double *)(h->state + h->lay.f64_offset); s.B = h->state + h->lay.u8_offset;
  s.I = (int32_t *)(h->state + h->lay.env_offset); s.seeds = (uint64_t *)(h->state + h->lay.seed_offset);
  s.A = (long long)h->E * h->N; s.E = h->E; s.N = h->N;
  return s;
}
#if MM_TU <= 1
template <int G, int KIND>
static void launch_reset_t(MMHandle h, int mode, const uint8_t *mask, const uint64_t *seeds, void *obs,
                           uint8_t *avail, hipStream_t s) {
  const long long threads = (long long)h->E * G;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  hipLaunchKernelGGL((reset_kernel<G, KIND>), dim3(grid), dim3(256), 0, s, dev_cfg(h), dev_state(h), mode, mask,
                     seeds, obs, avail);
}
template <int G>
static void launch_reset_g(MMHandle h, int mode, const uint8_t *mask, const uint64_t *seeds, void *obs,
                           uint8_t *avail, hipStream_t s) {
  if (h->cfg.env_kind == MM_ENV_V1) launch_reset_t<G, MM_ENV_V1>(h, mode, mask, seeds, obs, avail, s);
  else launch_reset_t<G, MM_ENV_V0>(h, mode, mask, seeds, obs, avail, s);
}
static int launch_reset(MMHandle h, int mode, const uint8_t *mask, const uint64_t *seeds, void *obs,
                        uint8_t *avail, MMStream stream) {
  hipStream_t s = (hipStream_t)stream;
#ifdef MM_ONLY_G
  launch_reset_g<MM_ONLY_G>(h, mode, mask, seeds, obs, avail, s);
#else
  switch (group_size(h->N)) {
    case 2: launch_reset_g<2>(h, mode, mask, seeds, obs, avail, s); break;
    case 4: launch_reset_g<4>(h, mode, mask, seeds, obs, avail, s); break;
    case 8: launch_reset_g<8>(h, mode, mask, seeds, obs, avail, s); break;
    default: launch_reset_g<16>(h, mode, mask, seeds, obs, avail, s); break;
  }
#endif
  hipError_t rc = hipGetLastError();
  return rc == hipSuccess ? MM_OK : hip_fail(h, rc, "reset launch");
}

extern "C" int32_t mm_reset(MMHandle h, const uint8_t *env_mask, const uint64_t *seeds, void *obs,
                            uint8_t *avail, MMStream stream) {
  if (!h) return MM_ERR_INVALID_ARG;
  if (mm_counts_check(&h->cfg, h->N, 1, h->err, sizeof h->err)) return MM_ERR_INVALID_ARG;  // the device spawns N - n_hdv CAVs + n_hdv HDVs
  return launch_reset(h, 0, env_mask, seeds, obs, avail, stream);
}
extern "C" int32_t mm_init_from_kinematics(MMHandle h, const uint8_t *env_mask, MMStream stream) {
  if (!h) return MM_ERR_INVALID_ARG;
  return launch_reset(h, 1, env_mask, nullptr, nullptr, nullptr, stream);
}
extern "C" int32_t mm_observe(MMHandle h, void *obs, uint8_t *avail, MMStream stream) {
  if (!h) return MM_ERR_INVALID_ARG;
  return launch_reset(h, 2, nullptr, nullptr, obs, avail, stream);
}

#endif  // MM_TU <= 1

template <int G, int KIND, int SHIELD, bool MIXED, bool IPM = false>
static void launch_step_t(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  const long long threads = (long long)h->E * G;
  const unsigned grid = kPow2<G> ? (unsigned)((threads + MM_STEP_BLOCK - 1) / MM_STEP_BLOCK)
                                 : (unsigned)((h->E + 64 / G - 1) / (64 / G));  // rotation layouts: 64 / G whole groups per wave
  if (out->trace)
    hipLaunchKernelGGL((step_kernel<G, KIND, SHIELD, MIXED, IPM, true>), dim3(grid), dim3(MM_STEP_BLOCK), 0, s, dev_cfg(h), dev_state(h),
                       actions, *out, h->metrics ? h->metrics_partial : nullptr, h->sweep, 0);
  else
    hipLaunchKernelGGL((step_kernel<G, KIND, SHIELD, MIXED, IPM, false>), dim3(grid), dim3(MM_STEP_BLOCK), 0, s, dev_cfg(h), dev_state(h),
                       actions, *out, h->metrics ? h->metrics_partial : nullptr, h->sweep, 0);
}
#if MM_TU == 0 || MM_TU == 6 || MM_TU == 7
// The split interior-point step (SweepBuf, sweep_kernel): nsub + 1 phase launches with a sweep launch after each act half.
// All on the caller's stream: each launch reads what the previous one wrote.  (MIXED: the general kernels -- HDVs / steer_vel.)
template <int G, int SHIELD, bool MIXED>
static void launch_split_gs(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  const long long threads = (long long)h->E * G;
  const unsigned grid = kPow2<G> ? (unsigned)((threads + MM_STEP_BLOCK - 1) / MM_STEP_BLOCK)
                                 : (unsigned)((h->E + 64 / G - 1) / (64 / G));
  const DevCfg dc = dev_cfg(h);
  const DevState ds = dev_state(h);
  const unsigned sgrid = (unsigned)((h->E + 63) / 64);
  constexpr bool MASS = SHIELD == MM_SHIELD_MASS;
  for (int kb = 0; kb <= dc.nsub; kb++) {
    hipLaunchKernelGGL((step_kernel<G, MM_ENV_V1, SHIELD, MIXED, true, true, true>), dim3(grid), dim3(MM_STEP_BLOCK), 0, s, dc, ds, actions,
                       *out, h->metrics ? h->metrics_partial : nullptr, h->sweep, kb);
    if (kb == dc.nsub) break;
    // One sweep wave per SIMD is the design point (65 536 envs = 1 024 waves = the chip's SIMDs; the kernel is bound by the
    // latency of a lone wave).  Its LDS footprint would let a CU take five single-wave workgroups -- two of them on one SIMD
    // at half speed each while another CU holds three -- so every launch asks for dynamic LDS up to 40 KB per workgroup
    // (160 KB / 4): at most four per CU.
    auto pad = [](size_t used) { return (unsigned)(used < 40960 ? 40960 - used : 0); };
    constexpr size_t kPerVeh = 6 * 64 * sizeof(double) + 3 * 64 * sizeof(unsigned short);  // s_w + s_pk / s_meta / s_cls of sweep_kernel (+ ~3 KB: verification queue)
    if (h->N <= 4) hipLaunchKernelGGL((sweep_kernel<4, MASS, MIXED>), dim3(sgrid), dim3(64), pad(4 * kPerVeh + 5632), s, dc, ds, h->sweep, kb, out->trace);
    else if (h->N <= 8) hipLaunchKernelGGL((sweep_kernel<8, MASS, MIXED>), dim3(sgrid), dim3(64), pad(8 * kPerVeh + 5632), s, dc, ds, h->sweep, kb, out->trace);
    else if (h->N <= 11) hipLaunchKernelGGL((sweep_kernel<11, MASS, MIXED>), dim3(sgrid), dim3(64), pad(11 * kPerVeh + 1024), s, dc, ds, h->sweep, kb, out->trace);  // (38 KB: density 3 = up to 11 vehicles)
    else hipLaunchKernelGGL((sweep_kernel<12, MASS, MIXED>), dim3(sgrid), dim3(64), pad(12 * kPerVeh + 1024), s, dc, ds, h->sweep, kb, out->trace);
  }
}
template <bool MIXED>
static void launch_split_all(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  const bool mass = h->cfg.shield == MM_SHIELD_MASS;
  switch (g) {
#define MM_SPLIT_CASE(GG) case GG: if (mass) launch_split_gs<GG, MM_SHIELD_MASS, MIXED>(h, actions, out, s); else launch_split_gs<GG, MM_SHIELD_HSS, MIXED>(h, actions, out, s); break;
#ifdef MM_ONLY_G
    MM_SPLIT_CASE(MM_ONLY_G)
#else
    MM_SPLIT_CASE(2) MM_SPLIT_CASE(4) MM_SPLIT_CASE(6) MM_SPLIT_CASE(8) MM_SPLIT_CASE(12) MM_SPLIT_CASE(16)
#endif
#undef MM_SPLIT_CASE
    default: break;
  }
}
#if MM_TU == 7
void mm_launch_step_split_general(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s) { launch_split_all<true>(h, g, actions, out, s); }
#elif MM_TU == 6
void mm_launch_step_split_general(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s);
void mm_launch_step_split(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  if (needs_general(h)) mm_launch_step_split_general(h, g, actions, out, s);  // HDVs / steer_vel: the kernels that carry IDM / MOBIL
  else launch_split_all<false>(h, g, actions, out, s);
}
#else  // single-TU tuning build
static void mm_launch_step_split(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
#ifdef MM_ONLY_MIXED
  launch_split_all<MM_ONLY_MIXED>(h, g, actions, out, s);
#else
  if (needs_general(h)) launch_split_all<true>(h, g, actions, out, s);
  else launch_split_all<false>(h, g, actions, out, s);
#endif
}
#endif
#endif
template <int G, bool MIXED>
static void launch_step_m(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
#ifdef MM_ONLY_SHIELD  // tuning builds: v1 with one shield
  launch_step_t<G, MM_ENV_V1, MM_ONLY_SHIELD, MIXED>(h, actions, out, s);
  return;
#endif
  if (h->cfg.env_kind == MM_ENV_V0) { launch_step_t<G, MM_ENV_V0, MM_SHIELD_NONE, MIXED>(h, actions, out, s); return; }
  switch (h->cfg.shield) {
    case MM_SHIELD_HSS: launch_step_t<G, MM_ENV_V1, MM_SHIELD_HSS, MIXED>(h, actions, out, s); break;
    case MM_SHIELD_MASS: launch_step_t<G, MM_ENV_V1, MM_SHIELD_MASS, MIXED>(h, actions, out, s); break;
    default: launch_step_t<G, MM_ENV_V1, MM_SHIELD_NONE, MIXED>(h, actions, out, s);
  }
}
template <int G, bool MIXED>
static void launch_step_ipm_gm(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  if (h->cfg.shield == MM_SHIELD_MASS) launch_step_t<G, MM_ENV_V1, MM_SHIELD_MASS, MIXED, true>(h, actions, out, s);
  else launch_step_t<G, MM_ENV_V1, MM_SHIELD_HSS, MIXED, true>(h, actions, out, s);
}
template <int G>
static void launch_step_ipm_g(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  if (needs_general(h)) launch_step_ipm_gm<G, true>(h, actions, out, s);
  else launch_step_ipm_gm<G, false>(h, actions, out, s);  // CAV-only: the lean literal-sweep kernel around the same IPM
}
// mixed traffic (cfg.n_hdv > 0) and steer_vel run the "general" kernels that carry the IDM/MOBIL code; the MM_QP_IPM
// fidelity mode has its own (general, literal-sweep) kernels; CAV-only batches keep the leaner instantiation.
// In the split build the three families are separate translation units (Makefile).
#if MM_TU != 0
void mm_launch_step_general(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s);
void mm_launch_step_ipm(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s);
void mm_launch_step_lanes(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s);
void mm_launch_step_lanes_ipm(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s);
void mm_launch_step_split(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s);
#endif
#if MM_TU == 0 || MM_TU == 5
#if MM_TU == 0
static
#endif
void mm_launch_step_lanes_ipm(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
#ifndef MM_ONLY_G
  if (g == 6) launch_step_ipm_g<6>(h, actions, out, s);
  else launch_step_ipm_g<12>(h, actions, out, s);
#endif
}
#endif
#if MM_TU == 0 || MM_TU == 4
#if MM_TU == 0
static
#endif
void mm_launch_step_lanes(MMHandle h, int g, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
#ifndef MM_ONLY_G
  if (h->cfg.env_kind == MM_ENV_V1 && h->cfg.shield != MM_SHIELD_NONE && h->cfg.qp_solver == MM_QP_IPM) {  // interior-point mode
    mm_launch_step_lanes_ipm(h, g, actions, out, s);
  } else if (needs_general(h)) {  // HDVs / steer_vel: the kernels that carry IDM / MOBIL
    if (g == 6) launch_step_m<6, true>(h, actions, out, s);
    else launch_step_m<12, true>(h, actions, out, s);
  } else {
    if (g == 6) launch_step_m<6, false>(h, actions, out, s);
    else launch_step_m<12, false>(h, actions, out, s);
  }
#endif
}
#endif
#if MM_TU == 4 || MM_TU == 5 || MM_TU == 6 || MM_TU == 7
#elif MM_TU == 2
void mm_launch_step_general(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  switch (group_size(h->N)) {
    case 2: launch_step_m<2, true>(h, actions, out, s); break;
    case 4: launch_step_m<4, true>(h, actions, out, s); break;
    case 8: launch_step_m<8, true>(h, actions, out, s); break;
    default: launch_step_m<16, true>(h, actions, out, s); break;
  }
}
#elif MM_TU == 3
void mm_launch_step_ipm(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
  switch (group_size(h->N)) {
    case 2: launch_step_ipm_g<2>(h, actions, out, s); break;
    case 4: launch_step_ipm_g<4>(h, actions, out, s); break;
    case 8: launch_step_ipm_g<8>(h, actions, out, s); break;
    default: launch_step_ipm_g<16>(h, actions, out, s); break;
  }
}
#else
template <int G>
static void launch_step_g(MMHandle h, const int32_t *actions, const MMStepOut *out, hipStream_t s) {
#ifdef MM_ONLY_MIXED  // tuning builds
#ifdef MM_ONLY_IPM
  launch_step_t<G, MM_ENV_V1, MM_ONLY_SHIELD, MM_ONLY_MIXED, true>(h, actions, out, s);
#else
  launch_step_m<G, MM_ONLY_MIXED>(h, actions, out, s);
#endif
#else
  const bool shielded = h->cfg.env_kind == MM_ENV_V1 && h->cfg.shield != MM_SHIELD_NONE;
  if (shielded && h->cfg.qp_solver == MM_QP_IPM)
#if MM_TU == 1
    mm_launch_step_ipm(h, actions, out, s);
#else
    launch_step_ipm_g<G>(h, actions, out, s);
#endif
  else if (needs_general(h))
#if MM_TU == 1
    mm_launch_step_general(h, actions, out, s);
#else
    launch_step_m<G, true>(h, actions, out, s);
#endif
  else launch_step_m<G, false>(h, actions, out, s);
#endif
}

extern "C" int32_t mm_step(MMHandle h, const int32_t *actions, const MMStepOut *out, MMStream stream) {
  if (!h || !actions || !out) return MM_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (out->trace) {
    // NaN = "sub-step did not run"; 0xFF bytes are a NaN pattern
    hipError_t rc = hipMemsetAsync(out->trace, 0xFF, (size_t)3 * MM_T_COUNT * h->E * h->N * sizeof(double), s);
    if (rc != hipSuccess) return hip_fail(h, rc, "trace memset");
  }
  if (steps_split(h)) {  // interior-point mode, CAV-only: phase kernels + sweep kernels (SweepBuf)
    if (!h->sweep_mem) { snprintf(h->err, sizeof h->err, "mm_step: the hand-off planes of the split interior-point step are missing"); return MM_ERR_DEVICE; }
    mm_launch_step_split(h, step_group(h), actions, out, s);
  } else
#ifdef MM_ONLY_G  // tuning builds: one group size, seconds to compile
  launch_step_g<MM_ONLY_G>(h, actions, out, s);
#else
  switch (step_group(h)) {
    case 2: launch_step_g<2>(h, actions, out, s); break;
    case 4: launch_step_g<4>(h, actions, out, s); break;
    case 8: launch_step_g<8>(h, actions, out, s); break;
    case 6: case 12: mm_launch_step_lanes(h, step_group(h), actions, out, s); break;
    default: launch_step_g<16>(h, actions, out, s); break;
  }
#endif
  if (h->metrics && !h->metrics_deferred) launch_metrics_flush(h, s, 0);  // fold this launch's per-wave partials into the caller's 8 doubles (same stream: ordered)
  hipError_t rc = hipGetLastError();
  return rc == hipSuccess ? MM_OK : hip_fail(h, rc, "step launch");
}


template <int G, bool IPM>
static void launch_shield_gi(MMHandle h, const double *as, const double *aa, double *ss, double *sa, uint8_t *stt,
                             double *mg, double *hw, hipStream_t s) {
  const long long threads = (long long)h->E * G;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  const int sh = h->cfg.env_kind == MM_ENV_V1 ? h->cfg.shield : MM_SHIELD_NONE;
  DevCfg dc = dev_cfg(h);
  dc.err = h->dev_err + MM_LW_SHIELD;  // this entry reports synchronously from its own latch word
  if (sh == MM_SHIELD_MASS)
    hipLaunchKernelGGL((shield_kernel<G, MM_SHIELD_MASS, IPM>), dim3(grid), dim3(256), 0, s, dc, dev_state(h), as, aa, ss, sa, stt, mg, hw);
  else if (sh == MM_SHIELD_HSS)
    hipLaunchKernelGGL((shield_kernel<G, MM_SHIELD_HSS, IPM>), dim3(grid), dim3(256), 0, s, dc, dev_state(h), as, aa, ss, sa, stt, mg, hw);
  else
    hipLaunchKernelGGL((shield_kernel<G, MM_SHIELD_NONE, false>), dim3(grid), dim3(256), 0, s, dc, dev_state(h), as, aa, ss, sa, stt, mg, hw);
}
template <int G>
static void launch_shield_g(MMHandle h, const double *as, const double *aa, double *ss, double *sa, uint8_t *stt,
                            double *mg, double *hw, hipStream_t s) {
  if (h->cfg.qp_solver == MM_QP_IPM) launch_shield_gi<G, true>(h, as, aa, ss, sa, stt, mg, hw, s);
  else launch_shield_gi<G, false>(h, as, aa, ss, sa, stt, mg, hw, s);
}
extern "C" int32_t mm_shield_actions(MMHandle h, const double *act_steer, const double *act_acc, double *safe_steer,
                                     double *safe_acc, uint8_t *status, double *margin, double *headway, MMStream stream) {
  if (!h || !act_steer || !act_acc || !safe_steer || !safe_acc) return MM_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
#ifdef MM_ONLY_G
  launch_shield_g<MM_ONLY_G>(h, act_steer, act_acc, safe_steer, safe_acc, status, margin, headway, s);
#else
  switch (group_size(h->N)) {
    case 2: launch_shield_g<2>(h, act_steer, act_acc, safe_steer, safe_acc, status, margin, headway, s); break;
    case 4: launch_shield_g<4>(h, act_steer, act_acc, safe_steer, safe_acc, status, margin, headway, s); break;
    case 8: launch_shield_g<8>(h, act_steer, act_acc, safe_steer, safe_acc, status, margin, headway, s); break;
    default: launch_shield_g<16>(h, act_steer, act_acc, safe_steer, safe_acc, status, margin, headway, s); break;
  }
#endif
  hipError_t rc = hipGetLastError();
  if (rc != hipSuccess) return hip_fail(h, rc, "shield launch");
  // the reference call raises check_bounds' ValueError synchronously (only the IPM iterate can leave the bounds)
  return h->cfg.qp_solver == MM_QP_IPM ? poll_latch(h, MM_LW_SHIELD, stream) : MM_OK;
}

extern "C" int32_t mm_shield_qp(MMHandle h, int32_t n, const double *G, const double *hvec, const int32_t *rows,
                                int32_t solver, double *u_out, uint8_t *status, int32_t *iters, MMStream stream) {
  if (!h || n < 0 || (solver != MM_QP_EXACT && solver != MM_QP_IPM)) return MM_ERR_INVALID_ARG;
  if (n == 0) return MM_OK;
  if (!G || !hvec || !rows || !u_out) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(qp_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, G, hvec, rows, (int)solver, u_out,
                     status, iters, h->dev_err + MM_LW_QP);
  hipError_t rc = hipGetLastError();
  if (rc != hipSuccess) return hip_fail(h, rc, "qp launch");
  return poll_latch(h, MM_LW_QP, stream);
}

// diagnostics: element-wise mm_math evaluation (CPU/GPU bit-equality tests)
__global__ void math_kernel(int fn, int n, const double *__restrict__ x, const double *__restrict__ x2,
                            double *__restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r;
  switch (fn) {
    case 0: r = mmm_sin(v); break;
    case 1: r = mmm_cos(v); break;
    case 2: r = mmm_tan(v); break;
    case 3: r = mmm_atan(v); break;
    case 4: r = mmm_asin(v); break;
    case 5: r = mmm_exp(v); break;
    case 6: r = mmm_log(v); break;
    case 7: r = sqrt(v); break;
    case 9: r = div_c(v, x2[i], 1.0 / x2[i]); break;
    default: r = v / x2[i]; break;
  }
  y[i] = r;
}
extern "C" int32_t mm_math_eval(int32_t fn, int32_t n, const double *x, const double *x2, double *y,
                                MMStream stream) {
  if (fn < 0 || fn > 9 || n <= 0) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(math_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, fn, n, x, x2, y);
  return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}

// diagnostics: the step kernel's geometric device functions on stand-alone rows (include/mm_abi.h: mm_geom_eval)
__global__ void geom_kernel(int fn, int n, const double *__restrict__ in, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (fn == MM_GEOM_POSE) {
    const double x = in[i * 3 + 0], y = in[i * 3 + 1], h = in[i * 3 + 2];
    double *o = out + (long long)i * 19;
    o[0] = closest_lane(x, y, h);
    for (int l = 0; l < 6; l++) {
      o[1 + l] = next_lane(l, x, y);
      o[7 + l] = lane_reachable(l, x, y) ? 1.0 : 0.0;
      o[13 + l] = lane_after_end(l, x) ? 1.0 : 0.0;
    }
  } else if (fn == MM_GEOM_STEER) {
    const double *r = in + (long long)i * 5;
    out[i] = steering_control(r[0], r[1], r[2], r[3], (int)r[4]);
  } else if (fn == MM_GEOM_RECT) {
    const double *r = in + (long long)i * 6;
    double *o = out + (long long)i * 4;
    const double dx = r[3] - r[0], dy = r[4] - r[1];
    const bool near = !((dx * dx + dy * dy) > kU5);  // the step kernel's form of `norm > LENGTH`
    const bool t_v = near && boxes_may_touch(dx, dy, r[2], r[5], 0.9 * kVehLength / 2, 0.9 * kVehWidth / 2);
    const bool t_o = near && boxes_may_touch(dx, dy, r[2], 0.0, 0.9 * 2.0 / 2, 0.9 * 2.0 / 2);
    const bool full_v = rects_intersect(r[0], r[1], r[2], r[3], r[4], kVehLength, kVehWidth, r[5]);
    const bool full_o = rects_intersect(r[0], r[1], r[2], r[3], r[4], 2.0, 2.0, 0.0);
    o[0] = (t_v && full_v) ? 1.0 : 0.0; o[1] = (t_o && full_o) ? 1.0 : 0.0;
    o[2] = full_v ? 1.0 : 0.0; o[3] = full_o ? 1.0 : 0.0;
  } else {
    out[i] = speed_to_index(in[i]);
  }
}
extern "C" int32_t mm_geom_eval(int32_t fn, int32_t n, const double *in, double *out, MMStream stream) {
  if (fn < MM_GEOM_POSE || fn > MM_GEOM_SPEED_INDEX || n <= 0 || !in || !out) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(geom_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, fn, n, in, out);
  return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}

// marl/mappo.py:220-236 exploration_action / action for a batch (include/mm_abi.h): softmax -> inverse-CDF
// categorical sample, one thread per agent; HBM-bound by construction (4 n_a + 4 bytes per agent).
__global__ __launch_bounds__(256) void sample_kernel(const float *__restrict__ logp, long long n, int n_a, uint64_t seed,
                                                     const uint64_t *__restrict__ counter, int32_t *__restrict__ actions) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr = *counter;
  double cdf[8], acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k < n_a) { acc = acc + mmm_exp((double)logp[i * n_a + k]); cdf[k] = acc; }
  uint32_t o[4];
  philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32) ^ 0x53414D50u,
             (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const double u = u53(o[0], o[1]);
  int a = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k < n_a) a += (cdf[k] / acc <= u) ? 1 : 0;  // searchsorted(cdf / cdf[-1], u, "right")
  actions[i] = a < n_a - 1 ? a : n_a - 1;
}
__global__ void bump_kernel(uint64_t *counter) { *counter += 1; }
extern "C" int32_t mm_sample_actions(const float *logp, int64_t n, int32_t n_a, uint64_t seed, uint64_t *counter,
                                     int32_t *actions, MMStream stream) {
  if (!logp || !actions || !counter || n < 0 || n_a < 1 || n_a > 8) return MM_ERR_INVALID_ARG;
  if (n == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, logp, (long long)n, (int)n_a, seed, counter, actions);
  hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, counter);
  return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}

// marl/mappo.py:147-156,364-370: reward scaling + discounted returns of a whole rollout, one thread per agent, T values
// each, consecutive threads on consecutive addresses at every t (include/mm_abi.h: mm_discount_returns)
__global__ __launch_bounds__(256) void discount_kernel(const double *rewards, const uint8_t *__restrict__ dones,
                                                       const double *__restrict__ final_value, int T, long long n_env, int n_agent,
                                                       double gamma, double scale, double *returns) {
  const long long A = n_env * n_agent;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A) return;
  const long long e = i / n_agent;
  double running = final_value[i];
  for (int t = T - 1; t >= 0; t--) {
    double r = rewards[(long long)t * A + i];
    if (scale > 0) r = r / scale;
    if (dones[(long long)t * n_env + e]) running = 0.0;
    running = running * gamma + r;
    returns[(long long)t * A + i] = running;
  }
}
extern "C" int32_t mm_discount_returns(const double *rewards, const uint8_t *dones, const double *final_value, int32_t T,
                                       int64_t n_env, int32_t n_agent, double gamma, double reward_scale, double *returns,
                                       MMStream stream) {
  if (!rewards || !dones || !final_value || !returns || T < 0 || n_env < 0 || n_agent < 1) return MM_ERR_INVALID_ARG;
  const long long A = (long long)n_env * n_agent;
  if (A == 0 || T == 0) return MM_OK;
  hipLaunchKernelGGL(discount_kernel, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, dones,
                     final_value, (int)T, (long long)n_env, (int)n_agent, gamma, reward_scale, returns);
  return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------------
// mm_policy_act: the actor forward of the rollout loop fused with the sampling above
// (marl/single_agent/Model_common.py:5-22 ActorNetwork: n_s -> 128 -> 128 -> n_a, ReLU, log-softmax;
// marl/mappo.py:220-236).  The one GEMM-shaped piece of the path, so it runs on the matrix cores:
// f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products and sums).
//
// Orientation: every layer is computed transposed, H_out^T [feature x agent] = W [out x in] . H_in^T, one
// wave per 32 agents.  A 32x32 accumulator tile then has its agent on the lane and its features in the 16
// registers, which is exactly the B-operand layout of the next layer's MFMAs (k = feature): activations
// never leave the register file -- no LDS round trip, no transposes.  Lane (j = l & 31, h = l >> 5) holds
// features row(r, h) = (r & 3) + 8 (r >> 2) + 4 h of a tile in register r, so k-step s of a 32-feature
// chunk pairs feature row(s, 0) (lanes h = 0) with row(s, 1) (lanes h = 1); the weight fragments are
// staged once per workgroup in LDS in that order (84 KB), read back as one conflict-free float4 per four
// MFMAs.  Layer 3 (n_a <= 8 outputs) would waste 27/32 of a tile: done with 64 FMAs per output per lane
// plus one cross-half add.  Then log-softmax (fp32) and the inverse-CDF sample (fp64), written by h = 0.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kPolHidden = 128;
MM_DEV int frag_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
constexpr int kPolThreads = 512;  // 8 waves = 2 per SIMD: one wave's VALU tail (layer 3, sampling) overlaps the other's MFMAs
__global__ __launch_bounds__(kPolThreads) void policy_kernel(const float *__restrict__ obs, long long n, int n_s,
                                                        const float *__restrict__ W1, const float *__restrict__ b1,
                                                        const float *__restrict__ W2, const float *__restrict__ b2,
                                                        const float *__restrict__ W3, const float *__restrict__ b3, int n_a,
                                                        uint64_t seed, const uint64_t *__restrict__ counter,
                                                        int32_t *__restrict__ actions, float *__restrict__ logp_out) {
  __shared__ float4 sW1[4][4][64];    // [out tile][k-step / 4][lane] : 16 KB
  __shared__ float4 sW2[4][16][64];   // 64 KB
  __shared__ float sW3[8][kPolHidden];
  __shared__ float sB1[kPolHidden], sB2[kPolHidden], sB3[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  // ---- stage the weights in MFMA A-fragment order: A[i = l & 31][k = l >> 5] of step s is W[32 m + i][k(s, h)]
  for (int t = tid; t < 4 * 4 * 64; t += kPolThreads) {
    const int l = t & 63, q = (t >> 6) & 3, m = t >> 8;
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = frag_row(4 * q + u, l >> 5);
      w[u] = k < n_s ? W1[(32 * m + (l & 31)) * n_s + k] : 0.0f;
    }
    sW1[m][q][l] = make_float4(w[0], w[1], w[2], w[3]);
  }
  for (int t = tid; t < 4 * 16 * 64; t += kPolThreads) {
    const int l = t & 63, q = (t >> 6) & 15, m = t >> 10;
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int s = 4 * q + u;  // k-step 0..63: chunk s >> 4, step in chunk s & 15
      w[u] = W2[(32 * m + (l & 31)) * kPolHidden + 32 * (s >> 4) + frag_row(s & 15, l >> 5)];
    }
    sW2[m][q][l] = make_float4(w[0], w[1], w[2], w[3]);
  }
  for (int t = tid; t < 8 * kPolHidden; t += kPolThreads) sW3[t / kPolHidden][t % kPolHidden] = (t / kPolHidden) < n_a ? W3[t] : 0.0f;
  if (tid < kPolHidden) { sB1[tid] = b1[tid]; sB2[tid] = b2[tid]; }
  if (tid < 8) sB3[tid] = tid < n_a ? b3[tid] : 0.0f;
  __syncthreads();
  const uint64_t ctr = *counter;
  const long long ntiles = (n + 31) / 32;
  constexpr int kWaves = kPolThreads / 64;
  for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < ntiles; tile += (long long)gridDim.x * kWaves) {
    // the weight fragments are tile-invariant: without this the compiler hoists all 80 float4 LDS reads out of
    // the persistent loop (320 registers) and spills the activations
    asm volatile("" ::: "memory");
    const long long ag = tile * 32 + j;
    const bool live = ag < n;
    float xk[16];  // B operand of layer 1: x[agent j][k(s, h)]
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const int k = frag_row(s, h);
      xk[s] = (live && k < n_s) ? obs[ag * n_s + k] : 0.0f;
    }
    f32x16 h1[4], h2[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = sB1[32 * m + frag_row(r, h)];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 a = sW1[m][q][lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xk[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xk[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xk[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xk[4 * q + 3], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) h1[m][r] = fmaxf(acc[r], 0.0f);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = sB2[32 * m + frag_row(r, h)];
#pragma unroll
      for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 a = sW2[m][4 * c + q][lane];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, h1[c][4 * q + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, h1[c][4 * q + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, h1[c][4 * q + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, h1[c][4 * q + 3], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) h2[m][r] = fmaxf(acc[r], 0.0f);
    }
    // ---- layer 3 + log-softmax + sample
    float logit[8];
#pragma unroll
    for (int o = 0; o < 8; o++) {
      float p = 0.0f;
      if (o < n_a) {
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int r = 0; r < 16; r++) p = fmaf(h2[m][r], sW3[o][32 * m + frag_row(r, h)], p);
      }
      p = p + __shfl_xor(p, 32, 64);
      logit[o] = o < n_a ? p + sB3[o] : -INFINITY;
    }
    float mx = logit[0];
#pragma unroll
    for (int o = 1; o < 8; o++) mx = fmaxf(mx, logit[o]);
    float se = 0.0f;
#pragma unroll
    for (int o = 0; o < 8; o++) se += (o < n_a) ? expf(logit[o] - mx) : 0.0f;
    const float lse = mx + logf(se);
    if (live && h == 0) {
      double cdf[8], acc = 0;
#pragma unroll
      for (int o = 0; o < 8; o++) {
        const float lp = logit[o] - lse;
        if (o < n_a) {
          if (logp_out) logp_out[ag * n_a + o] = lp;
          acc = acc + mmm_exp((double)lp);
          cdf[o] = acc;
        }
      }
      uint32_t w4[4];
      philox4x32((uint32_t)ag, (uint32_t)((uint64_t)ag >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32) ^ 0x53414D50u,
                 (uint32_t)seed, (uint32_t)(seed >> 32), w4);
      const double u = u53(w4[0], w4[1]);
      int a = 0;
#pragma unroll
      for (int o = 0; o < 8; o++)
        if (o < n_a) a += (cdf[o] / acc <= u) ? 1 : 0;
      actions[ag] = a < n_a - 1 ? a : n_a - 1;
    }
  }
}
extern "C" int32_t mm_policy_act(const float *obs, int64_t n, int32_t n_s, const float *W1, const float *b1, const float *W2,
                                 const float *b2, const float *W3, const float *b3, int32_t hidden, int32_t n_a, uint64_t seed,
                                 uint64_t *counter, int32_t *actions, float *logp, MMStream stream) {
  if (!obs || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !counter || !actions) return MM_ERR_INVALID_ARG;
  if (n < 0 || n_s < 1 || n_s > 32 || hidden != kPolHidden || n_a < 1 || n_a > 8) return MM_ERR_INVALID_ARG;
  if (n == 0) return MM_OK;
  hipStream_t s = (hipStream_t)stream;
  const long long ntiles = (n + 31) / 32;
  constexpr int kW = kPolThreads / 64;
  const unsigned grid = (unsigned)(ntiles < kW * 256 ? (ntiles + kW - 1) / kW : 256);  // one persistent workgroup per CU
  hipLaunchKernelGGL(policy_kernel, dim3(grid), dim3(kPolThreads), 0, s, obs, (long long)n, (int)n_s, W1, b1, W2, b2, W3, b3, (int)n_a,
                     seed, counter, actions, logp);
  hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, counter);
  return hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}

#ifdef MM_STAMPS
extern "C" int32_t mm_debug_read_sweep_stamps(unsigned long long *out8, int32_t reset) {
  static unsigned long long host[4096 * 8];
  hipError_t rc = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps_s), sizeof host);
  for (int k = 0; k < 8; k++) out8[k] = 0;
  for (int w = 0; w < 4096; w++)
    for (int k = 0; k < 8; k++) out8[k] += host[w * 8 + k];
  if (rc == hipSuccess && reset) {
    memset(host, 0, sizeof host);
    rc = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_s), host, sizeof host);
  }
  return rc == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}
extern "C" int32_t mm_debug_read_stamps(unsigned long long *out16, int32_t reset) {
  static unsigned long long *host = nullptr;
  if (!host) host = (unsigned long long *)malloc(sizeof(unsigned long long) * MM_STAMP_WAVES * 16);
  hipError_t rc = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps_w), sizeof(unsigned long long) * MM_STAMP_WAVES * 16);
  for (int k = 0; k < 16; k++) out16[k] = 0;
  for (long long w = 0; w < MM_STAMP_WAVES; w++)
    for (int k = 0; k < 16; k++) out16[k] += host[w * 16 + k];
  if (rc == hipSuccess && reset) {
    memset(host, 0, sizeof(unsigned long long) * MM_STAMP_WAVES * 16);
    rc = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_w), host, sizeof(unsigned long long) * MM_STAMP_WAVES * 16);
  }
  return rc == hipSuccess ? MM_OK : MM_ERR_DEVICE;
}
#endif
#endif  // MM_TU <= 1 (the #else of the general / ipm launchers)
