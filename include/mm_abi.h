/*
 * mm_abi.h -- C ABI of the MI355X-native batched on-ramp-merge environment + CBF shield.
 *
 * This is the drop-in boundary for the ONE hot path of hkbharath/MARL-MASS: `env.reset` /
 * `env.step` of the merge envs and the HSS / MASS CBF action filter.  The reference has no FFI
 * layer (it is 100 % Python); each entry point below names the Python interface it replaces.
 * All paths are relative to the reference repository root.
 *
 * Two libraries export exactly these symbols:
 *   marl-mass_amd/csrc  -> libmm_hip.so     pointers marked DEV are device (HBM) pointers
 *   oracle/             -> libmm_oracle.so  CPU twin, same signatures, DEV pointers are host
 *                                           pointers and `stream` is ignored (test infra only)
 *
 * Memory model: the CALLER owns every buffer (PyTorch tensors in practice).  The library never
 * allocates in reset/step; the handle is a small host struct that borrows the state buffer.
 * Re-entrant per handle, no global mutable state (the reference keeps class-level globals
 * CBFType.GAMMA_B / CBFType.TAU, cbf.py:18,24 -- here they are MMConfig fields).
 *
 * State layout ("(env,agent)-major struct of arrays"): agent index i = e * N + a, a in [0, N).
 * The state buffer is  [MM_F_COUNT planes of double[E*N]] [MM_B_COUNT planes of uint8[E*N]]
 * [MM_E_COUNT planes of int32[E]] [1 plane of uint64[E] (per-env seed)], each block 256-B aligned;
 * byte offsets come from mm_state_layout().
 */
#ifndef MM_ABI_H
#define MM_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 6
#define MM_MAX_AGENTS 16 /* vehicles per env (reference draws 2..11, merge_env_v1.py:180-211) */
#define MM_N_ACTIONS 5   /* DiscreteMetaAction.ACTIONS_ALL, envs/common/action.py:141-147 */
#define MM_OBS_ROWS 5    /* KinematicObservation vehicles_count, envs/common/observation.py:132 */

/* env_kind: which registered env the handle mirrors (merge_env_v1.py:681-689) */
#define MM_ENV_V0 0 /* merge-multi-agent-v0: MDPVehicle, Kinematics obs (5 features, n_s = 25) */
#define MM_ENV_V1 1 /* merge-multi-agent-v1: MDPLCVehicle, KinematicLC obs (6 features, n_s = 30) */

/* shield: config["safety_guarantee"] after `.split("-")[1]` (safe_controller.py:241) */
#define MM_SHIELD_NONE 0
#define MM_SHIELD_HSS 1  /* "cbf-av" | "cbf-avs" | "cbf-avs_cint" | "cbf-hss"  -> safe_action_hss  */
#define MM_SHIELD_MASS 2 /* "cbf-cav" | "cbf-mass"                              -> safe_action_mass */

/* lane ids = insertion order of the road network (merge_env_v1.py:231-245, road.py:60-64) */
#define MM_LANE_AB0 0
#define MM_LANE_BC0 1
#define MM_LANE_BC1 2
#define MM_LANE_CD0 3
#define MM_LANE_JK0 4
#define MM_LANE_KB0 5

/* per-agent float64 planes */
enum {
  MM_F_X = 0,        /* Vehicle.position[0]                      kinematics.py:41  */
  MM_F_Y,            /* Vehicle.position[1]                                         */
  MM_F_HEADING,      /* Vehicle.heading                                              */
  MM_F_SPEED,        /* Vehicle.speed                                                */
  MM_F_TARGET_SPEED, /* ControlledVehicle.target_speed           controller.py:47  */
  MM_F_SAFE_STEER,   /* MDPLCVehicle.safe_action["steering"]     safe_controller.py:122 */
  MM_F_SAFE_ACC,     /* MDPLCVehicle.safe_action["acceleration"]                     */
  MM_F_G_VX,         /* MDPLCVehicle.fg_params["g"]["vx"]        safe_controller.py:167 */
  MM_F_H1_X,         /* state_hist[-1]["x"], ["vx"]              safe_controller.py:187-201 */
  MM_F_H1_VX,        /*   (the shield reads only x and vx of a history record: the heading /   */
  MM_F_H2_X,         /*    speed entries feed dpsi terms that no CBF row uses, cbf.py:206-257)  */
  MM_F_H2_VX,        /* state_hist[-2] (decentral_layer.py:126,175,202)                        */
  MM_F_STEER_ANGLE,  /* MDPLCVehicle.steering_angle, lateral_control "steer_vel" only
                        (safe_controller.py:54,124-150); stays 0 and is not touched in "steer" mode */
  MM_F_COUNT
};

/* per-agent uint8 planes */
enum {
  MM_B_LANE = 0,    /* Vehicle.lane_index as lane id            kinematics.py:44 */
  MM_B_TARGET_LANE, /* ControlledVehicle.target_lane_index      controller.py:46 */
  MM_B_SPEED_INDEX, /* MDPVehicle.speed_index                   controller.py:290 */
  MM_B_CRASHED,     /* Vehicle.crashed                          kinematics.py:48 */
  MM_B_HL_ACTION,   /* MDPLCVehicle.hl_action (0..4), 255=None  safe_controller.py:48,65 */
  MM_B_FLAGS,       /* bit0 collaborate_adj, bit1 is_lc_safe, bit2 is_collaborating (:50,60,61) */
  MM_B_HIST_LEN,    /* min(len(state_hist), 2); >=1 also means fg_params is set (:232-239) */
  MM_B_KIND,        /* 0 absent, 1 controlled CAV, 2 HDV (IDMVehicle / IDMVehicleHist, behavior.py).
                       Vehicles of an env are a prefix: CAVs first, then HDVs (creation order).
                       For an HDV the SAFE_STEER / SAFE_ACC planes persist its last IDM action and
                       G_VX its MOBIL timer (it has no safe_action / fg_params). */
  MM_B_COUNT
};
#define MM_LATERAL_STEER 0
#define MM_LATERAL_STEER_VEL 1
#define MM_FLAG_COLLABORATE_ADJ 1u
#define MM_FLAG_IS_LC_SAFE 2u  /* vehicle.is_lc_safe of the last shield call (decentral_layer.py:504,742).  The HIP step also reads it as the FIRST GUESS of
                                  this sub-step's veto (a scheduling prior: results do not depend on it, tests/test_hip_parity.py) */
#define MM_FLAG_IS_COLLABORATING 4u
#define MM_HL_NONE 255u

/* per-env int32 planes */
enum {
  MM_E_STEPS = 0, /* AbstractEnv.steps   abstract.py:67,457 */
  MM_E_TIME,      /* AbstractEnv.time    abstract.py:66,521 */
  MM_E_N_MERGE,   /* MergeEnv.n_merge    merge_env_v1.py:262,364 */
  MM_E_EPISODE,   /* episodes started on this env slot (device RNG counter) */
  MM_E_COUNT
};

typedef struct MMStateLayout {
  uint64_t f64_offset; /* double[MM_F_COUNT][E*N] */
  uint64_t u8_offset;  /* uint8 [MM_B_COUNT][E*N] */
  uint64_t env_offset; /* int32 [MM_E_COUNT][E]   */
  uint64_t seed_offset;/* uint64[E]               */
  uint64_t total_bytes;
} MMStateLayout;

/*
 * Host-side configuration.  Mirrors the env.config keys run_mappo.py:145-171 writes plus the two
 * CBF class globals (run_mappo.py:138-139).  Defaults are merge_env_v1.py:33-57.
 */
typedef struct MMConfig {
  int32_t abi_version;          /* MM_ABI_VERSION */
  int32_t env_kind;             /* MM_ENV_V0 | MM_ENV_V1 */
  int32_t shield;               /* MM_SHIELD_* ; ignored (none) for MM_ENV_V0 */
  int32_t simulation_frequency; /* 15 */
  int32_t policy_frequency;     /* 5  */
  int32_t duration;             /* 20 -> T = duration * policy_frequency = 100 */
  int32_t action_masking;       /* config["action_masking"] (abstract.py:200-207,474-481) */
  int32_t auto_reset;           /* 1: step() re-spawns finished envs with the device RNG */
  int32_t obs_f64;              /* 0: obs written as float32, 1: float64 (reference dtype) */
  int32_t debug_flags;          /* bit0: force the literal serial shield sweep (split interior-point step: the sweep kernel classifies
                                   every ego itself); bit1: step in power-of-two lane groups only (no 6- / 12-lane layout for
                                   N = 5..6 / 9..12); bit2: interior-point mode in the fused kernel at any batch size; bit3: in the split
                                   step (phase kernels + lane-per-env sweep kernel) at any batch size -- by default the split step runs
                                   for batches above one fused wave per SIMD (two for N <= 4).  Validation / A-B timing: same results either way */
  double collision_reward;      /* COLLISION_REWARD 200 */
  double high_speed_reward;     /* HIGH_SPEED_REWARD 1  */
  double headway_cost;          /* HEADWAY_COST 4       */
  double headway_time;          /* HEADWAY_TIME 1.2 (reward, merge_env_v1.py:82) */
  double merging_lane_cost;     /* MERGING_LANE_COST 4  */
  double reward_speed_lo;       /* reward_speed_range[0] 10 */
  double reward_speed_hi;       /* reward_speed_range[1] 30 */
  double cbf_eta;               /* CBFType.GAMMA_B (cbf_eta in the .ini) */
  double cbf_tau;               /* CBFType.TAU     (HEADWAY_TIME in the .ini) */
  uint64_t seed;                /* base seed of the device RNG; env e uses seed + e unless seeds given */
  int32_t n_hdv;                /* device reset: the last n_hdv of the N vehicles are IDM/MOBIL HDVs
                                   (mixed traffic, merge_env_v1.py:298-362); 0 = CAV-only */
  int32_t agent_reward;         /* config["agent_reward"] (merge_env_v1.py:439-474): 0 default, 1 srew, 2 mrew */
  int32_t lateral_control;      /* config["lateral_control"] (merge_env_v1.py:499, safe_controller.py:30-44):
                                   MM_LATERAL_STEER (1st-order, default) | MM_LATERAL_STEER_VEL (steering velocity,
                                   KP_STEER 20, STEER_TARGET_RF 0.125); v1 CAVs only, as in the reference */
  int32_t qp_solver;            /* how the shield's QP (cbf.py:128-135, cvxopt.solvers.qp) is solved:
                                   MM_QP_IPM    the interior-point iterate cvxopt's coneqp algorithm stops at (Mehrotra
                                                predictor-corrector, NT scaling, abstol 1e-7 / reltol 1e-6 / feastol 1e-7,
                                                <= 100 iterations) incl. its "unknown" status -> is_optimal = 0
                                                (include/mm_qp.h): the reference's behaviour and the default of every host
                                                entry point (_cabi.make_config, VecMergeEnv, MergeEnvCompat);
                                   MM_QP_EXACT  its exact KKT point in closed form: explicit opt-in -- the true minimiser, up
                                                to 3e-4 m/s away from the interior-point iterate (outside north_star's 1e-5) */
  int32_t traffic_density;      /* device reset / auto-reset vehicle counts (MergeEnv._num_vehicles, merge_env_v1.py:180-211):
                                   0  fixed: N - n_hdv CAVs + n_hdv HDVs in every episode (BASELINE configs);
                                   1..3  config["traffic_density"]: every episode draws num_CAV and num_HDV uniformly from
                                         {1,2,3} / {2,3,4} / CAV {4,5,6}, HDV {3,4,5} (ragged batch: unused slots are
                                         absent, MM_B_KIND = 0); N is the slot capacity and must hold the largest draw */
  int32_t mixed_traffic;        /* with traffic_density > 0: 1 = the drawn HDVs are IDM/MOBIL vehicles, 0 = they become CAVs
                                   (config["mixed_traffic"] False / traffic_type "cav": num_CAV += num_HDV, :206-209) */
  int32_t num_cav;              /* with traffic_density > 0: reset(num_CAV=...) override of the CAV draw, 0 = draw (:185,192,199) */
  int32_t reserved1;
} MMConfig;
#define MM_QP_EXACT 0
#define MM_QP_IPM 1

/*
 * Outputs of one step.  Any pointer may be NULL (that output is skipped).  DEV pointers.
 * Mirrors the (obs, reward, done, info) tuple of MergeEnv.step (merge_env_v1.py:126-166) and
 * AbstractEnv.step (abstract.py:443-510), batched over E envs.
 */
typedef struct MMStepOut {
  void *obs;                /* [E][N][5*F] float32|float64, F = 5 (v0) | 6 (v1)           */
  double *reward;           /* [E]      _reward: mean of agent rewards (:59-62)            */
  uint8_t *done;            /* [E]      _is_terminal (:168-172)                            */
  double *agents_rewards;   /* [E][N]   info["agents_rewards"]   (:139-142)                */
  double *regional_rewards; /* [E][N]   info["regional_rewards"] (:144-145)                */
  uint8_t *agents_dones;    /* [E][N]   info["agents_dones"]     (:131)                    */
  double *agents_info;      /* [E][N][3] info["agents_info"] = x, y, speed (:132-136)      */
  uint8_t *crashed;         /* [E][N]   vehicle.crashed at the end of the step             */
  double *average_speed;    /* [E]      info["average_speed"] (abstract.py:483-485)        */
  double *traffic_speed;    /* [E]      info["traffic_speed"] (:147-151)                   */
  double *min_headway;      /* [E]      info["min_headway"]   (:152, :373-386)             */
  double *merge_percent;    /* [E]      info["merge_percent"], NaN unless done (:154-163)  */
  uint8_t *action_mask;     /* [E][N][5] info["action_mask"]  (abstract.py:474-481)        */
  double *trace;            /* [3][MM_T_COUNT][E*N] per-sub-step trace (tests only) or NULL */
} MMStepOut;

/* planes of the optional per-sub-step trace (NaN where a sub-step did not run) */
enum {
  MM_T_X = 0, MM_T_Y, MM_T_HEADING, MM_T_SPEED,
  MM_T_ACT_STEER, MM_T_ACT_ACC,   /* self.action after clip_actions            */
  MM_T_SAFE_STEER, MM_T_SAFE_ACC, /* safe_action actually integrated            */
  MM_T_LANE, MM_T_TARGET_LANE, MM_T_CRASHED, MM_T_FLAGS,
  MM_T_QP_ROWS,                   /* 0 = shield did not run, else 3 | 4         */
  MM_T_QP_A,                      /* G[0][0] = g_e.vx * dt                      */
  MM_T_QP_H0, MM_T_QP_H1, MM_T_QP_H2, MM_T_QP_H3, /* h vector (H3 NaN if 3 rows) */
  MM_T_QP_D,                      /* u_bar[0] returned by the QP                */
  MM_T_LC_MARGIN,                 /* min of the four is_lc_allowed quantities (cbf.py:335-339):
                                     lane change allowed <=> margin >= 0.  Structurally ~0 (sign =
                                     rounding noise) when the adjacent CBF row is the active one */
  MM_T_STATUS,                    /* MM_ST_* bits of the in-step shield call (is_optimal / is_safe / is_invariant,
                                     cbf.py:341-357; MM_ST_QP_BOUNDS if check_bounds would have raised);
                                     NaN if the shield did not run                                          */
  MM_T_HEADWAY,                   /* vehicle.min_headway set by the shield: (x_ol - x_e - LENGTH) / vx_e
                                     (decentral_layer.py:466,700); NaN if the shield did not run             */
  MM_T_COUNT
};

typedef struct MMHandle_ *MMHandle;
typedef void *MMStream; /* hipStream_t for libmm_hip, ignored by the oracle */

/* error codes (mirrors the exceptions of SURVEY 8b "errors") */
#define MM_OK 0
#define MM_ERR_INVALID_ARG (-1)   /* ValueError: unknown safety / env type, bad sizes      */
#define MM_ERR_NOT_READY (-2)     /* NotImplementedError: road / vehicles not initialised  */
#define MM_ERR_DEVICE (-3)        /* HIP runtime failure (message in mm_last_error)        */
#define MM_ERR_QP_BOUNDS (-4)     /* ValueError of CBFType.check_bounds (cbf.py:87-96)     */

int32_t mm_abi_version(void);

/* Byte layout of the caller-owned state buffer for E envs x N agents. */
int32_t mm_state_layout(int32_t E, int32_t N, MMStateLayout *out);

/*
 * Construction = gym.make(id) + the env.config[...] writes of run_mappo.py:143-171.
 * `state` (DEV, layout above, >= total_bytes, 256-B aligned) is borrowed until mm_destroy.
 * `first_env` is the global index of local env 0 (multi-GPU sharding: RNG streams are keyed on
 * first_env + e, so a batch sharded over ranks draws the same episodes as one big batch).
 */
int32_t mm_create(const MMConfig *cfg, int32_t E, int32_t N, int32_t device, void *state,
                  uint64_t state_bytes, int64_t first_env, MMHandle *out);
int32_t mm_destroy(MMHandle h);
/* env.config[...] = ... after construction (takes effect from the next call). */
int32_t mm_set_config(MMHandle h, const MMConfig *cfg);

/*
 * reset(): AbstractEnv.reset (abstract.py:176-209) for the envs selected by env_mask (DEV
 * uint8[E], NULL = all).  Spawns N/2 vehicles on ab0 and N - N/2 on jk0 with the counter-based
 * device RNG (seeds DEV uint64[E] or NULL -> cfg.seed + first_env + e), then observes.
 * obs: DEV [E][N][5F]; avail: DEV uint8 [E][N][5] (NULL ok).
 */
int32_t mm_reset(MMHandle h, const uint8_t *env_mask, const uint64_t *seeds, void *obs,
                 uint8_t *avail, MMStream stream);

/*
 * Finish a reset whose kinematic state (x, y, heading, speed of every agent + KIND) the caller
 * wrote into the state buffer (host-side numpy-compatible spawn, fixtures): derives lane,
 * target lane, speed index / target speed exactly as Vehicle/ControlledVehicle/MDPVehicle.__init__
 * do (kinematics.py:36-53, controller.py:35-50,277-291), clears episode counters and histories.
 */
int32_t mm_init_from_kinematics(MMHandle h, const uint8_t *env_mask, MMStream stream);

/* observation_type.observe() (+ action mask) of the current state; no state change. */
int32_t mm_observe(MMHandle h, void *obs, uint8_t *avail, MMStream stream);

/*
 * step(): MergeEnv.step (merge_env_v1.py:126-166).  actions: DEV int32[E][N] in 0..4.
 * `out` is a HOST struct of DEV pointers.
 */
int32_t mm_step(MMHandle h, const int32_t *actions, const MMStepOut *out, MMStream stream);

/*
 * Stand-alone batched shield QP (unit parity of cbf.py:110-161 + cvxopt.solvers.qp):
 *   min 1/2 (d^2 + e^2 + 1e18 s^2)  s.t.  G u <= h,  u = (d, e, s),
 * for the G the reference's get_G builds (cbf.py:288-304,386-403) and no other:
 *   rows [a 0 -1], [1 0 0], [-1 0 0] and, with rows[k] == 4, a fourth [a 0 -1].
 * G: DEV double[n][4][3] row-major, h: DEV double[n][4], rows: DEV int32[n] (3 or 4).
 * solver: MM_QP_EXACT | MM_QP_IPM (see MMConfig.qp_solver).
 * u_out: DEV double[n][3]; status: DEV uint8[n] -- MM_QPS_OPTIMAL, MM_QPS_UNKNOWN (the IPM stopped at its
 * iteration cap or on a singular KKT matrix: sol["status"] == "unknown", cbf.py:140) or MM_QPS_BAD_STRUCTURE
 * (G is not of the form above: u_out is NaN and the call returns MM_ERR_INVALID_ARG);
 * iters: DEV int32[n] IPM iteration counts or NULL.  Synchronises the stream (a unit entry, not a hot one).
 */
#define MM_QPS_UNKNOWN 0u
#define MM_QPS_OPTIMAL 1u
#define MM_QPS_BAD_STRUCTURE 255u
int32_t mm_shield_qp(MMHandle h, int32_t n, const double *G, const double *hvec, const int32_t *rows,
                     int32_t solver, double *u_out, uint8_t *status, int32_t *iters, MMStream stream);

/*
 * safety_layer(safety_type, action, vehicle, dt, ...) (decentral_layer.py:767-817; call site
 * safe_controller.py:242-250) for EVERY controlled vehicle at once, each evaluated independently on
 * the CURRENT state with the nominal low-level action given (nothing is stepped, nothing is
 * written to the state: the reference's side effects on the vehicle -- is_lc_safe,
 * is_collaborating, collaborate_adj, the veto's target-lane reset -- come back as status bits).
 * act_steer / act_acc / safe_steer / safe_acc: DEV double[E][N]; status: DEV uint8[E][N]
 * (MM_ST_* bits; 0 = shield gated off as in safe_controller.py:232-239, action returned unchanged);
 * margin: DEV double[E][N] LC margin or NULL; headway: DEV double[E][N] or NULL -- the value the call's
 * vehicle.set_min_headway(...) leaves in vehicle.min_headway (decentral_layer.py:466,700 -> safe_controller.py:264-265:
 * (x_leader - x_ego - LENGTH) / vx_ego), NaN where the shield was gated off.
 */
#define MM_ST_RAN 1u           /* the CBF ran (safe_status is not None) */
#define MM_ST_IS_OPTIMAL 2u    /* status["is_optimal"] */
#define MM_ST_IS_SAFE 4u       /* status["is_safe"]      h_lon(s)  >= -1e-6  (cbf.py:341-351) */
#define MM_ST_IS_INVARIANT 8u  /* status["is_invariant"] h_lon(s') + (eta-1) h_lon(s) >= -1e-6 */
#define MM_ST_IS_LC_SAFE 16u   /* vehicle.is_lc_safe: no lane-change veto */
#define MM_ST_IS_COLLABORATING 32u
#define MM_ST_COLLABORATE_ADJ 64u
#define MM_ST_QP_BOUNDS 128u   /* CBFType.check_bounds (cbf.py:87-96) would have raised ValueError: u_safe[0] is
                                  more than 1e-3 outside [v_min, v_max]; the call returns MM_ERR_QP_BOUNDS */
int32_t mm_shield_actions(MMHandle h, const double *act_steer, const double *act_acc, double *safe_steer,
                          double *safe_acc, uint8_t *status, double *margin, double *headway, MMStream stream);

/*
 * Errors a launch cannot return synchronously.  mm_step / mm_reset only enqueue work; conditions the
 * reference raises from inside step() -- check_bounds' ValueError (cbf.py:87-96), an action outside 0..4
 * (KeyError in DiscreteMetaAction.act, action.py:194-196) -- are latched in a device word of the handle.
 * This call synchronises `stream`, returns MM_ERR_QP_BOUNDS / MM_ERR_INVALID_ARG (message in mm_last_error)
 * if one was latched since the last poll, else MM_OK, and clears the latch.  mm_shield_qp and
 * mm_shield_actions poll it themselves.
 */
int32_t mm_poll_errors(MMHandle h, MMStream stream);

/*
 * Rollout metric accumulator (the only cross-GPU quantity): adds this step's
 * {sum reward, crashed episodes, sum average_speed, sum traffic_speed, env-steps, sum merge %,
 *  finished episodes} into metrics[0..6] and min-reduces min_headway into metrics[7].
 * metrics: DEV double[8], caller-initialised (zeros, +inf).  Optional.
 */
int32_t mm_set_metrics_buffer(MMHandle h, double *metrics);

/*
 * Deferred folding of the rollout metrics.  By default every mm_step is followed by a small launch that folds the
 * step's per-wave partial sums into the caller's 8 doubles, so they are current after every step.  A rollout loop
 * (marl/mappo.py:102-158) reads them once, at its end: with deferred != 0 the partials accumulate across steps in the
 * handle's own device buffer and reach the caller's 8 doubles only in mm_flush_metrics -- which mm_poll_errors also
 * calls -- i.e. one fold per rollout instead of one per step (the fold is 4 of the 75 microseconds of a step at 8 192
 * envs).  Needs a metrics buffer (mm_set_metrics_buffer first); turning deferral off flushes.  Both calls only
 * enqueue work on `stream` (the stream of the mm_step calls).
 */
int32_t mm_defer_metrics(MMHandle h, int32_t deferred, MMStream stream);
int32_t mm_flush_metrics(MMHandle h, MMStream stream);

const char *mm_last_error(MMHandle h);

/*
 * Policy head of the rollout loop: exploration_action / action (marl/mappo.py:220-236) for a batch,
 *   softmax_action = exp(actor(state));  action = np.random.choice(n_a, p=softmax_action)
 * np.random.choice is inverse-CDF sampling: cdf = cumsum(p); cdf /= cdf[-1]; cdf.searchsorted(u, "right").
 * One thread per agent does exactly that in fp64 from the actor's float32 log-probabilities, with
 * u = 53-bit uniform from Philox4x32-10 keyed by (seed; *counter, agent index) -- no RNG tensor, no
 * softmax / cumsum round trips through HBM (24 B per agent in and out).
 * logp: DEV float[n][n_a] (n_a <= 8); counter: DEV uint64 (read by the launch, then incremented by one
 * on the stream, so hipGraph replays draw fresh numbers); actions: DEV int32[n].  Stateless.
 */
int32_t mm_sample_actions(const float *logp, int64_t n, int32_t n_a, uint64_t seed, uint64_t *counter,
                          int32_t *actions, MMStream stream);

/*
 * The same with the actor network in front (marl/single_agent/Model_common.py:5-22: n_s -> hidden -> hidden
 * -> n_a, ReLU, log-softmax; weights in torch nn.Linear layout [out][in], float32): one launch from the
 * observation rows mm_step wrote to the next actions.  HIP build: f32-input MFMA, activations kept in
 * registers (marl-mass_amd/csrc/mm_kernels.hip policy_kernel).  hidden must be 128 (the reference's only
 * value), n_s <= 32, n_a <= 8.  obs: DEV float[n][n_s]; logp: optional DEV float[n][n_a] (the log-softmax
 * the sample was drawn from); counter / actions as for mm_sample_actions.
 */
int32_t mm_policy_act(const float *obs, int64_t n, int32_t n_s, const float *W1, const float *b1, const float *W2,
                      const float *b2, const float *W3, const float *b3, int32_t hidden, int32_t n_a, uint64_t seed,
                      uint64_t *counter, int32_t *actions, float *logp, MMStream stream);

/*
 * End of a rollout (marl/mappo.py:147-156 + _discount_reward :364-370), for every agent of a batch at once:
 *   rewards = rewards / reward_scale            (reward_scale > 0; :152-153)
 *   running = final_value;  for t = T-1 .. 0:  if done[t][e]: running = 0;  running = running * gamma + rewards[t];
 *   returns[t] = running
 * -- the reference discounts one episode list per agent; a `done` at step t cuts the chain, because what follows in
 * the batch's rollout belongs to the next episode of that env slot.  One thread per agent walks its T values once
 * (2 x 8 B per value of traffic) instead of three element-wise launches per step of the rollout.  fp64, one rounding
 * per operation in the order written.  rewards: DEV double[T][n_env][n_agent]; dones: DEV uint8[T][n_env];
 * final_value: DEV double[n_env][n_agent]; returns: DEV double[T][n_env][n_agent] (may alias rewards).  Stateless.
 */
int32_t mm_discount_returns(const double *rewards, const uint8_t *dones, const double *final_value, int32_t T,
                            int64_t n_env, int32_t n_agent, double gamma, double reward_scale, double *returns,
                            MMStream stream);

/*
 * Diagnostics: evaluate one elementary function of include/mm_math.h element-wise
 * (fn: 0 sin, 1 cos, 2 tan, 3 atan, 4 asin, 5 exp, 6 log, 7 sqrt, 8 x/y with y = x2[i],
 * 9 x/y through the HIP path's constant-divisor form div_c (the oracle uses plain division)).
 * x, x2 (may be NULL unless fn == 8), y: DEV double[n].  Used to prove CPU/GPU bit equality.
 */
int32_t mm_math_eval(int32_t fn, int32_t n, const double *x, const double *x2, double *y, MMStream stream);

/*
 * Diagnostics: the geometric building blocks of the step kernel on stand-alone inputs, one row per item, so that the
 * reference's unit tables (tests/golden/units.npz: utils.py:90-121 rotated_rectangles_intersect, road.py:51-109
 * get_closest_lane_index / next_lane, lane.py:61-95, controller.py:146-187,327-337) run against the DEVICE functions
 * themselves and not only through trajectories.  in / out: DEV double rows (integers and flags as doubles).
 *   MM_GEOM_POSE   in [n][3] x, y, heading          -> out [n][19] closest lane, next_lane of lane 0..5, is_reachable_from
 *                                                       of lane 0..5, after_end of lane 0..5
 *   MM_GEOM_STEER  in [n][5] x, y, heading, speed, target lane -> out [n][1] steering_control
 *   MM_GEOM_RECT   in [n][6] x1, y1, h1, x2, y2, h2 -> out [n][4]: [0] vehicle 1 collides with a vehicle at pose 2 AS THE
 *                  STEP KERNEL DECIDES IT (norm <= LENGTH pre-check of kinematics.py:205, then the kernel's own exact
 *                  early-out boxes_may_touch, then the 9-point test), [1] the same against an Obstacle (2 x 2, heading 0) at
 *                  (x2, y2), [2] / [3] the 9-point test alone (= rotated_rectangles_intersect) for the two cases
 *   MM_GEOM_SPEED_INDEX in [n][1] speed -> out [n][1] MDPVehicle.speed_to_index
 */
#define MM_GEOM_POSE 0
#define MM_GEOM_STEER 1
#define MM_GEOM_RECT 2
#define MM_GEOM_SPEED_INDEX 3
int32_t mm_geom_eval(int32_t fn, int32_t n, const double *in, double *out, MMStream stream);

#ifdef __cplusplus
}
#endif
#endif /* MM_ABI_H */
