"""ctypes binding of the C ABI in include/mm_abi.h (the drop-in boundary).

Pure plumbing: structures, prototypes and error mapping.  The same binding serves the HIP
library (device pointers) and -- from tests only -- the CPU oracle (host pointers); which
shared object is loaded is decided by the caller, never silently.
"""
import ctypes as C
import os

MM_ABI_VERSION = 6
MM_MAX_AGENTS = 16
ENV_V0, ENV_V1 = 0, 1
SHIELD_NONE, SHIELD_HSS, SHIELD_MASS = 0, 1, 2

# plane indices (keep in sync with include/mm_abi.h; checked by tests/test_abi.py)
F_PLANES = ["X", "Y", "HEADING", "SPEED", "TARGET_SPEED", "SAFE_STEER", "SAFE_ACC", "G_VX",
            "H1_X", "H1_VX", "H2_X", "H2_VX", "STEER_ANGLE"]
B_PLANES = ["LANE", "TARGET_LANE", "SPEED_INDEX", "CRASHED", "HL_ACTION", "FLAGS", "HIST_LEN", "KIND"]
E_PLANES = ["STEPS", "TIME", "N_MERGE", "EPISODE"]
T_PLANES = ["X", "Y", "HEADING", "SPEED", "ACT_STEER", "ACT_ACC", "SAFE_STEER", "SAFE_ACC", "LANE",
            "TARGET_LANE", "CRASHED", "FLAGS", "QP_ROWS", "QP_A", "QP_H0", "QP_H1", "QP_H2", "QP_H3",
            "QP_D", "LC_MARGIN", "STATUS", "HEADWAY"]
F = {n: i for i, n in enumerate(F_PLANES)}
B = {n: i for i, n in enumerate(B_PLANES)}
EP = {n: i for i, n in enumerate(E_PLANES)}
T = {n: i for i, n in enumerate(T_PLANES)}

FLAG_COLLABORATE_ADJ, FLAG_IS_LC_SAFE, FLAG_IS_COLLABORATING = 1, 2, 4
HL_NONE = 255

ST_RAN, ST_IS_OPTIMAL, ST_IS_SAFE, ST_IS_INVARIANT, ST_IS_LC_SAFE, ST_IS_COLLABORATING, ST_COLLABORATE_ADJ = 1, 2, 4, 8, 16, 32, 64
ST_QP_BOUNDS = 128
QP_EXACT, QP_IPM = 0, 1
QPS_UNKNOWN, QPS_OPTIMAL, QPS_BAD_STRUCTURE = 0, 1, 255
MM_OK, MM_ERR_INVALID_ARG, MM_ERR_NOT_READY, MM_ERR_DEVICE, MM_ERR_QP_BOUNDS = 0, -1, -2, -3, -4

# (from, to, id) lane tuples of the reference <-> lane ids (merge_env_v1.py:231-245)
LANE_INDEX = [("a", "b", 0), ("b", "c", 0), ("b", "c", 1), ("c", "d", 0), ("j", "k", 0), ("k", "b", 0)]
LANE_ID = {l: i for i, l in enumerate(LANE_INDEX)}


class MMStateLayout(C.Structure):
    _fields_ = [("f64_offset", C.c_uint64), ("u8_offset", C.c_uint64), ("env_offset", C.c_uint64),
                ("seed_offset", C.c_uint64), ("total_bytes", C.c_uint64)]


class MMConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("env_kind", C.c_int32), ("shield", C.c_int32),
                ("simulation_frequency", C.c_int32), ("policy_frequency", C.c_int32),
                ("duration", C.c_int32), ("action_masking", C.c_int32), ("auto_reset", C.c_int32),
                ("obs_f64", C.c_int32), ("debug_flags", C.c_int32),
                ("collision_reward", C.c_double), ("high_speed_reward", C.c_double),
                ("headway_cost", C.c_double), ("headway_time", C.c_double),
                ("merging_lane_cost", C.c_double), ("reward_speed_lo", C.c_double),
                ("reward_speed_hi", C.c_double), ("cbf_eta", C.c_double), ("cbf_tau", C.c_double),
                ("seed", C.c_uint64), ("n_hdv", C.c_int32), ("agent_reward", C.c_int32),
                ("lateral_control", C.c_int32), ("qp_solver", C.c_int32),
                ("traffic_density", C.c_int32), ("mixed_traffic", C.c_int32), ("num_cav", C.c_int32), ("reserved1", C.c_int32)]


GEOM_POSE, GEOM_STEER, GEOM_RECT, GEOM_SPEED_INDEX = 0, 1, 2, 3  # mm_geom_eval functions (include/mm_abi.h)


class MMStepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "obs", "reward", "done", "agents_rewards", "regional_rewards", "agents_dones", "agents_info",
        "crashed", "average_speed", "traffic_speed", "min_headway", "merge_percent", "action_mask",
        "trace")]


SUPERVISED = ("priority", "dmc")


def check_supervisor(value):
    """"priority" (the reference's DEFAULT, merge_env_v1.py:47) and "dmc" make AbstractEnv.step rewrite the joint
    action through safety_supervisor / safety_layer_dmc (abstract.py:460-464) -- competing baselines outside
    this path.  Stepping them as if unsupervised would silently change behaviour and safety,
    so step() raises: set "none" or "cbf-*" explicitly, as run_mappo.py does from the .ini.  (reset() is
    unaffected, as in the reference: the supervisor only acts inside step.)"""
    if value in SUPERVISED:
        raise NotImplementedError(
            "safety_guarantee=%r (the reference's action-replacement supervisors, abstract.py:460-464) is not part of "
            "this path: set env.config['safety_guarantee'] to 'none' or 'cbf-*' before stepping" % (value,))


def shield_from_safety_guarantee(value):
    """config["safety_guarantee"] -> VEHICLE-level shield id, as safe_controller.py:229-241 +
    decentral_layer.py:767-817 dispatch it ("priority" / "dmc" have none: their supervisor sits in
    AbstractEnv.step, see check_supervisor).  Unknown "cbf-*" types raise ValueError like safety_layer does."""
    if value in (None, "none") or value in SUPERVISED or "cbf-" not in value:
        return SHIELD_NONE
    kind = value.split("-")[1]
    if kind in ("hss", "av", "avs", "avs_cint"):
        return SHIELD_HSS
    if kind in ("mass", "cav"):
        return SHIELD_MASS
    raise ValueError("Undefined safety_type:{0}".format(kind))


def default_env_config(env_id):
    """Default config of the registered envs (merge_env_v1.py:33-57, :389-437, abstract.py:106-133)."""
    cfg = {
        "simulation_frequency": 15, "policy_frequency": 5, "duration": 20,
        "reward_speed_range": [10, 30], "COLLISION_REWARD": 200, "HIGH_SPEED_REWARD": 1,
        "HEADWAY_COST": 4, "HEADWAY_TIME": 1.2, "MERGING_LANE_COST": 4, "traffic_density": 1,
        "safety_guarantee": "priority", "seed": 0, "action_masking": True, "mixed_traffic": True,
        "controlled_vehicles": 4,
    }
    if env_id == "merge-multi-agent-v1":
        cfg.update({"action_masking": False, "lateral_control": "steer", "traffic_type": "cav",
                    "agent_reward": "default"})
    elif env_id != "merge-multi-agent-v0":
        raise ValueError("unsupported env id %r (hot path covers merge-multi-agent-v0 / -v1)" % env_id)
    return cfg


def qp_solver_id(name):
    """'ipm' (the iterate cvxopt's coneqp stops at -- the reference's own behaviour, the default of every entry point) |
    'exact' (the closed-form KKT point: explicit opt-in, outside north_star's 1e-5 of the interior-point iterate)."""
    try:
        return {"exact": QP_EXACT, "ipm": QP_IPM, QP_EXACT: QP_EXACT, QP_IPM: QP_IPM}[name]
    except KeyError:
        raise ValueError("qp_solver must be 'exact' or 'ipm', got %r" % (name,))


def make_config(env_id, config, cbf_eta=0.0, cbf_tau=None, auto_reset=False, obs_f64=False, seed=0, debug_flags=0,
                n_hdv=0, qp_solver="ipm", draw_counts=False, num_cav=0):
    """env.config dict (+ CBFType.GAMMA_B / CBFType.TAU, run_mappo.py:138-139) -> MMConfig."""
    c = MMConfig()
    c.abi_version = MM_ABI_VERSION
    c.env_kind = ENV_V1 if env_id == "merge-multi-agent-v1" else ENV_V0
    c.shield = shield_from_safety_guarantee(config.get("safety_guarantee")) if c.env_kind == ENV_V1 else SHIELD_NONE
    lat = config.get("lateral_control", "steer") if c.env_kind == ENV_V1 else "steer"
    if lat not in ("steer", "steer_vel"):  # safe_controller.py:181-184
        raise AttributeError("Lateral control: {0} is not supported".format(lat))
    c.lateral_control = 1 if lat == "steer_vel" else 0
    c.simulation_frequency = int(config["simulation_frequency"])
    c.policy_frequency = int(config["policy_frequency"])
    c.duration = int(config["duration"])
    c.action_masking = int(bool(config.get("action_masking", False)))
    c.auto_reset = int(bool(auto_reset))
    c.obs_f64 = int(bool(obs_f64))
    c.debug_flags = int(debug_flags)
    c.collision_reward = float(config["COLLISION_REWARD"])
    c.high_speed_reward = float(config["HIGH_SPEED_REWARD"])
    c.headway_cost = float(config["HEADWAY_COST"])
    c.headway_time = float(config["HEADWAY_TIME"])
    c.merging_lane_cost = float(config["MERGING_LANE_COST"])
    c.reward_speed_lo = float(config["reward_speed_range"][0])
    c.reward_speed_hi = float(config["reward_speed_range"][1])
    c.cbf_eta = float(cbf_eta)
    c.cbf_tau = float(config["HEADWAY_TIME"] if cbf_tau is None else cbf_tau)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.n_hdv = int(n_hdv)
    ar = config.get("agent_reward", "default") if c.env_kind == ENV_V1 else "default"
    c.agent_reward = {"srew": 1, "mrew": 2}.get(ar, 0)  # anything else falls back to the default reward (:446)
    c.qp_solver = qp_solver_id(qp_solver)
    # draw_counts: the device reset draws the vehicle counts per episode like MergeEnv._num_vehicles does
    # (config["traffic_density"] 1..3, config["mixed_traffic"]); otherwise every episode has N - n_hdv CAVs + n_hdv HDVs
    c.traffic_density = int(config.get("traffic_density", 1)) if draw_counts else 0
    if draw_counts and c.traffic_density not in (1, 2, 3):
        raise ValueError("traffic_density must be 1, 2 or 3 when the vehicle counts are drawn")
    # MergeEnv._num_vehicles folds the HDVs into the CAVs only when mixed_traffic `is not None and not ...`
    # (merge_env_v1.py:206-209): None -- run_mappo.py's fallback -- means mixed
    mixed = config.get("mixed_traffic", True)
    mixed = 1 if mixed is None else int(bool(mixed))
    if c.env_kind == ENV_V1:  # MergeEnvLCMARL._num_vehicles: traffic_type decides (merge_env_v1.py:476-495)
        tt = config.get("traffic_type", "cav")
        if tt in ("mixed", "cav"):
            mixed = int(tt == "mixed")
        elif tt == "av":  # one CAV, every other drawn vehicle an HDV (:485-489); the total does not depend on mixed_traffic
            mixed = 2
        elif tt == "hdv" and draw_counts:  # no controlled vehicle at all: nothing to step or observe on this path
            raise NotImplementedError("traffic_type='hdv' (zero controlled vehicles) is not supported by the device-side count draw")
    c.mixed_traffic = mixed
    c.num_cav = int(num_cav)
    return c


class CLib(object):
    """One loaded implementation of include/mm_abi.h."""

    SYMBOLS = ["mm_abi_version", "mm_state_layout", "mm_create", "mm_destroy", "mm_set_config",
               "mm_reset", "mm_init_from_kinematics", "mm_observe", "mm_step", "mm_shield_qp",
               "mm_set_metrics_buffer", "mm_last_error", "mm_math_eval", "mm_shield_actions", "mm_sample_actions",
               "mm_policy_act", "mm_poll_errors", "mm_geom_eval", "mm_defer_metrics", "mm_flush_metrics", "mm_discount_returns"]

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s is missing: build it first (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        self.lib = lib = C.CDLL(path)
        vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
        lib.mm_abi_version.restype = i32
        lib.mm_state_layout.argtypes = [i32, i32, C.POINTER(MMStateLayout)]
        lib.mm_create.argtypes = [C.POINTER(MMConfig), i32, i32, i32, vp, u64, i64, C.POINTER(vp)]
        lib.mm_destroy.argtypes = [vp]
        lib.mm_set_config.argtypes = [vp, C.POINTER(MMConfig)]
        lib.mm_reset.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.mm_init_from_kinematics.argtypes = [vp, vp, vp]
        lib.mm_observe.argtypes = [vp, vp, vp, vp]
        lib.mm_step.argtypes = [vp, vp, C.POINTER(MMStepOut), vp]
        lib.mm_shield_qp.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]
        lib.mm_poll_errors.argtypes = [vp, vp]
        lib.mm_set_metrics_buffer.argtypes = [vp, vp]
        lib.mm_defer_metrics.argtypes = [vp, i32, vp]
        lib.mm_flush_metrics.argtypes = [vp, vp]
        lib.mm_discount_returns.argtypes = [vp, vp, vp, i32, i64, i32, C.c_double, C.c_double, vp, vp]
        lib.mm_last_error.argtypes = [vp]
        lib.mm_math_eval.argtypes = [i32, i32, vp, vp, vp, vp]
        lib.mm_geom_eval.argtypes = [i32, i32, vp, vp, vp]
        lib.mm_shield_actions.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.mm_sample_actions.argtypes = [vp, i64, i32, u64, vp, vp, vp]
        lib.mm_policy_act.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, i32, i32, u64, vp, vp, vp, vp]
        lib.mm_last_error.restype = C.c_char_p
        for s in self.SYMBOLS:
            if s not in ("mm_abi_version", "mm_last_error"):
                getattr(lib, s).restype = i32
        if lib.mm_abi_version() != MM_ABI_VERSION:
            raise RuntimeError("ABI version mismatch in %s" % path)

    def check(self, rc, handle=None):
        """Map C status codes to the exception types the reference raises (SURVEY 8b 'errors')."""
        if rc == MM_OK:
            return
        msg = self.lib.mm_last_error(handle if handle else None).decode()  # (no handle: why the last mm_create refused)
        if rc in (MM_ERR_INVALID_ARG, MM_ERR_QP_BOUNDS):
            raise ValueError(msg or "invalid argument")
        if rc == MM_ERR_NOT_READY:
            raise NotImplementedError(msg or "The road and vehicle must be initialized in the environment implementation")
        raise RuntimeError("mm error %d: %s" % (rc, msg))

    def state_layout(self, E, N):
        lay = MMStateLayout()
        self.check(self.lib.mm_state_layout(E, N, C.byref(lay)))
        return lay
