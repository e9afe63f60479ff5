"""Drop-in adapter with the reference's gym-0.14 style Env surface.

`MergeEnvCompat` is a batch-of-1 view of the batched engine exposing exactly what
marl/mappo.py and run_mappo.py touch: `reset`, `step`, `config[...]`, `seed`, `n_s`, `n_a`, `T`,
`controlled_vehicles`, `road.vehicles`, `is_crashed()`, `close()`, `ACTIONS_ALL`, `unwrapped`.
The episode spawn replays the reference's *global* numpy RNG call sequence (abstract.py:182-187,
merge_env_v1.py:180-211,265-364) so that a MAPPO loop seeded the same way sees the same episodes
and the same downstream `np.random` stream.

`CBFType`, `cbf_factory`, `safety_layer` are the same-named shims of the shield interface
(cbf.py:433-440, decentral_layer.py:767-817): they pack the QP and call the batched C entry.
"""
import random

import numpy as np
import torch

from . import _cabi as abi
from .vec_env import VecMergeEnv

MAX_VEHICLES = 12  # 6 + 6 spawn slots (merge_env_v1.py:284-285)


class CBFType(object):
    """Class-level knobs the reference sets from the .ini (cbf.py:18,24; run_mappo.py:138-139)."""
    GAMMA_B = 0
    TAU = 0.5
    ADJ_BUFFER = 2.0134
    ACCELERATION_RANGE = (-12.5, 6)
    # not in the reference: how `solvers.qp` (cbf.py:134) is answered; read at reset like GAMMA_B / TAU.
    #   "ipm" (default): the iterate cvxopt's interior-point algorithm stops at, with its status -- the reference's own
    #                    behaviour, incl. is_optimal = False on "unknown" and check_bounds (include/mm_qp.h);
    #   "exact":         the closed-form KKT point of the same QP: its true minimiser, ~20x cheaper on the device, but up to
    #                    3e-4 m/s away from what cvxopt returns -- an explicit opt-in.
    QP_SOLVER = "ipm"


class _VehicleView(object):
    """Read-only window on one agent of the device state (what callers read off a vehicle)."""
    LENGTH, WIDTH = 5.0, 2.0

    def __init__(self, env, index):
        self._env, self.id = env, index

    def _f(self, name):
        return float(self._env._b.f64[abi.F[name], 0, self.id])

    def _b(self, name):
        return int(self._env._b.u8[abi.B[name], 0, self.id])

    @property
    def position(self):
        return np.array([self._f("X"), self._f("Y")])

    heading = property(lambda s: s._f("HEADING"))
    steering_angle = property(lambda s: s._f("STEER_ANGLE"))  # lateral_control "steer_vel" (safe_controller.py:54)
    speed = property(lambda s: s._f("SPEED"))
    target_speed = property(lambda s: s._f("TARGET_SPEED"))
    crashed = property(lambda s: bool(s._b("CRASHED")))
    speed_index = property(lambda s: s._b("SPEED_INDEX"))
    lane_index = property(lambda s: abi.LANE_INDEX[s._b("LANE")])
    target_lane_index = property(lambda s: abi.LANE_INDEX[s._b("TARGET_LANE")])
    collaborate_adj = property(lambda s: bool(s._b("FLAGS") & abi.FLAG_COLLABORATE_ADJ))
    is_lc_safe = property(lambda s: bool(s._b("FLAGS") & abi.FLAG_IS_LC_SAFE))
    is_collaborating = property(lambda s: bool(s._b("FLAGS") & abi.FLAG_IS_COLLABORATING))

    @property
    def velocity(self):
        return self.speed * np.array([np.cos(self.heading), np.sin(self.heading)])

    @property
    def safe_action(self):
        return {"steering": self._f("SAFE_STEER"), "acceleration": self._f("SAFE_ACC")}


class _Obstacle(object):
    LENGTH = WIDTH = 2.0
    position = np.array([420.0, 4.0])
    heading = 0.0
    speed = 0.0


class _Road(object):
    def __init__(self):
        self.vehicles = []
        self.objects = [_Obstacle()]


class MergeEnvCompat(object):
    """gym.make("merge-multi-agent-v0" | "-v1") replacement; see module docstring."""

    n_a = 5
    metadata = {"render.modes": []}

    def __init__(self, env_id="merge-multi-agent-v0", config=None, backend_factory=None, device="cuda:0",
                 store_profile=False):
        """`store_profile=True` keeps the per-sub-step control profile of every vehicle
        (`state_hist` / `action_hist`, safe_controller.py:187-227, behavior.py:505-521) -- what
        MAPPOControlEval.evaluation (marl/mappo.py:420-438) hands to log_profiles; see control_profile()."""
        self.env_id = env_id
        self.store_profile = bool(store_profile)
        self.config = abi.default_env_config(env_id)
        if config:
            self.config.update(config)
        self.n_s = 30 if env_id == "merge-multi-agent-v1" else 25
        self.seed = self.config["seed"]
        self.ACTIONS_ALL = {"LANE_LEFT": 0, "IDLE": 1, "LANE_RIGHT": 2, "FASTER": 3, "SLOWER": 4}
        self.ends = [220, 100, 100, 1000]
        self._factory = backend_factory or (lambda **kw: VecMergeEnv(device=device, **kw))
        self._b = None
        self.road = None
        self.controlled_vehicles = []
        self.time = self.steps = 0
        self.done = False
        self.n_merge = 0
        self.T = int(self.config["duration"] * self.config["policy_frequency"])
        try:
            self.reset()  # abstract.py:86 (the constructor already resets)
        except NotImplementedError:
            # a configuration the constructor's reset cannot serve yet (e.g. traffic_type "hdv": no controlled vehicle):
            # stay un-initialised until the caller has written env.config[...] (run_mappo.py:145-171)
            # and calls reset(); step() before that raises like abstract.py:454-455.
            self.road, self.controlled_vehicles = None, []

    unwrapped = property(lambda self: self)

    # -- spawn: merge_env_v1.py:180-211 (+ :476-495 for v1) -------------------------------
    def _num_vehicles(self, num_CAV=0):
        cfg = self.config
        if self.env_id == "merge-multi-agent-v1":
            tt = cfg.get("traffic_type", "cav")
            if tt == "mixed":
                cfg["mixed_traffic"] = True
            elif tt == "cav":
                cfg["mixed_traffic"] = False
            elif tt == "av":  # merge_env_v1.py:485-489: one CAV, everything else HDV
                num_CAV, num_HDV = self._draw_counts(num_CAV)
                return 1, num_CAV + num_HDV - 1
            elif tt == "hdv":
                raise NotImplementedError("traffic_type='hdv' (no controlled vehicle, MergeEnvLCHDV) is out of scope")
        return self._draw_counts(num_CAV)

    def _draw_counts(self, num_CAV=0):
        cfg = self.config
        num_HDV = 0
        lo_hi = {1: ((1, 4), (1, 4)), 2: ((2, 5), (2, 5)), 3: ((4, 7), (3, 6))}.get(cfg["traffic_density"])
        if lo_hi:
            if num_CAV == 0:
                num_CAV = np.random.choice(np.arange(*lo_hi[0]), 1)[0]
            num_HDV = np.random.choice(np.arange(*lo_hi[1]), 1)[0]
        if cfg.get("mixed_traffic") is not None and not cfg["mixed_traffic"]:
            num_CAV, num_HDV = num_CAV + num_HDV, 0
        return int(num_CAV), int(num_HDV)

    def _make_vehicles(self, num_CAV, num_HDV):
        """Spawn draw of merge_env_v1.py:265-364 on the GLOBAL numpy RNG.  What has to match the reference is the
        sequence of RNG calls (a MAPPO loop seeded the same way must see the same episodes and the same
        downstream stream): per vehicle class (CAVs, then HDVs) one `choice(2)` iff the class has exactly one
        vehicle, then `choice(free slots, k, replace=False)` for the main road and for the ramp; finally one
        `rand(n)` for the speeds and one for the position noise, consumed in creation order (CAVs main, CAVs
        ramp, HDVs main, HDVs ramp).  Returns x, y, speed in that order and the number of ramp CAVs."""
        n_all = num_CAV + num_HDV
        if n_all > MAX_VEHICLES:
            raise ValueError("at most %d vehicles (6 + 6 spawn slots)" % MAX_VEHICLES)
        free = {"main": [10 + 50 * k for k in range(6)], "ramp": [5 + 50 * k for k in range(6)]}
        lane_y = {"main": 0.0, "ramp": 6.5 + 4}
        groups = []  # (road, slots) in creation order
        n_ramp_cav = 0
        for cls, count in (("cav", num_CAV), ("hdv", num_HDV)):
            on_main = count // 2 if count != 1 else np.random.choice(2)
            split = {"main": on_main, "ramp": count - on_main}
            if cls == "cav":
                n_ramp_cav = split["ramp"]
            for road in ("main", "ramp"):
                picked = [int(p) for p in np.random.choice(free[road], split[road], replace=False)]
                free[road] = [p for p in free[road] if p not in picked]
                groups.append((road, picked))
        speed = np.random.rand(n_all) * 2 + 25
        noise = np.random.rand(n_all) * 8 - 4
        slot = np.array([p for _, picked in groups for p in picked], dtype=np.float64)
        y = np.array([lane_y[road] for road, picked in groups for _ in picked], dtype=np.float64)
        return slot + noise, y, speed.astype(np.float64), int(n_ramp_cav)

    def _backend(self):
        if self._b is None:
            self._b = self._factory(E=1, N=MAX_VEHICLES, env_id=self.env_id, config=self.config,
                                    cbf_eta=CBFType.GAMMA_B, cbf_tau=CBFType.TAU, obs_f64=True,
                                    trace=self.store_profile, qp_solver=CBFType.QP_SOLVER)
        return self._b

    # -- Env API ------------------------------------------------------------------------------
    def reset(self, is_training=True, testing_seeds=0, num_CAV=0):
        """abstract.py:176-209 -> (obs[n_agents, n_s] float64, available_actions[n_agents, 5])."""
        if is_training:
            np.random.seed(self.seed)
            random.seed(self.seed)
        else:
            np.random.seed(testing_seeds)
            random.seed(testing_seeds)
        self.time = self.steps = 0
        self.seed += 1
        self.done = False
        self.vehicle_speed, self.vehicle_pos = [], []
        n_cav, n_hdv = self._num_vehicles(num_CAV=num_CAV)
        x, y, v, n_merge = self._make_vehicles(n_cav, n_hdv)
        self.n_merge = n_merge
        self.T = int(self.config["duration"] * self.config["policy_frequency"])
        b = self._backend()
        b.configure(self.config, cbf_eta=CBFType.GAMMA_B, cbf_tau=CBFType.TAU, n_hdv=n_hdv, qp_solver=CBFType.QP_SOLVER)
        n_all = len(x)
        self._n = n = n_cav
        pad = lambda a, fill: np.concatenate([a, np.full(MAX_VEHICLES - n_all, fill)])  # noqa: E731
        kind = pad(np.array([1] * n_cav + [2] * n_hdv), 0)
        obs, avail = b.set_kinematics(pad(x, np.nan)[None], pad(y, 0.0)[None], np.zeros((1, MAX_VEHICLES)),
                                      pad(v, 0.0)[None], n_merge=np.array([n_merge]), kind=kind[None])
        self.road = _Road()
        self.controlled_vehicles = [_VehicleView(self, i) for i in range(n_cav)]
        self.road.vehicles = self.controlled_vehicles + [_VehicleView(self, i) for i in range(n_cav, n_all)]
        for veh in self.road.vehicles:  # MDPLCVehicle.__init__ :46-56 / IDMVehicleHist.__init__
            veh.state_hist, veh.action_hist, veh.t_step = [], [], 0.0
            veh.min_headway = 180.0 / 40.0  # PERCEPTION_DIST / MAX_SPEED (:56)
            veh._sang = 0.0
        return (obs[0, :n].cpu().numpy().astype(np.float64).reshape(n, -1),
                avail[0, :n].cpu().numpy().astype(np.int64))

    @property
    def vehicle(self):
        return self.controlled_vehicles[0] if self.controlled_vehicles else None

    def step(self, action):
        """merge_env_v1.py:126-166 / abstract.py:443-510 -> (obs, reward, done, info)."""
        if self.road is None or self.vehicle is None:
            raise NotImplementedError("The road and vehicle must be initialized in the environment implementation")
        n = self._n
        action = tuple(int(a) for a in action)
        assert len(action) == n
        b = self._b
        act = torch.ones(1, MAX_VEHICLES, dtype=torch.int32)
        act[0, :n] = torch.tensor(action, dtype=torch.int32)
        obs, reward, done, out = b.step(act.to(b.device))
        b.poll_errors()  # check_bounds' ValueError (cbf.py:87-96) / an action outside 0..4, raised like the reference does
        if self.store_profile:
            self._log_profiles(b)
        self.steps += 1
        self.time = int(b.env_i32[abi.EP["TIME"], 0])
        o = {k: v[0].cpu().numpy() for k, v in out.items()}
        terminal = bool(o["done"])
        speeds = o["agents_info"][:n, 2]
        self.vehicle_speed.append([float(s) for s in speeds])
        self.vehicle_pos.append([float(p) for p in o["agents_info"][:n, 0]])
        info = {
            "speed": float(speeds[0]), "crashed": bool(o["crashed"][0]), "action": action,
            "new_action": action, "action_mask": o["action_mask"][:n].astype(np.int64),
            "average_speed": float(o["average_speed"]),
            "vehicle_speed": np.array(self.vehicle_speed), "vehicle_position": np.array(self.vehicle_pos),
            "agents_dones": tuple(bool(d) for d in o["agents_dones"][:n]),
            "agents_info": [[float(a) for a in row] for row in o["agents_info"][:n]],
            "agents_rewards": tuple(float(r) for r in o["agents_rewards"][:n]),
            "regional_rewards": tuple(float(r) for r in o["regional_rewards"][:n]),
            "traffic_speed": float(o["traffic_speed"]), "min_headway": float(o["min_headway"]),
        }
        if terminal:
            info["merge_percent"] = float(o["merge_percent"])
        return (obs[0, :n].cpu().numpy().astype(np.float64).reshape(n, -1), float(o["reward"]), terminal, info)

    def _log_profiles(self, b):
        """log_step of every vehicle for the sub-steps the last step ran (from the device trace)."""
        T = abi.T
        tr = b.trace[:, :, 0].cpu().numpy()  # [3, planes, MAX_VEHICLES]
        dt = 1.0 / self.config["simulation_frequency"]
        is_lc = self.env_id == "merge-multi-agent-v1"
        steer_vel = is_lc and self.config.get("lateral_control", "steer") == "steer_vel"
        hl_name = {v: k for k, v in self.ACTIONS_ALL.items()}
        for k in range(tr.shape[0]):
            if np.isnan(tr[k, T["X"], 0]):
                continue  # early exit on a terminal state (abstract.py:529-531)
            for j, veh in enumerate(self.road.vehicles):
                g = lambda name: float(tr[k, T[name], j])  # noqa: E731
                cav = j < self._n
                veh.t_step += dt
                h, speed = g("HEADING"), g("SPEED")
                state_rec = {"presence": 1, "x": g("X"), "y": g("Y"), "vx": speed * np.cos(h), "vy": speed * np.sin(h),
                             "heading": h, "cos_h": np.cos(h), "sin_h": np.sin(h), "cos_d": 0.0, "sin_d": 0.0, "speed": speed}
                if cav and is_lc:
                    ran = not np.isnan(g("STATUS"))
                    safe = {"steering": g("SAFE_STEER"), "acceleration": g("SAFE_ACC")}
                    if steer_vel:
                        veh._sang += safe["steering"] * dt  # safe_controller.py:139
                        state_rec["steering_angle"] = veh._sang
                    else:
                        state_rec["steering_angle"] = g("ACT_STEER")
                    if ran:
                        bits = int(g("STATUS"))
                        state_rec["safe_status"] = {"is_optimal": bool(bits & abi.ST_IS_OPTIMAL), "is_safe": bool(bits & abi.ST_IS_SAFE),
                                                    "is_invariant": bool(bits & abi.ST_IS_INVARIANT)}
                        veh.min_headway = g("HEADWAY")
                    state_rec["t_step"] = veh.t_step
                    state_rec["headway"] = veh.min_headway
                    action_rec = dict(safe)
                    action_rec["ull_acceleration"], action_rec["ull_steering"] = g("ACT_ACC"), g("ACT_STEER")
                    hl = veh._b("HL_ACTION")
                    action_rec["lc_action"] = self.ACTIONS_ALL[hl_name[hl]] if hl != abi.HL_NONE else None
                    if ran:
                        action_rec["safe_diff"] = {"acceleration": safe["acceleration"] - g("ACT_ACC"),
                                                   "steering": safe["steering"] - g("ACT_STEER")}
                    action_rec["t_step"] = veh.t_step
                elif is_lc:  # IDMVehicleHist.log_step behavior.py:509-521
                    state_rec["steering_angle"] = g("ACT_STEER")
                    state_rec["t_step"] = veh.t_step
                    action_rec = {"steering": g("ACT_STEER"), "acceleration": g("ACT_ACC"), "t_step": veh.t_step}
                else:
                    continue  # v0 vehicles (MDPVehicle / IDMVehicle) keep no profile
                veh.state_hist.append(state_rec)
                veh.action_hist.append(action_rec)

    def control_profile(self):
        """ext_info["control_profile"] of MAPPOControlEval.evaluation (marl/mappo.py:424-437)."""
        cp = {}
        for j, veh in enumerate(self.road.vehicles):
            cp[("av" if j < self._n else "hdv") + str(veh.id)] = {"state_hist": veh.state_hist, "action_hist": veh.action_hist}
        return cp

    def is_crashed(self):
        return any(v.crashed for v in self.controlled_vehicles)

    def render(self, mode="human"):
        raise NotImplementedError("pygame rendering is outside the hot path (SURVEY 2 row 11)")

    def close(self):
        self.done = True


def make(env_id, **kw):
    """gym.make counterpart for the two env ids on the hot path (merge_env_v1.py:681-689)."""
    return MergeEnvCompat(env_id, **kw)


# ---------------------------------------------------------------------------------------------
# shield interface shims (cbf.py / decentral_layer.py)
# ---------------------------------------------------------------------------------------------
class _CBF(object):
    """CBF_AV / CBF_CAV counterpart (cbf.py:196-430): row assembly on the host, solve on the device."""
    STATE_SPACE = ["x", "heading"]
    ACCELERATION_RANGE = (-12.5, 6)

    def __init__(self, action_size, action_bound, vehicle_size, vehicle_lane=0, is_ma=False, solver=None):
        self.action_size, self.action_bound, self.vehicle_size = action_size, action_bound, vehicle_size
        self.is_ma_dynamics, self.constrain_adj = is_ma, False
        self.safe_dists = [0, 0, 0]
        self.is_optimal = self.is_safe = self.is_invariant = None
        self.p_lon = np.array([-1, 0, 1, 0, 0, 0, 0, 0.0])
        self.p_lona = np.array([-1, 0, 0, 0, 1, 0, 0, 0.0])
        self.p_lonr = np.array([1, 0, 0, 0, 0, 0, -1, 0.0])
        self._solver = solver

    def define_pq(self, x=None):
        self.q_lon = -self.vehicle_size[0] - self.safe_dists[0]
        self.q_lona = -self.vehicle_size[0] - self.safe_dists[1]
        self.q_lonr = -self.vehicle_size[0] - self.safe_dists[2]
        if self.is_ma_dynamics and self.constrain_adj:
            self.q_lona = -self.vehicle_size[0] - self.safe_dists[1] - CBFType.ADJ_BUFFER

    def get_G(self, g):
        G = np.array([np.append(-np.dot(self.p_lon, g[:, :2]), -1.0), [1, 0, 0], [-1, 0, 0]], dtype=float)
        if self.is_ma_dynamics and self.constrain_adj:
            G = np.vstack([G, np.append(-np.dot(self.p_lona, g[:, :2]), -1.0)])
        return G

    def _row(self, p, q, f, g, x, u, eta):
        return np.dot(p, f) + (eta - 1) * np.dot(p, x) + eta * q + np.dot(p, np.squeeze(np.dot(g, u)))

    def get_h(self, f, g, x, u_ll, eta=None):
        eta = CBFType.GAMMA_B if eta is None else eta
        h = [self._row(self.p_lon, self.q_lon, f, g, x, u_ll, eta),
             self.action_bound[0][1] - u_ll[0], -self.action_bound[0][0] + u_ll[0]]
        if self.is_ma_dynamics and self.constrain_adj:
            h.append(self._row(self.p_lona, self.q_lona, f, g, x, u_ll, eta))
        return np.array(h, dtype=float)

    def control_barrier(self, u_ll, f, g, x, dt=0):
        """cbf.py:110-161: returns u_safe[2]; the QP runs in the batched device solver."""
        u_ll = np.squeeze(np.asarray(u_ll, dtype=float))
        self.define_pq(x)
        G, h = self.get_G(g), self.get_h(f, g, x, u_ll)
        Gp, hp = np.zeros((1, 4, 3)), np.zeros((1, 4))
        Gp[0, :G.shape[0]], hp[0, :h.shape[0]] = G, h
        solver = self._solver or _default_solver()
        u_bar, status = solver.shield_qp(Gp, hp, np.array([G.shape[0]], dtype=np.int32), solver=CBFType.QP_SOLVER)
        u_bar = u_bar[0].cpu().numpy()
        u_safe = u_ll[:2] + u_bar[:2]
        if u_safe[0] - 0.001 > self.action_bound[0][1] or u_safe[0] + 0.001 < self.action_bound[0][0]:
            raise ValueError("Error in QP. Invalid accceleration: {0}".format(u_safe[0]))
        self.is_optimal = int(status[0]) == abi.QPS_OPTIMAL  # sol["status"] != "unknown" (cbf.py:140)
        return np.array(u_safe)

    def get_status(self):
        return {"is_optimal": float(bool(self.is_optimal))}


_SOLVER = None


def _default_solver():
    """Device context for stand-alone `control_barrier` calls of CBFs built without `solver=` (the reference's
    cbf_factory has no such argument): one minimal handle, created on first use, only ever used for mm_shield_qp."""
    global _SOLVER
    if _SOLVER is None:
        _SOLVER = VecMergeEnv(E=1, N=2, config={"safety_guarantee": "none"})
    return _SOLVER


def cbf_factory(cbf_type, solver=None, **kwargs):
    """cbf.py:433-440."""
    if cbf_type in ("hss", "av", "avs", "avs_cint"):
        return _CBF(is_ma=False, solver=solver, **kwargs)
    elif cbf_type in ("mass", "cav"):
        return _CBF(is_ma=True, solver=solver, **kwargs)
    raise ValueError("Undefined cbf_type:{0}".format(cbf_type))


def safety_layer(safety_type, action, vehicle, dt, safe_dist="theadway", **kwargs):
    """decentral_layer.py:767-817: (safe_action, safe_diff, status) for one vehicle of a MergeEnvCompat.

    `vehicle` is one of `env.controlled_vehicles`; `action` its nominal {"steering", "acceleration"}.
    The evaluation runs on the device (`mm_shield_actions`) against the env's current state.  The reference call
    also WRITES to the vehicle (decentral_layer.py:497-506 / :725-752): `is_collaborating`, `is_lc_safe`,
    `collaborate_adj` (MASS) and, when the lane change is vetoed, `target_lane_index = lane_index`; this shim does
    the same on the vehicle's state planes (so the next `env.step` sees them exactly as after the reference call)
    and also returns the three flags in `status` next to is_optimal / is_safe / is_invariant, and sets
    `vehicle.min_headway` like the call's `vehicle.set_min_headway(...)` (decentral_layer.py:466,700).  What it does not
    reproduce: the in-place history edit of an HDV twin, which the in-step shield applies itself."""
    if safety_type not in ("hss", "av", "avs", "avs_cint", "mass", "cav"):
        raise ValueError("Undefined safety_type:{0}".format(safety_type))
    if safe_dist != "theadway":
        raise ValueError("safe_dist type {} not supported".format(safe_dist))
    env = vehicle._env
    want = abi.SHIELD_MASS if safety_type in ("mass", "cav") else abi.SHIELD_HSS
    b = env._b
    if b._cfg.shield != want:
        raise ValueError("env is configured with safety_guarantee=%r; safety_type %r needs the matching env.config"
                         % (env.config.get("safety_guarantee"), safety_type))
    if abs(dt * b._cfg.simulation_frequency - 1.0) > 1e-12:
        raise ValueError("dt must be 1 / simulation_frequency")
    steer = torch.zeros(1, MAX_VEHICLES, dtype=torch.float64)
    acc = torch.zeros(1, MAX_VEHICLES, dtype=torch.float64)
    steer[0, vehicle.id], acc[0, vehicle.id] = float(action["steering"]), float(action["acceleration"])
    s_s, s_a, st, _, hw = b.shield_actions(steer, acc)
    bits = int(st[0, vehicle.id])
    safe_action = {"acceleration": float(s_a[0, vehicle.id]), "steering": float(s_s[0, vehicle.id])}
    if not bits & abi.ST_RAN:  # gated off like get_safe_action (safe_controller.py:229-239)
        return dict(action), None, None
    safe_diff = {"acceleration": safe_action["acceleration"] - action["acceleration"],
                 "steering": safe_action["steering"] - action["steering"]}
    status = {"is_optimal": float(bool(bits & abi.ST_IS_OPTIMAL)), "is_safe": float(bool(bits & abi.ST_IS_SAFE)),
              "is_invariant": float(bool(bits & abi.ST_IS_INVARIANT)), "is_lc_safe": bool(bits & abi.ST_IS_LC_SAFE),
              "is_collaborating": bool(bits & abi.ST_IS_COLLABORATING), "collaborate_adj": bool(bits & abi.ST_COLLABORATE_ADJ)}
    # the call's side effects on the vehicle (see the docstring)
    vehicle.min_headway = float(hw[0, vehicle.id])  # set_min_headway (safe_controller.py:264-265)
    flags = (abi.FLAG_IS_LC_SAFE if status["is_lc_safe"] else 0) | (abi.FLAG_IS_COLLABORATING if status["is_collaborating"] else 0)
    if want == abi.SHIELD_MASS:
        flags |= abi.FLAG_COLLABORATE_ADJ if status["collaborate_adj"] else 0
    else:  # HSS leaves vehicle.collaborate_adj as it was
        flags |= int(b.u8[abi.B["FLAGS"], 0, vehicle.id]) & abi.FLAG_COLLABORATE_ADJ
    b.u8[abi.B["FLAGS"], 0, vehicle.id] = flags
    if not status["is_lc_safe"]:  # "Avoiding lane change"
        b.u8[abi.B["TARGET_LANE"], 0, vehicle.id] = b.u8[abi.B["LANE"], 0, vehicle.id]
    return safe_action, safe_diff, status
