#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE (runs in the build container only).

Imports hkbharath/MARL-MASS from /root/reference under the stand-ins in tools/refshim/
(gym / pygame stubs, numpy+pandas compat, and a cvxopt stand-in -- cvxopt 1.2.7 is not installable
here -- whose `solvers.qp` is either the exact KKT closed form or, for the ipm_* tapes, a restatement
of cvxopt's own interior-point algorithm; see tools/refshim/cvxopt/__init__.py and coneqp.py).
Everything else that runs is the reference's own arithmetic.  The outputs are DATA only
(tests/golden/*.npz); no reference source is copied.

    python tools/gen_golden.py            # regenerate every fixture

Fixture families
  units.npz        pure-function tables (lane argmin, steering control, rect intersection, ...)
  reset.npz        reference reset() results for given seeds / vehicle counts
  ep_*.npz         episode tapes: initial state, action tape, per-sub-step vehicle state,
                   per-step obs / rewards / dones / info, and (shield runs) every QP (G, h, x)
  ipm_*.npz        the same with `solvers.qp` answered by the coneqp restatement (MM_QP_IPM mode)
  am_*.npz         action_masking = True (abstract.py:219-240,474-481, incl. the row aliasing)
  ed_*.npz         placed edge cases: x < 0 terminal, |v| > 40 clamp, a <= 0 and lo > hi QPs
"""
import os
import sys
import json

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference"]

import numpy as np  # noqa: E402
import _refcompat  # noqa: E402,F401
import gym  # noqa: E402
import highway_env  # noqa: E402,F401
import cvxopt  # noqa: E402
from highway_env import utils as hutils  # noqa: E402
from highway_env.road.road import Road  # noqa: E402
from highway_env.vehicle.safety.cbf import CBFType  # noqa: E402
from highway_env.vehicle.controller import MDPVehicle  # noqa: E402
from highway_env.vehicle.safe_controller import MDPLCVehicle  # noqa: E402

OUT = os.environ.get("MM_GOLDEN_OUT", os.path.join(REPO, "tests", "golden"))
LANE_ID = {("a", "b", 0): 0, ("b", "c", 0): 1, ("b", "c", 1): 2,
           ("c", "d", 0): 3, ("j", "k", 0): 4, ("k", "b", 0): 5}
LANE_IX = {v: k for k, v in LANE_ID.items()}
HL = {None: -1, "LANE_LEFT": 0, "IDLE": 1, "LANE_RIGHT": 2, "FASTER": 3, "SLOWER": 4}

_SUBSTEP_LOG = None
ONLY = []  # tape-name prefixes to (re)generate; empty = everything
_orig_road_step = Road.step


def _veh_snapshot(v):
    """One row of per-vehicle state as the reference holds it after a sub-step."""
    is_lc = isinstance(v, MDPLCVehicle)
    sa = getattr(v, "safe_action", None) or v.action
    fg = getattr(v, "fg_params", None)
    f = [v.position[0], v.position[1], v.heading, v.speed, v.target_speed,
         float(v.action["steering"]), float(v.action["acceleration"]),
         float(sa["steering"]), float(sa["acceleration"]),
         float(fg["g"]["vx"]) if fg else np.nan,
         float(getattr(v, "timer", np.nan))]
    i = [LANE_ID[tuple(v.lane_index)], LANE_ID[tuple(v.target_lane_index)], int(getattr(v, "speed_index", 0)),
         int(bool(v.crashed)), HL[getattr(v, "hl_action", None)],
         int(bool(getattr(v, "collaborate_adj", False))) if is_lc else 0,
         int(bool(getattr(v, "is_lc_safe", False))) if is_lc else 0,
         int(bool(getattr(v, "is_collaborating", False))) if is_lc else 0,
         1 if isinstance(v, MDPVehicle) else 2]  # kind: 1 controlled CAV, 2 HDV (IDMVehicle / IDMVehicleHist)
    return f, i


def _profile_tail(v):
    """What log_step (safe_controller.py:187-227) just appended to the vehicle's control profile:
    [shield ran, is_optimal, is_safe, is_invariant, headway]."""
    if not isinstance(v, MDPLCVehicle) or not v.state_hist:
        return [0, 0, 0, 0, np.nan]
    rec = v.state_hist[-1]
    st = rec.get("safe_status")
    if st is None:
        return [0, 0, 0, 0, float(rec["headway"])]
    return [1, float(bool(st["is_optimal"])), float(bool(st["is_safe"])), float(bool(st["is_invariant"])), float(rec["headway"])]


def _road_step_logged(self, dt):
    _orig_road_step(self, dt)
    if _SUBSTEP_LOG is not None:
        rows = [_veh_snapshot(v) for v in self.vehicles]
        _SUBSTEP_LOG.append((np.array([r[0] for r in rows], dtype=np.float64),
                             np.array([r[1] for r in rows], dtype=np.int32),
                             np.array([float(getattr(v, "steering_angle", 0.0)) for v in self.vehicles]),
                             np.array([_profile_tail(v) for v in self.vehicles])))


Road.step = _road_step_logged


def make_env(env_id, shield, n_cav, headway_time, eta, n_hdv=0, agent_reward="default", lateral_control="steer",
             action_masking=False):
    """Mirror of how run_mappo.py:137-171 configures an env (values from the cited .ini files)."""
    CBFType.GAMMA_B = eta
    CBFType.TAU = headway_time
    env = gym.make(env_id)
    env.config["simulation_frequency"] = 15
    env.config["duration"] = 20
    env.config["policy_frequency"] = 5
    env.config["COLLISION_REWARD"] = 200
    env.config["HIGH_SPEED_REWARD"] = 1
    env.config["HEADWAY_COST"] = 4
    env.config["HEADWAY_TIME"] = headway_time
    env.config["MERGING_LANE_COST"] = 4
    env.config["traffic_density"] = 1
    env.config["action_masking"] = bool(action_masking)
    env.config["safety_guarantee"] = shield
    env.config["lateral_control"] = lateral_control
    env.config["mixed_traffic"] = n_hdv > 0
    env.config["traffic_type"] = "mixed" if n_hdv > 0 else "cav"
    env.config["agent_reward"] = agent_reward
    # BASELINE configs fix the vehicle count (N CAVs, 0 HDVs); the reference draws it at random.
    env._num_vehicles = lambda num_CAV=0: (n_cav, n_hdv)
    return env


def _place_vehicles(env, placement):
    """Replace the spawned vehicles by a scripted scenario: [(x, y, speed[, heading]), ...] in creation order
    (the way test/cbf/cbf_test_env.py:126-160,240-268 builds its fixed scenarios)."""
    road = env.road
    road.vehicles, env.controlled_vehicles = [], []
    for k, row in enumerate(placement):
        x, y, speed = row[:3]
        hdv = len(row) > 4 and row[4] == "h"  # an IDM / MOBIL vehicle (merge_env_v1.py:344-362); HDVs follow the CAVs
        make = env._make_hdv_vehicle if hdv else env._make_ego_vehicle
        v = make(road=road, position=np.array([x, y], dtype=float), speed=speed, veh_id=k)
        if len(row) > 3 and row[3] is not None:  # Vehicle.__init__(heading=...) (kinematics.py:36-53): lane index from the heading too
            v.heading = float(row[3])
            v.lane_index = road.network.get_closest_lane_index(v.position, v.heading)
            v.lane = road.network.get_lane(v.lane_index)
            v.target_lane_index = v.lane_index
        if not hdv:
            env.controlled_vehicles.append(v)
        road.vehicles.append(v)
    env._record_vehicle_count(n_merge=sum(1 for row in placement if row[1] > 5 and not (len(row) > 4 and row[4] == "h")))
    env.define_spaces()
    obs = env.observation_type.observe()
    return np.asarray(obs).reshape((len(obs), -1))


def _probe_safety_layer(env, shield, t, rng):
    """Stand-alone reference calls safety_layer(safety_type, action, vehicle, dt, ...) (decentral_layer.py:767-817)
    for every controlled vehicle at the start of step t, each on its own deep copy of the env (the call
    has side effects).  Returns (t, actions[n,2], safe[n,2], status[n,6])."""
    import copy
    from highway_env.vehicle.safety.decentral_layer import safety_layer as ref_safety_layer
    saved = cvxopt.solvers.log
    cvxopt.solvers.log = None
    acts, safes, stats = [], [], []
    for k in range(len(env.controlled_vehicles)):
        ec = copy.deepcopy(env)
        veh = ec.controlled_vehicles[k]
        action = {"steering": float(veh.action["steering"]) + float(rng.uniform(-0.02, 0.02)),
                  "acceleration": float(rng.uniform(-6, 6))}
        sa, sd, st = ref_safety_layer(safety_type=shield.split("-")[1], action=dict(action), vehicle=veh, dt=1 / 15,
                                      road=ec.road, perception_dist=180, safe_dist="theadway")
        acts.append([action["steering"], action["acceleration"]])
        safes.append([sa["steering"], sa["acceleration"]])
        stats.append([st["is_optimal"], st["is_safe"], st["is_invariant"], float(veh.is_lc_safe),
                      float(veh.is_collaborating), float(veh.collaborate_adj)])
    cvxopt.solvers.log = saved
    return t, acts, safes, stats


def run_episode(name, env_id, shield, n_cav, seed, tape_seed, headway_time, eta, p=None,
                max_steps=100, scripted=None, placement=None, n_hdv=0, agent_reward="default", probe_shield=False,
                lateral_control="steer", qp="exact", action_masking=False):
    global _SUBSTEP_LOG
    if ONLY and not any(name.startswith(o) for o in ONLY):
        return None
    cvxopt.solvers.mode = qp  # "exact" | "coneqp" (tools/refshim/cvxopt)
    env = make_env(env_id, shield, n_cav, headway_time, eta, n_hdv=n_hdv, agent_reward=agent_reward,
                   lateral_control=lateral_control, action_masking=action_masking)
    obs0, mask0 = env.reset(is_training=False, testing_seeds=seed)
    if placement is not None:
        obs0 = _place_vehicles(env, placement)
    n = len(env.controlled_vehicles)
    assert n == n_cav and len(env.road.vehicles) == n_cav + n_hdv
    init_f, init_i = zip(*[_veh_snapshot(v) for v in env.road.vehicles])
    rng = np.random.RandomState(tape_seed)
    p = p or [0.1, 0.6, 0.1, 0.1, 0.1]
    cvxopt.solvers.log = []
    sub_f, sub_i, sub_sa, sub_pf, sub_count = [], [], [], [], []
    rec = {k: [] for k in ("actions", "obs", "reward", "done", "agents_rewards", "regional_rewards",
                           "agents_dones", "average_speed", "traffic_speed", "min_headway",
                           "merge_percent", "action_mask", "qp_count")}
    done = False
    t = 0
    probes = []
    while not done and t < max_steps:
        if probe_shield and t >= 2 and t % 6 == 3:
            probes.append(_probe_safety_layer(env, shield, t, rng))
        if scripted is not None:
            a = tuple(int(x) for x in scripted[min(t, len(scripted) - 1)])
        else:
            a = tuple(int(x) for x in rng.choice(5, size=n, p=p))
        _SUBSTEP_LOG = []
        nqp0 = len(cvxopt.solvers.log)
        obs, reward, done, info = env.step(a)
        sub_count.append(len(_SUBSTEP_LOG))
        for f, i, sa, pf in _SUBSTEP_LOG:
            sub_f.append(f)
            sub_i.append(i)
            sub_sa.append(sa)
            sub_pf.append(pf)
        _SUBSTEP_LOG = None
        rec["actions"].append(a)
        rec["obs"].append(np.asarray(obs, dtype=np.float64))
        rec["reward"].append(float(reward))
        rec["done"].append(bool(done))
        rec["agents_rewards"].append(np.array(info["agents_rewards"], dtype=np.float64))
        rec["regional_rewards"].append(np.array(info["regional_rewards"], dtype=np.float64))
        rec["agents_dones"].append(np.array(info["agents_dones"], dtype=bool))
        rec["average_speed"].append(float(info["average_speed"]))
        rec["traffic_speed"].append(float(info["traffic_speed"]))
        rec["min_headway"].append(float(info["min_headway"]))
        rec["merge_percent"].append(float(info.get("merge_percent", np.nan)))
        rec["action_mask"].append(np.asarray(info["action_mask"], dtype=np.uint8))
        rec["qp_count"].append(len(cvxopt.solvers.log) - nqp0)
        t += 1
    qps = cvxopt.solvers.log
    cvxopt.solvers.log = None
    qp_rows = np.array([q_[0].shape[0] for q_ in qps], dtype=np.int32)
    qp_G = np.zeros((len(qps), 4, 3))
    qp_h = np.full((len(qps), 4), np.nan)
    qp_x = np.zeros((len(qps), 3))        # what solvers.qp returned (the mode in meta["qp_solver"])
    qp_x_alt = np.zeros((len(qps), 3))    # the other mode's answer to the same (G, h)
    qp_status = np.zeros(len(qps), dtype=np.uint8)   # coneqp: 1 "optimal" / 0 "unknown"
    qp_iters = np.zeros(len(qps), dtype=np.int32)    # coneqp iterations
    for k, (g, h, x, x_alt, status, iters) in enumerate(qps):
        qp_G[k, :g.shape[0]] = g
        qp_h[k, :h.shape[0]] = h
        qp_x[k] = x
        qp_x_alt[k] = x_alt
        qp_status[k] = status == "optimal"
        qp_iters[k] = iters
    meta = dict(name=name, env_id=env_id, shield=shield, n=n_cav, n_hdv=n_hdv, agent_reward=agent_reward,
                lateral_control=lateral_control, seed=seed, tape_seed=tape_seed,
                headway_time=headway_time, eta=eta, n_merge=int(env.n_merge),
                n_s=int(env.n_s), crashed=bool(env.is_crashed()), steps=t, action_masking=bool(action_masking),
                qp_solver=("exact-KKT closed form (cvxopt 1.2.7 unavailable)" if qp == "exact" else
                           "coneqp restatement of cvxopt's interior-point algorithm (cvxopt 1.2.7 unavailable)"))
    cvxopt.solvers.mode = "exact"
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        meta=json.dumps(meta),
        init_f=np.array(init_f), init_i=np.array(init_i, dtype=np.int32),
        obs0=np.asarray(obs0, dtype=np.float64), mask0=np.asarray(mask0, dtype=np.uint8),
        sub_f=np.array(sub_f), sub_i=np.array(sub_i, dtype=np.int32),
        sub_count=np.array(sub_count, dtype=np.int32),
        actions=np.array(rec["actions"], dtype=np.int32), obs=np.array(rec["obs"]),
        reward=np.array(rec["reward"]), done=np.array(rec["done"]),
        agents_rewards=np.array(rec["agents_rewards"]),
        regional_rewards=np.array(rec["regional_rewards"]),
        agents_dones=np.array(rec["agents_dones"]),
        average_speed=np.array(rec["average_speed"]), traffic_speed=np.array(rec["traffic_speed"]),
        min_headway=np.array(rec["min_headway"]), merge_percent=np.array(rec["merge_percent"]),
        action_mask=np.array(rec["action_mask"]), qp_count=np.array(rec["qp_count"], dtype=np.int32),
        qp_rows=qp_rows, qp_G=qp_G, qp_h=qp_h, qp_x=qp_x, qp_x_alt=qp_x_alt, qp_status=qp_status, qp_iters=qp_iters,
        # MDPLCVehicle.steering_angle after every sub-step (non-zero only under lateral_control="steer_vel")
        **({"sub_sa": np.array(sub_sa)} if lateral_control != "steer" else {}),
        # control-profile tail per sub-step: [shield ran, is_optimal, is_safe, is_invariant, headway]
        sub_pf=np.array(sub_pf),
        sl_t=np.array([p_[0] for p_ in probes], dtype=np.int32),
        sl_act=np.array([p_[1] for p_ in probes], dtype=np.float64).reshape(len(probes), n_cav, 2),
        sl_safe=np.array([p_[2] for p_ in probes], dtype=np.float64).reshape(len(probes), n_cav, 2),
        sl_status=np.array([p_[3] for p_ in probes], dtype=np.float64).reshape(len(probes), n_cav, 6))
    print("%-40s steps=%3d crashed=%d qps=%d" % (name, t, meta["crashed"], len(qps)))
    return meta


def gen_units():
    env = make_env("merge-multi-agent-v1", "none", 4, 1.2, 0.0)
    env.reset(is_training=False, testing_seeds=0)
    net = env.road.network
    rs = np.random.RandomState(7)
    out = {}
    # lane constants as the reference builds them (merge_env_v1.py:222-248)
    lanes = [net.get_lane(LANE_IX[i]) for i in range(6)]
    out["lane_start"] = np.array([l.start for l in lanes], dtype=np.float64)
    out["lane_end"] = np.array([l.end for l in lanes], dtype=np.float64)
    out["lane_length"] = np.array([l.length for l in lanes], dtype=np.float64)
    out["lane_forbidden"] = np.array([l.forbidden for l in lanes], dtype=np.uint8)
    sl = lanes[5]
    out["sine"] = np.array([sl.amplitude, sl.pulsation, sl.phase], dtype=np.float64)
    out["obstacle"] = np.array(env.road.objects[0].position, dtype=np.float64)
    # closest lane + next lane + per-lane frames on random poses (road.py:51-109, lane.py)
    n = 4000
    px = rs.uniform(-20, 520, n)
    py = rs.uniform(-3, 14, n)
    ph = rs.uniform(-0.6, 0.6, n)
    # plus poses that hug the lane centre lines (typical states)
    px[: n // 2] = rs.uniform(0, 500, n // 2)
    ysel = rs.randint(0, 4, n // 2)
    py[: n // 2] = np.where(ysel == 0, 0.0, np.where(ysel == 1, 4.0, np.where(ysel == 2, 10.5, 0)))
    onramp = ysel == 3
    py[: n // 2][onramp] = 7.25 + 3.25 * np.cos(np.pi * (px[: n // 2][onramp] - 220) / 100)
    py[: n // 2] += rs.normal(0, 0.3, n // 2)
    ph[: n // 2] = rs.normal(0, 0.05, n // 2)
    closest = np.zeros(n, dtype=np.int32)
    nxt = np.zeros((n, 6), dtype=np.int32)
    local = np.zeros((n, 6, 2))
    lane_heading = np.zeros((n, 6))
    dist_h = np.zeros((n, 6))
    onlane = np.zeros((n, 6), dtype=np.uint8)
    reach = np.zeros((n, 6), dtype=np.uint8)
    after = np.zeros((n, 6), dtype=np.uint8)
    for k in range(n):
        pos = np.array([px[k], py[k]])
        closest[k] = LANE_ID[net.get_closest_lane_index(pos, ph[k])]
        for li in range(6):
            l = lanes[li]
            s, r = l.local_coordinates(pos)
            local[k, li] = (s, r)
            lane_heading[k, li] = l.heading_at(s)
            dist_h[k, li] = l.distance_with_heading(pos, ph[k])
            onlane[k, li] = bool(l.on_lane(pos))
            reach[k, li] = bool(l.is_reachable_from(pos))
            after[k, li] = bool(l.after_end(pos))
            nxt[k, li] = LANE_ID[net.next_lane(LANE_IX[li], position=pos)]
    out.update(pose_x=px, pose_y=py, pose_h=ph, closest=closest, next_lane=nxt, local=local,
               lane_heading=lane_heading, dist_heading=dist_h, on_lane=onlane, reachable=reach,
               after_end=after)
    # steering_control / speed_control / get_corner (controller.py:146-197,257-267)
    veh = env.controlled_vehicles[0]
    m = 3000
    sx = rs.uniform(0, 500, m)
    sy = rs.uniform(-2, 12.5, m)
    sh = rs.normal(0, 0.15, m)
    sv = rs.uniform(0, 35, m)
    sv[:50] = rs.uniform(-0.02, 0.02, 50)  # not_zero branch
    stl = rs.randint(0, 6, m)
    steer = np.zeros(m)
    corner = np.zeros((m, 2, 2))
    for k in range(m):
        veh.position = np.array([sx[k], sy[k]])
        veh.heading = sh[k]
        veh.speed = sv[k]
        steer[k] = veh.steering_control(LANE_IX[int(stl[k])])
        corner[k, 0] = veh.get_corner("L")
        corner[k, 1] = veh.get_corner("R")
    out.update(sc_x=sx, sc_y=sy, sc_h=sh, sc_v=sv, sc_lane=stl.astype(np.int32), sc_steer=steer,
               corner=corner)
    # speed_to_index incl. exact .5 boundaries (controller.py:327-337); wrap_to_pi; not_zero
    sp = np.concatenate([np.linspace(0, 45, 181), [12.5, 17.5, 22.5, 27.5, 7.5, 32.5]])
    out["sti_speed"] = sp
    out["sti_index"] = np.array([veh.speed_to_index(s) for s in sp], dtype=np.int32)
    wa = np.concatenate([rs.uniform(-20, 20, 500), [np.pi, -np.pi, 0.0, 3 * np.pi, -3 * np.pi]])
    out["wrap_in"] = wa
    out["wrap_out"] = np.array([hutils.wrap_to_pi(a) for a in wa])
    # rotated rectangle intersection (utils.py:90-121), vehicle vs vehicle and vehicle vs obstacle
    q = 6000
    rect = np.zeros((q, 6))
    hit = np.zeros(q, dtype=np.uint8)
    hit_obs = np.zeros(q, dtype=np.uint8)
    for k in range(q):
        dx, dy = rs.uniform(-6, 6), rs.uniform(-3.5, 3.5)
        a1, a2 = rs.normal(0, 0.3), rs.normal(0, 0.3)
        cx, cy = rs.uniform(0, 400), rs.uniform(0, 10)
        rect[k] = (cx, cy, a1, cx + dx, cy + dy, a2)
        hit[k] = hutils.rotated_rectangles_intersect(
            (np.array([cx, cy]), 4.5, 1.8, a1), (np.array([cx + dx, cy + dy]), 4.5, 1.8, a2))
        hit_obs[k] = hutils.rotated_rectangles_intersect(
            (np.array([cx, cy]), 4.5, 1.8, a1), (np.array([cx + dx, cy + dy]), 1.8, 1.8, 0.0))
    out.update(rect=rect, rect_hit=hit, rect_hit_obstacle=hit_obs)
    np.savez_compressed(os.path.join(OUT, "units.npz"), **out)
    print("units.npz: %d poses, %d steering cases, %d rect pairs (%d/%d hits)"
          % (n, m, q, hit.sum(), hit_obs.sum()))


def gen_reset():
    """reset() of the reference for a list of (seed, n_cav): initial state + obs (abstract.py:176-209)."""
    rows = []
    for env_id in ("merge-multi-agent-v0", "merge-multi-agent-v1"):
        for n_cav in (2, 3, 4, 5, 8, 11):
            for seed in (0, 25, 50, 1000, 1001):
                env = make_env(env_id, "none", n_cav, 1.2, 0.0)
                obs, mask = env.reset(is_training=False, testing_seeds=seed)
                f, i = zip(*[_veh_snapshot(v) for v in env.road.vehicles])
                rows.append(dict(env=env_id, n=n_cav, seed=seed, n_merge=int(env.n_merge),
                                 f=np.array(f)[:, :5].tolist(), i=np.array(i)[:, :3].tolist(),
                                 obs=np.asarray(obs).tolist()))
    # also the reference's own count draw (merge_env_v1.py:180-211) for traffic_density 1..3
    counts = []
    for td in (1, 2, 3):
        for seed in (0, 25, 50, 75):
            env = gym.make("merge-multi-agent-v1")
            env.config["traffic_density"] = td
            env.config["traffic_type"] = "cav"
            env.reset(is_training=False, testing_seeds=seed)
            f, i = zip(*[_veh_snapshot(v) for v in env.road.vehicles])
            counts.append(dict(td=td, seed=seed, n=len(env.controlled_vehicles),
                               n_merge=int(env.n_merge), f=np.array(f)[:, :5].tolist()))
    # mixed traffic: CAV and HDV counts drawn, HDVs spawned on the remaining slots (:298-362)
    mixed = []
    for env_id in ("merge-multi-agent-v0", "merge-multi-agent-v1"):
        for td in (1, 2, 3):
            for seed in (0, 25, 50):
                env = gym.make(env_id)
                env.config["traffic_density"] = td
                env.config["traffic_type"] = "mixed"
                env.config["mixed_traffic"] = True
                env.config["safety_guarantee"] = "none"
                obs, mask = env.reset(is_training=False, testing_seeds=seed)
                f, i = zip(*[_veh_snapshot(v) for v in env.road.vehicles])
                mixed.append(dict(env=env_id, td=td, seed=seed, n=len(env.controlled_vehicles),
                                  n_all=len(env.road.vehicles), n_merge=int(env.n_merge),
                                  f=np.array(f)[:, :5].tolist(), timer=np.nan_to_num(np.array(f)[:, 10]).tolist(),
                                  kind=np.array(i)[:, 8].tolist(), obs=np.asarray(obs).tolist()))
    # reset(num_CAV=k): only the HDV count is drawn (merge_env_v1.py:186-187,194-195,202-203), as MAPPO.evaluation can ask
    numcav = []
    for tt in ("cav", "mixed"):
        for td, k in ((1, 2), (2, 3), (3, 5), (1, 1)):
            for seed in (0, 25):
                env = gym.make("merge-multi-agent-v1")
                env.config.update({"traffic_density": td, "traffic_type": tt, "mixed_traffic": tt == "mixed", "safety_guarantee": "none"})
                env.reset(is_training=False, testing_seeds=seed, num_CAV=k)
                f, i = zip(*[_veh_snapshot(v) for v in env.road.vehicles])
                numcav.append(dict(td=td, tt=tt, seed=seed, num_CAV=k, n=len(env.controlled_vehicles), n_all=len(env.road.vehicles),
                                   n_merge=int(env.n_merge), f=np.array(f)[:, :5].tolist(), kind=np.array(i)[:, 8].tolist()))
    with open(os.path.join(OUT, "reset.json"), "w") as fh:
        json.dump(dict(fixed=rows, drawn=counts, mixed=mixed, numcav=numcav), fh)
    print("reset.json: %d fixed-count resets, %d drawn-count resets" % (len(rows), len(counts)))


def main():
    os.makedirs(OUT, exist_ok=True)
    if not ONLY:
        gen_units()
        gen_reset()
    metas = []
    v0, v1 = "merge-multi-agent-v0", "merge-multi-agent-v1"
    # (1) unshielded, reference arithmetic end to end (configs_marl-cav-unsafe.ini: tau 1.2)
    for seed in (0, 25, 50):
        metas.append(run_episode("ep_v0_none_N4_s%d" % seed, v0, "none", 4, seed, 123 + seed, 1.2, 0.0))
    metas.append(run_episode("ep_v0_none_N8_s0", v0, "none", 8, 0, 7, 1.2, 0.0))
    metas.append(run_episode("ep_v1_none_N4_s0", v1, "none", 4, 0, 123, 1.2, 0.0))
    metas.append(run_episode("ep_v1_none_N8_s25", v1, "none", 8, 25, 11, 0.5, 0.0))
    # idle-only tapes run the full 100 steps without a shield (long-horizon drift check)
    metas.append(run_episode("ep_v0_idle_N4_s0", v0, "none", 4, 0, 0, 1.2, 0.0,
                             scripted=[(1, 1, 1, 1)]))
    # (2) shields; eta / tau from marl_cav-heading-t_headway-cbf-{avs_cint,cav}.ini
    for seed in (0, 25, 50):
        metas.append(run_episode("ep_v1_hss_N4_s%d" % seed, v1, "cbf-avs_cint", 4, seed, 123 + seed,
                                 0.5, 0.03125))
        metas.append(run_episode("ep_v1_mass_N4_s%d" % seed, v1, "cbf-cav", 4, seed, 123 + seed,
                                 0.5, 0.03125))
    for seed in (0, 25):
        metas.append(run_episode("ep_v1_mass_N8_s%d" % seed, v1, "cbf-cav", 8, seed, 200 + seed,
                                 0.5, 0.03125))
    metas.append(run_episode("ep_v1_hss_N8_s0", v1, "cbf-avs_cint", 8, 0, 200, 0.5, 0.03125))
    # lane-change heavy tapes exercise the LC veto / constrain_adj / collaborate paths
    lc = [0.3, 0.2, 0.3, 0.1, 0.1]
    metas.append(run_episode("ep_v1_mass_N8_lc_s75", v1, "cbf-cav", 8, 75, 5, 0.5, 0.03125, p=lc))
    metas.append(run_episode("ep_v1_hss_N4_lc_s75", v1, "cbf-avs_cint", 4, 75, 5, 0.5, 0.03125, p=lc))
    metas.append(run_episode("ep_v1_mass_N11_s100", v1, "cbf-cav", 11, 100, 9, 0.5, 0.03125))
    # a different eta / tau pair (eta_binary_search.sh:3-8 sweeps eta)
    metas.append(run_episode("ep_v1_mass_N4_eta05_s125", v1, "cbf-cav", 4, 125, 3, 1.2, 0.5))
    # (3) scripted crash scenarios of test/cbf (cbf_test_env.py:26-27,163-176,203-206,277-335), placed
    # on the merge map: a faster follower closing on a slower leader, the "extreme" speed pair, and a
    # ramp vehicle forcing its lane change into a main-road platoon.  Unshielded they crash; the
    # shields must not.
    A = {"LEFT": 0, "IDLE": 1, "RIGHT": 2, "FASTER": 3, "SLOWER": 4}
    scen = {
        "lon": ([(25, 0.0, 25), (65, 0.0, 20)], [(A["FASTER"], A["IDLE"])]),
        "lonx": ([(25, 0.0, 30), (65, 0.0, 15)], [(A["FASTER"], A["IDLE"])]),
        "lon3": ([(25, 0.0, 25), (65, 0.0, 25), (105, 0.0, 20)], [(A["FASTER"], A["FASTER"], A["SLOWER"])]),
        "merge": ([(170, 0.0, 25), (135, 0.0, 27), (172, 10.5, 25), (140, 10.5, 26)],
                  [(A["IDLE"], A["FASTER"], A["LEFT"], A["LEFT"])]),
    }
    for sname, (placement, script) in scen.items():
        for tag, shield in (("none", "none"), ("hss", "cbf-avs_cint"), ("mass", "cbf-cav")):
            metas.append(run_episode("sc_%s_%s" % (sname, tag), v1, shield, len(placement), 0, 0, 0.5, 0.03125,
                                     scripted=script, placement=placement))
    # (4) mixed traffic: CAVs + IDM/MOBIL HDVs (behavior.py:74-266; configs *-mixed*.ini)
    for seed, (nc, nh) in ((0, (3, 3)), (25, (2, 2)), (50, (4, 4))):
        metas.append(run_episode("mx_v0_none_%dc%dh_s%d" % (nc, nh, seed), v0, "none", nc, seed, 31 + seed, 1.2, 0.0, n_hdv=nh))
    for seed, (nc, nh) in ((0, (3, 3)), (25, (4, 3)), (50, (2, 2)), (75, (6, 5))):
        metas.append(run_episode("mx_v1_none_%dc%dh_s%d" % (nc, nh, seed), v1, "none", nc, seed, 41 + seed, 0.5, 0.0, n_hdv=nh))
        metas.append(run_episode("mx_v1_hss_%dc%dh_s%d" % (nc, nh, seed), v1, "cbf-avs_cint", nc, seed, 41 + seed, 0.5, 0.03125, n_hdv=nh))
        metas.append(run_episode("mx_v1_mass_%dc%dh_s%d" % (nc, nh, seed), v1, "cbf-cav", nc, seed, 41 + seed, 0.5, 0.03125, n_hdv=nh))
    lc = [0.3, 0.2, 0.3, 0.1, 0.1]
    metas.append(run_episode("mx_v1_mass_4c4h_lc_s100", v1, "cbf-cav", 4, 100, 7, 0.5, 0.03125, p=lc, n_hdv=4))
    # (5) alternate agent rewards (marl_cav-heading-t_headway-cbf-*-srew.ini / -mrew.ini)
    lc2 = [0.25, 0.3, 0.25, 0.1, 0.1]
    metas.append(run_episode("rw_v1_hss_srew_N4_s0", v1, "cbf-avs_cint", 4, 0, 77, 0.5, 0.03125, p=lc2, agent_reward="srew"))
    metas.append(run_episode("rw_v1_mass_mrew_N8_s25", v1, "cbf-cav", 8, 25, 78, 0.5, 0.03125, p=lc2, agent_reward="mrew"))
    metas.append(run_episode("rw_v1_mass_mrew_3c3h_s50", v1, "cbf-cav", 3, 50, 79, 0.5, 0.03125, p=lc2, n_hdv=3, agent_reward="mrew"))
    # (6) stand-alone safety_layer(...) probes: the reference function called per vehicle on deep copies
    metas.append(run_episode("sl_v1_hss_N4_s0", v1, "cbf-avs_cint", 4, 0, 91, 0.5, 0.03125, p=lc2, probe_shield=True))
    metas.append(run_episode("sl_v1_mass_N8_s25", v1, "cbf-cav", 8, 25, 92, 0.5, 0.03125, p=lc2, probe_shield=True))
    metas.append(run_episode("sl_v1_mass_3c3h_s50", v1, "cbf-cav", 3, 50, 93, 0.5, 0.03125, p=lc2, n_hdv=3, probe_shield=True))
    # (7) lateral_control = "steer_vel" (safe_controller.py:84-98,124-150): steering-velocity command, steering-angle state
    metas.append(run_episode("sv_v1_none_N4_s0", v1, "none", 4, 0, 61, 0.5, 0.0, p=lc2, lateral_control="steer_vel"))
    metas.append(run_episode("sv_v1_hss_N4_s25", v1, "cbf-avs_cint", 4, 25, 62, 0.5, 0.03125, p=lc2, lateral_control="steer_vel"))
    metas.append(run_episode("sv_v1_mass_N8_s50", v1, "cbf-cav", 8, 50, 63, 0.5, 0.03125, p=lc2, lateral_control="steer_vel"))
    metas.append(run_episode("sv_v1_mass_3c3h_s75", v1, "cbf-cav", 3, 75, 64, 0.5, 0.03125, p=lc2, n_hdv=3, lateral_control="steer_vel"))
    # (8) MM_QP_IPM fidelity mode: `solvers.qp` answered by the coneqp restatement (tools/refshim/cvxopt/coneqp.py)
    ipm = dict(qp="coneqp")
    metas.append(run_episode("ipm_v1_mass_N8_s0", v1, "cbf-cav", 8, 0, 200, 0.5, 0.03125, **ipm))
    metas.append(run_episode("ipm_v1_mass_N8_lc_s75", v1, "cbf-cav", 8, 75, 5, 0.5, 0.03125, p=lc, **ipm))
    metas.append(run_episode("ipm_v1_mass_N4_s25", v1, "cbf-cav", 4, 25, 148, 0.5, 0.03125, **ipm))
    metas.append(run_episode("ipm_v1_hss_N4_s50", v1, "cbf-avs_cint", 4, 50, 173, 0.5, 0.03125, **ipm))
    metas.append(run_episode("ipm_v1_hss_N8_lc_s0", v1, "cbf-avs_cint", 8, 0, 6, 0.5, 0.03125, p=lc, **ipm))
    metas.append(run_episode("ipm_v1_mass_N11_s100", v1, "cbf-cav", 11, 100, 9, 0.5, 0.03125, **ipm))
    metas.append(run_episode("ipm_v1_mass_4c3h_s25", v1, "cbf-cav", 4, 25, 66, 0.5, 0.03125, n_hdv=3, **ipm))
    metas.append(run_episode("ipm_v1_mass_mrew_N8_s25", v1, "cbf-cav", 8, 25, 78, 0.5, 0.03125, p=lc2, agent_reward="mrew", **ipm))
    metas.append(run_episode("ipm_sl_v1_mass_N8_s25", v1, "cbf-cav", 8, 25, 92, 0.5, 0.03125, p=lc2, probe_shield=True, **ipm))
    for sname in ("lonx", "merge"):  # slack-active / "unknown"-status QPs live in the hard-braking scenarios
        placement, script = scen[sname]
        for tag, shield in (("hss", "cbf-avs_cint"), ("mass", "cbf-cav")):
            metas.append(run_episode("ipm_sc_%s_%s" % (sname, tag), v1, shield, len(placement), 0, 0, 0.5, 0.03125,
                                     scripted=script, placement=placement, **ipm))
    # (9) action masking ON (abstract.py:219-240 _get_available_actions, :200-207,474-481 incl. the `[[0]*n_a]*n` aliasing:
    # every row of the returned mask is the same list object, so each agent's row is the OR over all agents)
    metas.append(run_episode("am_v0_none_N4_s0", v0, "none", 4, 0, 301, 1.2, 0.0, p=lc, action_masking=True))
    metas.append(run_episode("am_v0_none_N8_s25", v0, "none", 8, 25, 302, 1.2, 0.0, p=lc2, action_masking=True))
    metas.append(run_episode("am_v1_mass_N8_s50", v1, "cbf-cav", 8, 50, 303, 0.5, 0.03125, p=lc, action_masking=True))
    metas.append(run_episode("am_v1_hss_3c3h_s75", v1, "cbf-avs_cint", 3, 75, 304, 0.5, 0.03125, p=lc2, n_hdv=3, action_masking=True))
    # (10) placed edge cases the random tapes never reach
    I = A["IDLE"]
    #  x < 0 terminal (merge_env_v1.py:168-178): a vehicle rolling backwards across the start of the road
    metas.append(run_episode("ed_v0_xneg", v0, "none", 2, 0, 0, 1.2, 0.0, scripted=[(I, I)], placement=[(0.3, 0.0, -6.0), (80.0, 0.0, 25.0)]))
    metas.append(run_episode("ed_v1_xneg_none", v1, "none", 2, 0, 0, 0.5, 0.0, scripted=[(I, I)], placement=[(0.3, 0.0, -8.0), (90.0, 0.0, 25.0)]))
    #  |v| > MAX_SPEED clamp of clip_actions (kinematics.py:143-152), without and with the shields
    fast = [(60.0, 0.0, 46.0), (160.0, 0.0, 44.5), (40.0, 10.5, 43.0)]
    for tag, shield in (("none", "none"), ("hss", "cbf-avs_cint"), ("mass", "cbf-cav")):
        metas.append(run_episode("ed_v1_fast_%s" % tag, v1, shield, 3, 0, 0, 0.5, 0.03125, max_steps=12,
                                 scripted=[(A["FASTER"], I, A["FASTER"])], placement=fast))
    metas.append(run_episode("ed_v0_fast", v0, "none", 3, 0, 0, 1.2, 0.0, max_steps=12, scripted=[(A["FASTER"], I, A["FASTER"])], placement=fast))
    #  QP corner branches: a = g.vx dt <= 0 (a vehicle pointing backwards: cos(heading + beta) < 0) and lo > hi
    #  (negative speed under MASS: v_min = max(0, .) > v_max) -- exact mode only, the IPM has no answer to pin there
    odd = [(150.0, 0.0, 12.0, 3.05), (200.0, 0.0, 20.0, 0.0), (120.0, 0.0, -3.0, 0.0), (90.0, 0.0, 24.0, 0.0)]
    for tag, shield in (("hss", "cbf-avs_cint"), ("mass", "cbf-cav")):
        metas.append(run_episode("ed_v1_odd_%s" % tag, v1, shield, 4, 0, 0, 0.5, 0.03125, max_steps=8,
                                 scripted=[(I, I, I, A["FASTER"])], placement=odd))
    #  pile-ups: check_collision returns at once for an already-crashed vehicle (kinematics.py:185-186), so in the
    #  creation-order pair loop (road.py:288-292) a third overlapping vehicle stays intact -- incl. out-of-order creation
    for tag, xs in (("a", (100.0, 103.0, 106.0)), ("b", (100.0, 102.0, 104.0)), ("c", (100.0, 104.0, 102.0)), ("d", (104.0, 100.0, 102.0, 140.0))):
        metas.append(run_episode("ed_v0_pileup_%s" % tag, v0, "none", len(xs), 0, 0, 1.2, 0.0, max_steps=3,
                                 scripted=[(I,) * len(xs)], placement=[(x, 0.0, 20.0) for x in xs]))
    metas.append(run_episode("ed_v1_pileup_mass", v1, "cbf-cav", 3, 0, 0, 0.5, 0.03125, max_steps=3,
                             scripted=[(I, I, I)], placement=[(100.0, 0.0, 20.0), (103.0, 0.0, 21.0), (106.0, 0.0, 19.0)]))
    #  crashed HDVs: an HDV-HDV crash does not end the episode (merge_env_v1.py:168-178 looks at controlled vehicles only);
    #  the wrecks stop acting (behavior.py:83-84), brake by clip_actions' crashed branch (kinematics.py:144-146) and a CAV
    #  approaches them with and without a shield
    wreck = [(60.0, 0.0, 25.0), (20.0, 10.5, 24.0), (150.0, 0.0, 22.0, None, "h"), (153.0, 0.0, 18.0, None, "h"), (100.0, 10.5, 20.0, None, "h")]
    for tag, shield in (("none", "none"), ("hss", "cbf-avs_cint"), ("mass", "cbf-cav")):
        metas.append(run_episode("ed_v1_wreck_%s" % tag, v1, shield, 2, 0, 0, 0.5, 0.03125, max_steps=40,
                                 scripted=[(I, I)], placement=wreck, n_hdv=3))
    metas.append(run_episode("ed_v0_wreck", v0, "none", 2, 0, 0, 1.2, 0.0, max_steps=40, scripted=[(I, A["FASTER"])], placement=wreck, n_hdv=3))
    # the index is rebuilt from the tapes on disk, so a partial regeneration (`only <prefix> ...`) keeps the rest
    import glob
    metas = []
    for f in sorted(glob.glob(os.path.join(OUT, "*_*.npz"))):
        m = json.loads(str(np.load(f)["meta"]))
        m["name"] = os.path.basename(f)[:-4]
        metas.append(m)
    with open(os.path.join(OUT, "index.json"), "w") as fh:
        json.dump(metas, fh, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "only":  # e.g. `gen_golden.py only sv_ sl_`: regenerate just those tapes
        ONLY = sys.argv[2:]
    if len(sys.argv) > 1 and sys.argv[1] == "reset":
        os.makedirs(OUT, exist_ok=True)
        gen_reset()
    else:
        main()
