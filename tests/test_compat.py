"""Host-side mirror of the reference's Env interface (MergeEnvCompat), driven on CPU through the
oracle backend: reset RNG parity with the reference and step() tuple parity with the golden tapes."""
import json
import os

import numpy as np
import pytest

import oracle_env
from golden_util import GOLDEN, is_ipm, load_episode
from marl_mass_amd import compat
from marl_mass_amd import _cabi as abi


def _factory(**kw):
    return oracle_env.OracleEnv(**kw)


@pytest.fixture(scope="module")
def resets():
    return json.load(open(os.path.join(GOLDEN, "reset.json")))


def test_reset_matches_reference_fixed_counts(resets):
    """abstract.py:176-209 + merge_env_v1.py:265-364: same global-RNG draws -> same spawn, bit for bit."""
    for row in resets["fixed"]:
        env = compat.MergeEnvCompat(row["env"], backend_factory=_factory)
        env._num_vehicles = lambda num_CAV=0, n=row["n"]: (n, 0)  # the fixture forced the count the same way
        obs, avail = env.reset(is_training=False, testing_seeds=row["seed"])
        f = np.array(row["f"])
        got = env._b.f64[:5, 0, :row["n"]].numpy().T
        np.testing.assert_array_equal(got, f)
        ints = np.array(row["i"])
        np.testing.assert_array_equal(env._b.u8[abi.B["LANE"], 0, :row["n"]].numpy(), ints[:, 0])
        np.testing.assert_array_equal(env._b.u8[abi.B["SPEED_INDEX"], 0, :row["n"]].numpy(), ints[:, 2])
        assert env.n_merge == row["n_merge"]
        np.testing.assert_allclose(obs, np.array(row["obs"]), rtol=0, atol=1e-12)
        assert obs.shape == (row["n"], env.n_s) and avail.shape == (row["n"], 5)


def test_reset_matches_reference_drawn_counts(resets):
    """merge_env_v1.py:180-211: the vehicle-count draw for traffic_density 1..3, CAV-only traffic."""
    for row in resets["drawn"]:
        env = compat.MergeEnvCompat("merge-multi-agent-v1", backend_factory=_factory)
        env.config["traffic_density"] = row["td"]
        env.config["traffic_type"] = "cav"
        env.reset(is_training=False, testing_seeds=row["seed"])
        assert len(env.controlled_vehicles) == row["n"] and env.n_merge == row["n_merge"]
        np.testing.assert_array_equal(env._b.f64[:5, 0, :row["n"]].numpy().T, np.array(row["f"]))


def test_reset_matches_reference_mixed_traffic(resets):
    """Mixed traffic: CAV + HDV counts drawn, HDVs on the remaining spawn slots, IDM timers."""
    for row in resets["mixed"]:
        env = compat.MergeEnvCompat(row["env"], backend_factory=_factory)
        env.config.update({"traffic_density": row["td"], "traffic_type": "mixed", "mixed_traffic": True,
                           "safety_guarantee": "none"})
        obs, _ = env.reset(is_training=False, testing_seeds=row["seed"])
        n, n_all = row["n"], row["n_all"]
        assert len(env.controlled_vehicles) == n and len(env.road.vehicles) == n_all and env.n_merge == row["n_merge"]
        np.testing.assert_array_equal(env._b.f64[:5, 0, :n_all].numpy().T, np.array(row["f"]))
        np.testing.assert_array_equal(env._b.u8[abi.B["KIND"], 0, :n_all].numpy(), np.array(row["kind"]))
        hd = np.array(row["kind"]) == 2
        np.testing.assert_allclose(env._b.f64[abi.F["G_VX"], 0, :n_all].numpy()[hd], np.array(row["timer"])[hd], rtol=0, atol=1e-12)
        np.testing.assert_allclose(obs, np.array(row["obs"]), rtol=0, atol=1e-12)


def test_reset_with_num_cav_matches_reference(resets):
    """reset(num_CAV=k) (merge_env_v1.py:186-187,194-195,202-203; MAPPO.evaluation's fixed-CAV-count episodes): the CAV count
    is taken, only the HDV count is drawn -- CAV-only traffic turns the drawn HDVs into CAVs as well (:206-209)."""
    for row in resets["numcav"]:
        env = compat.MergeEnvCompat("merge-multi-agent-v1", backend_factory=_factory)
        env.config.update({"traffic_density": row["td"], "traffic_type": row["tt"], "mixed_traffic": row["tt"] == "mixed",
                           "safety_guarantee": "none"})
        env.reset(is_training=False, testing_seeds=row["seed"], num_CAV=row["num_CAV"])
        n, n_all = row["n"], row["n_all"]
        assert len(env.controlled_vehicles) == n and len(env.road.vehicles) == n_all and env.n_merge == row["n_merge"], row
        np.testing.assert_array_equal(env._b.f64[:5, 0, :n_all].numpy().T, np.array(row["f"]))
        np.testing.assert_array_equal(env._b.u8[abi.B["KIND"], 0, :n_all].numpy(), np.array(row["kind"]))


def test_training_seed_increments():
    env = compat.MergeEnvCompat("merge-multi-agent-v1", backend_factory=_factory)
    env.config["traffic_type"] = "cav"
    s0 = env.seed
    env.reset()
    assert env.seed == s0 + 1  # abstract.py:190


@pytest.fixture(autouse=True)
def _restore_cbf_knobs():
    """CBFType is class-level state (as in the reference): every test leaves it as it found it."""
    saved = (compat.CBFType.GAMMA_B, compat.CBFType.TAU, compat.CBFType.QP_SOLVER)
    yield
    compat.CBFType.GAMMA_B, compat.CBFType.TAU, compat.CBFType.QP_SOLVER = saved


def test_drop_in_default_is_the_reference_solver_behaviour():
    """The adapter answers solvers.qp with cvxopt's interior-point iterate unless told otherwise (the closed form is an
    explicit opt-in): the backend it builds runs qp_solver = ipm."""
    assert compat.CBFType.QP_SOLVER == "ipm"
    env = compat.MergeEnvCompat("merge-multi-agent-v1", backend_factory=_factory)
    env.config.update({"safety_guarantee": "cbf-cav", "traffic_type": "cav", "traffic_density": 1})
    env.reset(is_training=False, testing_seeds=0)
    assert env._b._cfg.qp_solver == abi.QP_IPM


@pytest.mark.parametrize("name", ["ep_v0_none_N4_s25", "ep_v1_mass_N8_s0", "ep_v1_hss_N4_s50", "mx_v1_mass_4c3h_s25",
                                  "mx_v0_none_3c3h_s0", "sv_v1_hss_N4_s25", "ipm_v1_mass_N8_s0", "ipm_v1_hss_N4_s50",
                                  "ipm_v1_mass_4c3h_s25"])
def test_step_tuple_matches_golden(name):
    """The (obs, reward, done, info) tuple of MergeEnv.step through the adapter, free-running."""
    z, meta = load_episode(os.path.join(GOLDEN, name + ".npz"))
    compat.CBFType.GAMMA_B, compat.CBFType.TAU = meta["eta"], meta["headway_time"]
    compat.CBFType.QP_SOLVER = "ipm" if is_ipm(meta) else "exact"  # what answered solvers.qp while the reference produced the tape
    env = compat.MergeEnvCompat(meta["env_id"], backend_factory=_factory)
    env.config.update({"safety_guarantee": meta["shield"], "HEADWAY_TIME": meta["headway_time"],
                       "action_masking": False, "traffic_type": "cav", "mixed_traffic": False,
                       "lateral_control": meta.get("lateral_control", "steer")})
    env._num_vehicles = lambda num_CAV=0: (meta["n"], meta.get("n_hdv", 0))
    obs, avail = env.reset(is_training=False, testing_seeds=meta["seed"])
    np.testing.assert_allclose(obs, z["obs0"], rtol=0, atol=1e-12)
    assert env.n_s == meta["n_s"] and env.T == 100 and len(env.controlled_vehicles) == meta["n"]
    for t in range(meta["steps"]):
        obs, reward, done, info = env.step(tuple(int(a) for a in z["actions"][t]))
        np.testing.assert_allclose(obs, z["obs"][t], rtol=0, atol=1e-9)
        assert abs(reward - z["reward"][t]) <= 1e-9 and done == bool(z["done"][t])
        np.testing.assert_allclose(info["regional_rewards"], z["regional_rewards"][t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(info["agents_rewards"], z["agents_rewards"][t], rtol=0, atol=1e-9)
        assert info["agents_dones"] == tuple(bool(d) for d in z["agents_dones"][t])
        assert abs(info["average_speed"] - z["average_speed"][t]) <= 1e-9
        assert abs(info["min_headway"] - z["min_headway"][t]) <= 1e-9
        assert info["vehicle_speed"].shape == (t + 1, meta["n"])
    assert done and abs(info["merge_percent"] - z["merge_percent"][meta["steps"] - 1]) <= 1e-9
    assert env.is_crashed() == meta["crashed"]
    assert env.controlled_vehicles[0].lane_index in abi.LANE_INDEX


@pytest.mark.parametrize("name", ["sl_v1_mass_3c3h_s50", "sv_v1_mass_N8_s50"])
def test_control_profile_matches_reference(name):
    """store_profile: the per-sub-step state_hist / action_hist records (safe_controller.py:187-227,
    behavior.py:509-521) that MAPPOControlEval.evaluation exports (marl/mappo.py:420-438)."""
    from golden_util import SF, SI
    z, meta = load_episode(os.path.join(GOLDEN, name + ".npz"))
    compat.CBFType.GAMMA_B, compat.CBFType.TAU = meta["eta"], meta["headway_time"]
    compat.CBFType.QP_SOLVER = "ipm" if is_ipm(meta) else "exact"
    env = compat.MergeEnvCompat(meta["env_id"], backend_factory=_factory, store_profile=True)
    env.config.update({"safety_guarantee": meta["shield"], "HEADWAY_TIME": meta["headway_time"], "action_masking": False,
                       "lateral_control": meta.get("lateral_control", "steer")})
    env._num_vehicles = lambda num_CAV=0: (meta["n"], meta.get("n_hdv", 0))
    env.reset(is_training=False, testing_seeds=meta["seed"])
    steps = 30
    for t in range(steps):
        env.step(tuple(int(a) for a in z["actions"][t]))
    cp = env.control_profile()
    n, n_all = meta["n"], meta["n"] + meta.get("n_hdv", 0)
    assert sorted(cp) == sorted(["av%d" % j for j in range(n)] + ["hdv%d" % j for j in range(n, n_all)])
    nsub = int(z["sub_count"][:steps].sum())
    for j in range(n_all):
        rec = cp[("av%d" if j < n else "hdv%d") % j]
        assert len(rec["state_hist"]) == len(rec["action_hist"]) == nsub
        for k in (0, 1, 2, 7, nsub - 1):
            gf, pf = z["sub_f"][k][j], z["sub_pf"][k][j]
            st, ac = rec["state_hist"][k], rec["action_hist"][k]
            assert abs(st["x"] - gf[SF["x"]]) <= 1e-9 and abs(st["speed"] - gf[SF["speed"]]) <= 1e-9
            assert abs(st["t_step"] - (k + 1) / 15) <= 1e-9 and abs(ac["t_step"] - st["t_step"]) == 0
            assert abs(ac["steering"] - gf[SF["safe_steer" if j < n else "act_steer"]]) <= 1e-9
            if j < n:
                assert abs(ac["acceleration"] - gf[SF["safe_acc"]]) <= 1e-9
                assert abs(ac["ull_acceleration"] - gf[SF["act_acc"]]) <= 1e-9
                assert ("safe_status" in st) == bool(pf[0]) == ("safe_diff" in ac)
                assert abs(st["headway"] - pf[4]) <= 1e-9
                if pf[0]:
                    assert (st["safe_status"]["is_optimal"], st["safe_status"]["is_safe"], st["safe_status"]["is_invariant"]) == \
                        (bool(pf[1]), bool(pf[2]), bool(pf[3]))
                if "sub_sa" in z.files:
                    assert abs(st["steering_angle"] - z["sub_sa"][k][j]) <= 1e-9
                assert ac["lc_action"] == int(z["sub_i"][k][j][SI["hl_action"]])


def test_error_behaviour():
    with pytest.raises(ValueError):
        compat.cbf_factory("nope", action_size=2, action_bound=[(0, 1), (-1, 1)], vehicle_size=[5, 2], vehicle_lane=0)
    with pytest.raises(ValueError):
        compat.safety_layer("nope", {}, None, 1 / 15)
    with pytest.raises(ValueError):
        compat.MergeEnvCompat("merge-v1", backend_factory=_factory)
    env = compat.MergeEnvCompat("merge-multi-agent-v1", backend_factory=_factory)
    env.config["safety_guarantee"] = "cbf-avlon"  # unknown CBF type: ValueError like decentral_layer.py:817
    with pytest.raises(ValueError):
        env.reset()


def test_cbf_control_barrier_shim_matches_golden_qp():
    """cbf_factory(...).control_barrier (cbf.py:110-161) packs the same G/h the reference logged."""
    z, meta = load_episode(os.path.join(GOLDEN, "ep_v1_mass_N4_s0.npz"))
    solver = oracle_env.OracleEnv(1, 2)
    dt = 1 / 15
    # rebuild one QP from its logged rows: a = G[0,0], bounds from h1/h2 with u_ll[0] = 20
    k = int(np.argmax(z["qp_x"][:, 0] < -1e-3))
    G, h = z["qp_G"][k], z["qp_h"][k]
    u0 = 20.0
    cbf = compat.cbf_factory("cav", solver=solver, action_size=2, action_bound=[(u0 - h[2], u0 + h[1]), (-4 * np.pi, 4 * np.pi)],
                             vehicle_size=[5.0, 2.0], vehicle_lane=0)
    # choose x / f / g / u so that the assembled row equals the logged one: p_lon.x = 0, u_ol = 0
    compat.CBFType.GAMMA_B = 0.0
    g = np.diag([G[0, 0], dt, dt, dt, dt, dt, dt, dt])
    x = np.zeros(8)
    cbf.safe_dists = [0, 0, 0]
    u_ll = np.array([u0, 0, 0, 0, 0, 0, 0, 0.0])
    u_safe = cbf.control_barrier(u_ll, x.copy(), g, x, dt)
    h0 = -G[0, 0] * u0  # what the shim's row evaluates to for this construction
    d = min(0.0, h0 / G[0, 0])
    d = min(max(d, -h[2]), h[1])
    assert abs(u_safe[0] - (u0 + d)) <= 1e-12


def test_safety_layer_shim():
    """compat.safety_layer(safety_type, action, vehicle, dt) -> (safe_action, safe_diff, status), reference signature."""
    compat.CBFType.GAMMA_B, compat.CBFType.TAU = 0.03125, 0.5
    env = compat.MergeEnvCompat("merge-multi-agent-v1", backend_factory=_factory)
    env.config.update({"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5, "traffic_type": "cav", "traffic_density": 2})
    env.reset(is_training=False, testing_seeds=3)
    n = len(env.controlled_vehicles)
    veh = env.controlled_vehicles[1]
    act = {"steering": 0.01, "acceleration": 3.0}
    # first two sub-steps of an episode: shield gated off -> action returned, diff / status None
    sa, sd, st = compat.safety_layer("cav", act, veh, 1 / 15)
    assert sa == act and sd is None and st is None
    for _ in range(3):
        env.step((1,) * n)
    sa, sd, st = compat.safety_layer("cav", act, veh, 1 / 15)
    assert set(sa) == {"acceleration", "steering"} and set(st) >= {"is_optimal", "is_safe", "is_invariant"}
    assert abs(sd["acceleration"] - (sa["acceleration"] - act["acceleration"])) < 1e-15
    assert sa["acceleration"] <= 6.0 + 1e-9  # within the acceleration bound of the QP
    # the call's side effects on the vehicle (decentral_layer.py:725-752): the three flags, and the target lane on a veto --
    # what the reference's own probes recorded as vehicle attributes after the call (tests/golden/sl_*: status columns 3-5)
    for v in env.controlled_vehicles:
        sa_v, _, st_v = compat.safety_layer("cav", {"steering": 0.2, "acceleration": 1.0}, v, 1 / 15)
        assert (v.is_lc_safe, v.is_collaborating, v.collaborate_adj) == (st_v["is_lc_safe"], st_v["is_collaborating"], st_v["collaborate_adj"])
        if not st_v["is_lc_safe"]:
            assert v.target_lane_index == v.lane_index
    # vehicle.set_min_headway of the call (decentral_layer.py:466,700 -> safe_controller.py:264-265): the gap to the leader the
    # shield selected over the ego's longitudinal speed -- (x_ol - x_e - LENGTH) / vx_e, 181 m ahead when there is no leader
    xs = sorted(float(w.position[0]) for w in env.controlled_vehicles)
    for v in env.controlled_vehicles:
        compat.safety_layer("cav", {"steering": 0.0, "acceleration": 0.0}, v, 1 / 15)
        hw = v.min_headway
        vx = max(float(v.speed) * float(np.cos(v.heading)), 1.0)
        assert np.isfinite(hw) and -5.0 / vx <= hw <= (181.0 - 5.0) / vx + 1e-9
        if float(v.position[0]) == xs[-1]:  # the front vehicle of the road has the phantom leader
            assert abs(hw - (181.0 - 5.0) / vx) < 1e-9
    with pytest.raises(ValueError):
        compat.safety_layer("avs_cint", act, veh, 1 / 15)  # env is configured for MASS
    with pytest.raises(ValueError):
        compat.safety_layer("cav", act, veh, 1 / 15, safe_dist="fixed")
