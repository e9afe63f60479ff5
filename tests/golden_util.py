"""Replay of the reference's golden episode tapes (tests/golden/ep_*.npz) on any backend.

The tapes were produced by tools/gen_golden.py from the reference itself.  cvxopt is unavailable in the image, so a
stand-in answered `solvers.qp` while the reference ran; each shield tape's meta says which one (`qp_solver`): the
closed-form exact-KKT solve (ep_ / mx_ / sc_ / ... tapes) or the restatement of cvxopt's coneqp (the 13 ipm_* tapes) --
is_ipm(meta) below -- and every tape also carries the other solver's answer to the same (G, h) as `qp_x_alt`.
"""
import glob
import json
import os

import numpy as np
import torch

from marl_mass_amd import _cabi as abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# columns of sub_f / sub_i written by tools/gen_golden.py:_veh_snapshot
SF = {"x": 0, "y": 1, "heading": 2, "speed": 3, "target_speed": 4, "act_steer": 5, "act_acc": 6,
      "safe_steer": 7, "safe_acc": 8, "g_vx": 9, "timer": 10}
SI = {"lane": 0, "target_lane": 1, "speed_index": 2, "crashed": 3, "hl_action": 4,
      "collaborate_adj": 5, "is_lc_safe": 6, "is_collaborating": 7, "kind": 8}

FLOAT_TOL = 1e-5  # north_star: "within 1e-5 on float state"


def episode_files(pattern="*_*.npz"):
    """ep_*: random / idle tapes from reference spawns; sc_*: scripted crash scenarios (test/cbf);
    mx_*: mixed traffic (CAVs + IDM/MOBIL HDVs); rw_*: srew / mrew agent rewards; sl_*: with stand-alone
    safety_layer probes; sv_*: lateral_control = "steer_vel"; ipm_*: `solvers.qp` answered by the coneqp
    restatement (MM_QP_IPM mode); am_*: action masking on; ed_*: placed edge cases."""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, pattern))
                  if os.path.basename(f).split("_")[0] in ("ep", "sc", "mx", "rw", "sl", "sv", "ipm", "am", "ed"))


def is_ipm(meta):
    return meta.get("qp_solver", "").startswith("coneqp")


def load_episode(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def env_kwargs(meta):
    cfg = {"safety_guarantee": meta["shield"], "HEADWAY_TIME": meta["headway_time"],
           "action_masking": bool(meta.get("action_masking", False)), "agent_reward": meta.get("agent_reward", "default"),
           "lateral_control": meta.get("lateral_control", "steer")}
    kw = dict(env_id=meta["env_id"], config=cfg, cbf_eta=meta["eta"], cbf_tau=meta["headway_time"],
              obs_f64=True, trace=True, n_hdv=meta.get("n_hdv", 0))
    # the ipm_* tapes (cvxopt's interior-point iterate answered solvers.qp) replay on the product's DEFAULT numerics -- no
    # qp_solver argument; the tapes recorded with the exact KKT point name that mode
    if not is_ipm(meta):
        kw["qp_solver"] = "exact"
    return kw


KNIFE_EDGE = 1e-9  # |LC margin| below this = the reference's own decision is rounding noise


def _force_state(env, z, meta, t, s_at):
    """Teacher forcing: load the reference's state at the start of step t (end of sub-step s_at-1)."""
    F, B, EP = abi.F, abi.B, abi.EP
    n = meta["n"] + meta.get("n_hdv", 0)
    gf, gi = z["sub_f"][s_at - 1].copy(), z["sub_i"][s_at - 1]
    hdv = gi[:, SI["kind"]] == 2
    # HDVs keep their last IDM action in the SAFE_* planes and the MOBIL timer in G_VX (mm_abi.h)
    gf[hdv, SF["safe_steer"]] = gf[hdv, SF["act_steer"]]
    gf[hdv, SF["safe_acc"]] = gf[hdv, SF["act_acc"]]
    gf[hdv, SF["g_vx"]] = gf[hdv, SF["timer"]]
    dev = env.device
    put = lambda plane, v: plane.__setitem__(0, torch.as_tensor(np.asarray(v), device=dev).to(plane.dtype))  # noqa: E731
    for name, col in (("X", "x"), ("Y", "y"), ("HEADING", "heading"), ("SPEED", "speed"),
                      ("TARGET_SPEED", "target_speed"), ("SAFE_STEER", "safe_steer"), ("SAFE_ACC", "safe_acc"),
                      ("G_VX", "g_vx")):
        put(env.f64[F[name]], gf[:, SF[col]])
    h1 = np.stack([gf[:, SF["x"]], gf[:, SF["speed"]] * np.cos(gf[:, SF["heading"]])])
    if s_at >= 2:
        pf = z["sub_f"][s_at - 2]
        h2 = np.stack([pf[:, SF["x"]], pf[:, SF["speed"]] * np.cos(pf[:, SF["heading"]])])
    else:
        h2 = np.zeros((2, n))
    if "sub_sa" in z.files:
        put(env.f64[F["STEER_ANGLE"]], z["sub_sa"][s_at - 1])
    for k, nm in enumerate(("X", "VX")):
        put(env.f64[F["H1_" + nm]], h1[k])
        put(env.f64[F["H2_" + nm]], h2[k])
    put(env.u8[B["LANE"]], gi[:, SI["lane"]])
    put(env.u8[B["TARGET_LANE"]], gi[:, SI["target_lane"]])
    put(env.u8[B["SPEED_INDEX"]], gi[:, SI["speed_index"]])
    put(env.u8[B["CRASHED"]], gi[:, SI["crashed"]])
    put(env.u8[B["HL_ACTION"]], np.where(gi[:, SI["hl_action"]] < 0, 255, gi[:, SI["hl_action"]]))
    put(env.u8[B["FLAGS"]], gi[:, SI["collaborate_adj"]] * abi.FLAG_COLLABORATE_ADJ
        + gi[:, SI["is_lc_safe"]] * abi.FLAG_IS_LC_SAFE + gi[:, SI["is_collaborating"]] * abi.FLAG_IS_COLLABORATING)
    put(env.u8[B["HIST_LEN"]], np.full(n, min(2, s_at)))
    put(env.u8[B["KIND"]], gi[:, SI["kind"]])
    env.env_i32[EP["STEPS"], 0] = t
    env.env_i32[EP["TIME"], 0] = s_at
    env.env_i32[EP["N_MERGE"], 0] = meta["n_merge"]


def replay(make_env, path, tol=FLOAT_TOL, check_qp=True, teacher_forcing=True, max_knife_edges=0):
    """Run one golden tape on `make_env(E=1, N=n, **kw)` and assert parity with the reference.

    Every step starts from the reference's own state (teacher forcing), so each of the tape's
    steps is an independent check.  Discrete state, flags and QP structure must match exactly and
    floats within `tol` -- except where the lane-change test of the shield sits on its structural
    knife-edge (|LC margin| < 1e-9, see include/mm_math.h): those steps are counted in
    err["knife_edges"] (bounded by `max_knife_edges`) and skipped."""
    z, meta = load_episode(path)
    nc = meta["n"]                    # controlled vehicles
    n = nc + meta.get("n_hdv", 0)     # all vehicles on the road
    env = make_env(E=1, N=n, **env_kwargs(meta))
    f0, dev = z["init_f"], env.device
    i0 = z["init_i"]
    kind0 = i0[:, SI["kind"]] if i0.shape[1] > SI["kind"] else np.ones(n, dtype=np.int64)
    obs, _ = env.set_kinematics(f0[None, :, 0], f0[None, :, 1], f0[None, :, 2], f0[None, :, 3],
                                n_merge=np.array([meta["n_merge"]]), kind=kind0[None])
    # derived initial state (Vehicle / MDPVehicle / IDMVehicle.__init__)
    assert np.array_equal(env.u8[abi.B["LANE"], 0].cpu().numpy(), i0[:, SI["lane"]])
    assert np.array_equal(env.u8[abi.B["SPEED_INDEX"], 0].cpu().numpy(), i0[:, SI["speed_index"]])
    np.testing.assert_allclose(env.f64[abi.F["TARGET_SPEED"], 0].cpu().numpy(), f0[:, SF["target_speed"]], atol=0, rtol=0)
    if (kind0 == 2).any():
        np.testing.assert_allclose(env.f64[abi.F["G_VX"], 0].cpu().numpy()[kind0 == 2], f0[kind0 == 2, SF["timer"]],
                                   atol=1e-12, rtol=0)
    err = {"obs0": float(np.abs(obs[0, :nc].cpu().numpy() - z["obs0"]).max())}
    assert err["obs0"] <= tol, err
    sub_f, sub_i, sub_count = z["sub_f"], z["sub_i"], z["sub_count"]
    qp_G, qp_h, qp_x, qp_rows = z["qp_G"], z["qp_h"], z["qp_x"], z["qp_rows"]
    is_v1 = meta["env_id"].endswith("v1")
    shielded = meta["shield"] != "none"
    mx = dict(state=0.0, action=0.0, obs=0.0, reward=0.0, info=0.0, qp=0.0)
    knife = 0
    s_at, q_at = 0, 0

    class _KnifeEdge(Exception):
        pass

    def discrete(cond, tr, k, ctx):
        """A discrete mismatch is legitimate only on the LC knife-edge of this or an earlier sub-step."""
        if cond:
            return
        marg = tr[: k + 1, abi.T["LC_MARGIN"]]
        if shielded and np.nanmin(np.abs(np.where(np.isnan(marg), np.inf, marg))) < KNIFE_EDGE:
            raise _KnifeEdge()
        raise AssertionError(ctx)

    for t in range(meta["steps"]):
        nsub = int(sub_count[t])
        nqp = int(z["qp_count"][t])
        if teacher_forcing and t > 0:
            _force_state(env, z, meta, t, s_at)
        act = np.ones((1, n), dtype=np.int32)
        act[0, :nc] = z["actions"][t]
        a = torch.tensor(act, dtype=torch.int32, device=dev)
        obs, rew, done, out = env.step(a)
        tr = env.trace[:, :, 0].cpu().numpy()  # [3, T, n]
        step_mx = dict(mx)
        try:
            ran = ~np.isnan(tr[:, abi.T["X"], 0])
            discrete(int(ran.sum()) == nsub, tr, 2, (path, t, "sub-step count", ran, nsub))
            q_loc = q_at
            for k in range(nsub):
                gf, gi = sub_f[s_at + k], sub_i[s_at + k]
                for name, col in (("LANE", "lane"), ("TARGET_LANE", "target_lane"), ("CRASHED", "crashed")):
                    got = tr[k, abi.T[name]].astype(np.int64)
                    discrete(np.array_equal(got, gi[:, SI[col]]), tr, k, (path, t, k, name, got, gi[:, SI[col]]))
                if is_v1 and shielded:
                    fl = tr[k, abi.T["FLAGS"]].astype(np.int64)
                    rows = tr[k, abi.T["QP_ROWS"]].astype(np.int64)
                    if rows.any():  # shield ran this sub-step: flags are defined
                        fl = np.where(gi[:, SI["kind"]] == 1, fl, 0)
                        for bit, col in ((abi.FLAG_IS_LC_SAFE, "is_lc_safe"), (abi.FLAG_IS_COLLABORATING, "is_collaborating"),
                                         (abi.FLAG_COLLABORATE_ADJ, "collaborate_adj")):
                            discrete(np.array_equal((fl & bit) != 0, gi[:, SI[col]] != 0), tr, k, (path, t, k, col))
                    if check_qp and rows.any():
                        # the reference solves in road.step order = descending pre-step x
                        xs_prev = (sub_f[s_at + k - 1][:, SF["x"]] if (s_at + k) > 0 else f0[:, SF["x"]])
                        for j in sorted(range(n), key=lambda j: -xs_prev[j]):
                            if gi[j, SI["kind"]] != 1:
                                continue  # HDVs solve no QP
                            discrete(rows[j] == qp_rows[q_loc], tr, k, (path, t, k, j, rows[j], qp_rows[q_loc]))
                            got_h = np.array([tr[k, abi.T["QP_H%d" % r], j] for r in range(rows[j])])
                            e = max(abs(tr[k, abi.T["QP_A"], j] - qp_G[q_loc, 0, 0]),
                                    float(np.abs(got_h - qp_h[q_loc, :rows[j]]).max()),
                                    abs(tr[k, abi.T["QP_D"], j] - qp_x[q_loc, 0]))
                            step_mx["qp"] = max(step_mx["qp"], float(e))
                            q_loc += 1
                if "sub_pf" in z.files and is_v1 and shielded:  # control-profile tail: safe_status + min_headway
                    pf = z["sub_pf"][s_at + k]
                    ran_ref = pf[:, 0] > 0
                    st = tr[k, abi.T["STATUS"]]
                    discrete(np.array_equal(~np.isnan(st), ran_ref), tr, k, (path, t, k, "shield ran", st, pf[:, 0]))
                    if ran_ref.any():
                        bits = np.where(ran_ref, np.nan_to_num(st), 0).astype(np.int64)
                        for b_, col_ in ((abi.ST_IS_OPTIMAL, 1), (abi.ST_IS_SAFE, 2), (abi.ST_IS_INVARIANT, 3)):
                            discrete(np.array_equal(((bits & b_) != 0)[ran_ref], pf[ran_ref, col_] > 0), tr, k, (path, t, k, "status", col_))
                        step_mx["state"] = max(step_mx["state"], float(np.abs(tr[k, abi.T["HEADWAY"]][ran_ref] - pf[ran_ref, 4]).max()))
                for name, col in (("X", "x"), ("Y", "y"), ("HEADING", "heading"), ("SPEED", "speed")):
                    step_mx["state"] = max(step_mx["state"], float(np.abs(tr[k, abi.T[name]] - gf[:, SF[col]]).max()))
                cols = [("ACT_STEER", "act_steer"), ("ACT_ACC", "act_acc")]
                if is_v1:
                    cols += [("SAFE_STEER", "safe_steer"), ("SAFE_ACC", "safe_acc")]
                for name, col in cols:
                    step_mx["action"] = max(step_mx["action"], float(np.abs(tr[k, abi.T[name]] - gf[:, SF[col]]).max()))
            if "sub_sa" in z.files and nsub > 0:  # MDPLCVehicle.steering_angle after the step
                got_sa = env.f64[abi.F["STEER_ANGLE"], 0].cpu().numpy()
                step_mx["state"] = max(step_mx["state"], float(np.abs(got_sa - z["sub_sa"][s_at + nsub - 1]).max()))
            o = {k2: v[0].cpu().numpy() for k2, v in out.items()}
            step_mx["obs"] = max(step_mx["obs"], float(np.abs(obs[0, :nc].cpu().numpy() - z["obs"][t]).max()))
            step_mx["reward"] = max(step_mx["reward"], abs(float(o["reward"]) - z["reward"][t]),
                                    float(np.abs(o["agents_rewards"][:nc] - z["agents_rewards"][t]).max()),
                                    float(np.abs(o["regional_rewards"][:nc] - z["regional_rewards"][t]).max()))
            step_mx["info"] = max(step_mx["info"], abs(float(o["average_speed"]) - z["average_speed"][t]),
                                  abs(float(o["traffic_speed"]) - z["traffic_speed"][t]),
                                  abs(float(o["min_headway"]) - z["min_headway"][t]))
            discrete(bool(o["done"]) == bool(z["done"][t]), tr, 2, (path, t, "done"))
            discrete(np.array_equal(o["agents_dones"][:nc].astype(bool), z["agents_dones"][t]), tr, 2, (path, t, "agents_dones"))
            discrete(np.array_equal(o["action_mask"][:nc], z["action_mask"][t]), tr, 2, (path, t, "action_mask"))
            if z["done"][t]:
                assert abs(float(o["merge_percent"]) - z["merge_percent"][t]) <= 1e-9
                assert bool(o["crashed"][:nc].any()) == meta["crashed"]
            for key, v in step_mx.items():
                discrete(v <= tol, tr, 2, (path, t, key, v))
            mx = step_mx
        except _KnifeEdge:
            knife += 1
            assert teacher_forcing, (path, t, "knife-edge flip without teacher forcing: trajectories diverge")
        s_at += nsub
        q_at += nqp
    assert q_at == len(qp_rows), (q_at, len(qp_rows))
    assert knife <= max_knife_edges, (path, "knife-edge decisions", knife)
    env.close()
    err.update(mx)
    err["knife_edges"] = knife
    return err


def free_run(make_env, path):
    """Run a tape's action script free-running (no teacher forcing) and return (steps, crashed, min headway)."""
    z, meta = load_episode(path)
    n = meta["n"] + meta.get("n_hdv", 0)
    kw = env_kwargs(meta)
    kw["trace"] = False
    env = make_env(E=1, N=n, **kw)
    f0, i0 = z["init_f"], z["init_i"]
    env.set_kinematics(f0[None, :, 0], f0[None, :, 1], f0[None, :, 2], f0[None, :, 3], n_merge=np.array([meta["n_merge"]]),
                       kind=i0[None, :, SI["kind"]])
    steps, done, mh = 0, False, float("inf")
    while not done and steps < 100:
        a = torch.tensor(z["actions"][min(steps, len(z["actions"]) - 1)][None], dtype=torch.int32, device=env.device)
        _, _, d, out = env.step(a)
        done = bool(d[0])
        mh = min(mh, float(out["min_headway"][0]))
        steps += 1
    crashed = bool(out["crashed"].any())
    env.close()
    return steps, crashed, mh


def check_safety_layer_probes(make_env, path, tol=1e-9):
    """Stand-alone safety_layer(...) parity: the tape holds reference calls for every controlled vehicle at
    the start of selected steps (tools/gen_golden.py:_probe_safety_layer); compare mm_shield_actions."""
    z, meta = load_episode(path)
    nc, n = meta["n"], meta["n"] + meta.get("n_hdv", 0)
    kw = env_kwargs(meta)
    kw["trace"] = False
    env = make_env(E=1, N=n, **kw)
    f0, i0 = z["init_f"], z["init_i"]
    env.set_kinematics(f0[None, :, 0], f0[None, :, 1], f0[None, :, 2], f0[None, :, 3], n_merge=np.array([meta["n_merge"]]),
                       kind=i0[None, :, SI["kind"]])
    cum = np.concatenate([[0], np.cumsum(z["sub_count"])])
    worst, checked = 0.0, 0
    for k, t in enumerate(z["sl_t"]):
        _force_state(env, z, meta, int(t), int(cum[t]))
        steer = np.zeros((1, n)); acc = np.zeros((1, n))
        steer[0, :nc], acc[0, :nc] = z["sl_act"][k, :, 0], z["sl_act"][k, :, 1]
        s_s, s_a, st, mg, _hw = env.shield_actions(steer, acc)
        s_s, s_a, st, mg = s_s[0, :nc].cpu().numpy(), s_a[0, :nc].cpu().numpy(), st[0, :nc].cpu().numpy(), mg[0, :nc].cpu().numpy()
        ref_st = z["sl_status"][k]
        for j in range(nc):
            got = [bool(st[j] & b) for b in (abi.ST_IS_OPTIMAL, abi.ST_IS_SAFE, abi.ST_IS_INVARIANT, abi.ST_IS_LC_SAFE,
                                             abi.ST_IS_COLLABORATING, abi.ST_COLLABORATE_ADJ)]
            assert st[j] & abi.ST_RAN
            if got != [bool(x) for x in ref_st[j]] or abs(s_s[j] - z["sl_safe"][k, j, 0]) > tol:
                assert abs(mg[j]) < KNIFE_EDGE, (path, t, j, got, ref_st[j], mg[j])  # only the LC knife-edge may differ
                continue
            worst = max(worst, abs(s_s[j] - z["sl_safe"][k, j, 0]), abs(s_a[j] - z["sl_safe"][k, j, 1]))
            checked += 1
    assert checked >= 0.9 * len(z["sl_t"]) * nc and worst <= tol, (checked, worst)
    env.close()
    return worst, checked
