#!/usr/bin/env python3
"""How far does the production QP mode (exact KKT point) drift from the fidelity mode (the iterate cvxopt's
interior-point algorithm stops at) over FREE-RUNNING episodes?  ("next round" item 1.ii)

Runs the same seeded batch of episodes twice on the CPU oracle (test infrastructure; this is a measurement tool,
not a product path) -- qp_solver="exact" vs "ipm", same spawns, same action tape -- and reports, per policy step,
the largest state deviation and every discrete decision that differs: crashed / done flags, lane, target lane,
is_lc_safe, is_collaborating, collaborate_adj.  Also the per-QP statistics over the recorded (G, h) of the tapes.

    python tools/qp_fidelity.py [E] [N] -> profiles/r02/qp_fidelity.json
"""
import glob
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")]
import oracle_env  # noqa: E402
from marl_mass_amd import _cabi as abi  # noqa: E402


def free_run(shield, N, E, steps=100, p=(0.1, 0.6, 0.1, 0.1, 0.1), n_hdv=0):
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": shield, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125,
              cbf_tau=0.5, seed=1000, auto_reset=False, n_hdv=n_hdv)
    ex, ip = oracle_env.OracleEnv(E, N, qp_solver="exact", **kw), oracle_env.OracleEnv(E, N, qp_solver="ipm", **kw)
    ex.reset(); ip.reset()
    g = torch.Generator().manual_seed(123)
    pt = torch.tensor(p)
    F, B = abi.F, abi.B
    out = dict(shield=shield, N=N, n_hdv=n_hdv, E=E, steps=steps, action_p=list(p))
    # An env counts as DIVERGED from the step at which either a discrete quantity differs or a float moved by more than
    # JUMP (a lane-change veto taken in one run only inside a sub-step shows up as a steering jump before any flag differs).
    JUMP = {"x": 5e-2, "y": 5e-3, "heading": 1e-3, "speed": 5e-2}
    dev = {k: [] for k in JUMP}
    first = {k: 0 for k in ("lane", "target_lane", "crashed", "is_lc_safe", "is_collaborating", "collaborate_adj", "done", "float_jump")}
    diverged = torch.zeros(E, dtype=torch.bool)
    alive = torch.ones(E, dtype=torch.bool)           # both runs still in the episode
    crash = [0, 0]
    diverged_by_step = []
    for t in range(steps):
        a = torch.multinomial(pt, E * N, True, generator=g).view(E, N).int()
        _, _, d_ex, i_ex = ex.step(a)
        _, _, d_ip, i_ip = ip.step(a)
        live = alive[:, None].expand(E, N)
        new_div = torch.zeros(E, dtype=torch.bool)
        kinds = {}
        for name, plane in (("lane", "LANE"), ("target_lane", "TARGET_LANE"), ("crashed", "CRASHED")):
            kinds[name] = ((ex.u8[B[plane]] != ip.u8[B[plane]]) & live).any(1)
        fe, fi = ex.u8[B["FLAGS"]], ip.u8[B["FLAGS"]]
        for name, bit in (("is_lc_safe", abi.FLAG_IS_LC_SAFE), ("is_collaborating", abi.FLAG_IS_COLLABORATING),
                          ("collaborate_adj", abi.FLAG_COLLABORATE_ADJ)):
            kinds[name] = ((((fe & bit) != 0) != ((fi & bit) != 0)) & live).any(1)
        kinds["done"] = (d_ex != d_ip) & alive
        dd = {name: ((ex.f64[F[plane]] - ip.f64[F[plane]]).abs() * live) for name, plane in
              (("x", "X"), ("y", "Y"), ("heading", "HEADING"), ("speed", "SPEED"))}
        kinds["float_jump"] = torch.zeros(E, dtype=torch.bool)
        for name in JUMP:
            kinds["float_jump"] |= (dd[name] > JUMP[name]).any(1)
        for name in first:  # attribute a newly diverged env to the first kind that shows (discrete kinds first)
            hit = kinds[name] & ~diverged & ~new_div
            first[name] += int(hit.sum())
            new_div |= hit
        diverged |= new_div
        ok = live & ~diverged[:, None]
        for name in JUMP:
            dev[name].append(float(dd[name][ok].max()) if ok.any() else 0.0)
        diverged_by_step.append(int(diverged.sum()))
        crash[0] += int((i_ex["crashed"].any(1) & d_ex.bool() & alive).sum()); crash[1] += int((i_ip["crashed"].any(1) & d_ip.bool() & alive).sum())
        alive &= ~(d_ex.bool() | d_ip.bool())
    out["max_abs_deviation_while_decisions_agree"] = {k: max(v) for k, v in dev.items()}
    out["deviation_x_by_step"] = [float("%.3g" % v) for v in dev["x"]]
    out["episodes_diverged_by_step"] = diverged_by_step
    out["first_difference_by_kind"] = first
    out["episodes_diverged"] = int(diverged.sum())
    out["crashed_episodes_exact_vs_ipm"] = crash
    ex.close(); ip.close()
    return out


def per_qp_stats():
    xs_e, xs_i, st, it, rows = [], [], [], [], []
    for f in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "*_*.npz"))):
        z = np.load(f)
        if "qp_rows" not in z.files or len(z["qp_rows"]) == 0:
            continue
        ipm = json.loads(str(z["meta"]))["qp_solver"].startswith("coneqp")
        xs_i.append(z["qp_x"] if ipm else z["qp_x_alt"]); xs_e.append(z["qp_x_alt"] if ipm else z["qp_x"])
        st.append(z["qp_status"]); it.append(z["qp_iters"]); rows.append(z["qp_rows"])
    xe, xi, st, it, rows = map(np.concatenate, (xs_e, xs_i, st, it, rows))
    dd, ds = np.abs(xi[:, 0] - xe[:, 0]), np.abs(xi[:, 2] - xe[:, 2])
    opt, inactive, slack = st == 1, xe[:, 0] == 0, xe[:, 2] > 0
    return dict(n_qp=int(len(st)), four_row=int((rows == 4).sum()), status_optimal=int(opt.sum()), status_unknown=int((~opt).sum()),
                unknown_all_have_active_slack=bool((xe[~opt, 2] > 0).all()), slack_active=int(slack.sum()),
                iterations_hist={str(k): int(v) for k, v in enumerate(np.bincount(it)) if v},
                max_abs_dd=dict(all=float(dd.max()), constraint_inactive=float(dd[inactive].max()),
                                constraint_active=float(dd[~inactive & ~slack].max()), slack_active=float(dd[slack].max())),
                max_abs_ds=float(ds.max()), median_abs_dd=float(np.median(dd)), p99_abs_dd=float(np.percentile(dd, 99)),
                max_abs_d_safe_acc=float(dd.max() * 15), note="d_safe_acc = dd / dt with dt = 1/15 s (derived_acceleration, decentral_layer.py:80-82)")


if __name__ == "__main__":
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    oracle_env.library().lib.orc_set_threads(8)
    res = dict(per_qp=per_qp_stats(), free_running=[])
    for shield, N, n_hdv in (("cbf-cav", 8, 0), ("cbf-avs_cint", 4, 0), ("cbf-cav", 4, 0), ("cbf-cav", 7, 3)):
        r = free_run(shield, N, E, n_hdv=n_hdv)
        print(json.dumps({k: v for k, v in r.items() if k not in ("deviation_x_by_step", "episodes_diverged_by_step")}))
        res["free_running"].append(r)
    lc = free_run("cbf-cav", 8, E, p=(0.3, 0.2, 0.3, 0.1, 0.1))
    print(json.dumps({k: v for k, v in lc.items() if k not in ("deviation_x_by_step", "episodes_diverged_by_step")}))
    res["free_running"].append(lc)
    os.makedirs(os.path.join(REPO, "profiles", "r02"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", "r02", "qp_fidelity.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res["per_qp"], indent=1))
