#!/usr/bin/env python3
"""Randomised differential soak, GPU box only: random points of the configuration space (env kind, shield, QP solver,
N, HDV count, reward variant, lateral control, action masking, re-drawn vehicle counts, eta / tau, action
distribution, obs dtype, trace on / off), each run free with auto-reset on the HIP library and on the oracle (math mode 1)
with the same seeds and action tape; every byte of state, observation, reward, done and info must agree at every step.
Test infrastructure (it drives the oracle): never part of the product path.

    python tools/fuzz_parity.py [--cases 120] [--seed 1] [--budget-s 600]  -> gpurun_out/fuzz_parity.json
"""
import argparse
import json
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import oracle_env  # noqa: E402
from marl_mass_amd import VecMergeEnv  # noqa: E402


def draw_case(rng):
    v1 = rng.random() < 0.85
    safety = rng.choice(["none", "cbf-avs_cint", "cbf-avs", "cbf-cav", "cbf-cav"]) if v1 else "none"
    shielded = safety != "none"
    draw_counts = rng.random() < 0.15
    if draw_counts:
        density = rng.choice([1, 2, 3])
        N = {1: 6, 2: 8, 3: 11}[density] + rng.choice([0, 0, 1])
        mixed = rng.choice([False, True, True, "av"]) if v1 else (rng.random() < 0.5)   # "av": one CAV among HDVs (v1 only)
        n_hdv = 0  # (the counts are drawn per episode)
    else:
        density, mixed = 0, False
        N = rng.choice([2, 3, 4, 5, 6, 7, 8, 8, 8, 9, 11, 12])
        n_hdv = rng.choice([0, 0, 0, 1, 2, 3, N // 2]) if N >= 3 else 0
        n_hdv = min(n_hdv, N - 1)
        # 6 spawn slots per road (merge_env_v1.py:284-285)
        n_cav = N - n_hdv
        while (n_cav - n_cav // 2) + (n_hdv - n_hdv // 2) > 6 or (n_cav // 2) + (n_hdv // 2) > 6:
            n_hdv = max(0, n_hdv - 1); n_cav = N - n_hdv
            if n_hdv == 0 and N > 12: N = 12; n_cav = 12
    cfg = {"safety_guarantee": safety, "HEADWAY_TIME": rng.choice([0.5, 0.5, 1.2]),
           "agent_reward": rng.choice(["default", "default", "srew", "mrew"]) if v1 else "default",
           "lateral_control": rng.choice(["steer", "steer", "steer", "steer_vel"]) if v1 else "steer",
           "action_masking": rng.random() < 0.3}
    if draw_counts:
        cfg["traffic_density"] = density
        cfg["traffic_type"] = mixed if isinstance(mixed, str) else ("mixed" if mixed else "cav")
        cfg["mixed_traffic"] = bool(mixed)
    kw = dict(env_id="merge-multi-agent-v1" if v1 else "merge-multi-agent-v0", config=cfg,
              cbf_eta=rng.choice([0.03125, 0.03125, 0.5, 0.1]) if shielded else 0.0, cbf_tau=cfg["HEADWAY_TIME"],
              obs_f64=rng.random() < 0.5, seed=rng.randrange(1, 1 << 30), auto_reset=True, n_hdv=n_hdv,
              qp_solver="ipm" if (shielded and rng.random() < float(os.environ.get("MM_FUZZ_IPM_P", "0.3"))) else "exact", trace=rng.random() < 0.3,
              draw_counts=draw_counts)
    # (odd batch sizes: the last wave of the launch is partly empty and an env group may be the only one in its wave)
    E = rng.choice([64, 128, 256, 512, 1, 7, 37, 100, 333]) if kw["qp_solver"] == "exact" else rng.choice([32, 64, 128, 5, 45])
    steps = rng.choice([40, 80, 120, 210])
    probs = rng.choice([[0.1, 0.6, 0.1, 0.1, 0.1], [0.25, 0.3, 0.25, 0.1, 0.1], [0.2, 0.2, 0.2, 0.2, 0.2], [0.05, 0.1, 0.05, 0.6, 0.2],
                        [0.05, 0.1, 0.05, 0.1, 0.7]])
    return E, N, steps, probs, kw


def run_case(E, N, steps, probs, kw):
    gpu = VecMergeEnv(E, N, device="cuda:0", debug_flags=int(os.environ.get("MM_DEBUG_FLAGS", "0")), **kw)
    cpu = oracle_env.OracleEnv(E, N, **kw)
    og, ag = gpu.reset()
    oc, ac = cpu.reset()
    if not (torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(og.cpu(), oc) and torch.equal(ag.cpu(), ac)):
        return "reset differs"
    g = torch.Generator().manual_seed(kw["seed"] ^ 0x5EED)
    p = torch.tensor(probs)
    for t in range(steps):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        if not torch.equal(gpu.u8.cpu(), cpu.u8):
            return "step %d: discrete state" % t
        if not torch.equal(gpu.env_i32.cpu(), cpu.env_i32):
            return "step %d: episode counters" % t
        if not torch.equal(gpu.f64.cpu().nan_to_num(nan=-7.0), cpu.f64.nan_to_num(nan=-7.0)):
            return "step %d: float state" % t
        if not (torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc)):
            return "step %d: obs / reward / done" % t
        for k in ic:
            x, y = ig[k].cpu(), ic[k]
            if x.is_floating_point():
                x, y = x.nan_to_num(nan=-7.0), y.nan_to_num(nan=-7.0)
            if not torch.equal(x, y):
                return "step %d: info[%s]" % (t, k)
        if kw["trace"] and not torch.equal(gpu.trace.cpu().nan_to_num(nan=-7.0), cpu.trace.nan_to_num(nan=-7.0)):
            return "step %d: trace" % t
        if t % 9 == 8 and kw["env_id"].endswith("v1") and kw["config"]["safety_guarantee"] != "none":
            # the stand-alone shield entry (safety_layer(...) on the current state, nothing is mutated)
            steer = (torch.rand(E, N, dtype=torch.float64, generator=g) - 0.5) * 0.3
            acc = (torch.rand(E, N, dtype=torch.float64, generator=g) - 0.5) * 12
            for x, y in zip(gpu.shield_actions(steer, acc), cpu.shield_actions(steer, acc)):
                if not torch.equal(x.cpu().nan_to_num(nan=-7.0) if x.is_floating_point() else x.cpu(),
                                   y.nan_to_num(nan=-7.0) if y.is_floating_point() else y):
                    return "step %d: shield_actions" % t
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=600.0)
    args = ap.parse_args()
    oracle_env.set_math_mode(1)
    oracle_env.library().lib.orc_set_threads(min(16, os.cpu_count() or 1))
    rng = random.Random(args.seed)
    t0 = time.time()
    done, failures, agent_steps = 0, [], 0
    for c in range(args.cases):
        if time.time() - t0 > args.budget_s:
            break
        E, N, steps, probs, kw = draw_case(rng)
        try:
            err = run_case(E, N, steps, probs, kw)
        except Exception as ex:  # a refused configuration must be refused by BOTH backends: run_case builds the GPU env first
            err = "exception: %r" % (ex,)
        done += 1
        agent_steps += E * N * steps
        if err:
            failures.append({"case": c, "E": E, "N": N, "steps": steps, "probs": probs, "kw": {k: v for k, v in kw.items()}, "error": err})
            print("FAIL case %d: %s  %s" % (c, err, json.dumps({"E": E, "N": N, "steps": steps, "kw": kw})), flush=True)
        elif c % 10 == 0:
            print("case %d ok (%d envs x %d x %d steps, %s, %s, hdv %d) %.0f s" % (c, E, N, steps, kw["config"]["safety_guarantee"],
                                                                               kw["qp_solver"], kw["n_hdv"], time.time() - t0), flush=True)
    out = {"seed": args.seed, "cases_run": done, "agent_steps": agent_steps, "failures": failures, "seconds": time.time() - t0}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "fuzz_parity.json"), "w"), indent=1)
    print("%d cases, %d agent-steps, %d failures, %.0f s" % (done, agent_steps, len(failures), out["seconds"]))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
