#!/usr/bin/env python3
"""Design tool: how many interior-point iterations does a WAVE have to sit through per policy step?

Runs the bench workload (MASS, qp_solver=ipm, stationary batch) on the CPU oracle's logging build
(oracle/libmm_oracle_qplog.so: one line per shield QP with its iteration count, status and the vehicles whose
decision of the same sub-step its right-hand side reads) and replays the log under different wave schedules:

  literal    the round-2 kernel: one sweep stage per rank, a stage lasts as long as its slowest QP in the wave
  chained    dependency-driven: a QP starts when the decisions it reads are final; the wave leaves the sub-step when
             its slowest env's chain is done (one shared wave-wide iteration loop)
  ideal      lane-iterations / 64 (perfect packing, no dependencies): the floor

Output: trips (= executions of the iteration body) per wave and policy step for each schedule, the iteration
histogram, the share of capped ("unknown") QPs and the chain-depth distribution.  Test infrastructure only.
"""
import argparse
import collections
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import torch  # noqa: E402


def run_log(E, N, steps, path, shield="cbf-cav"):
    import oracle_env
    from marl_mass_amd import _cabi as abi
    from marl_mass_amd.vec_env import BatchedMergeEnv
    lib_path = os.path.join(REPO, "oracle", "libmm_oracle_qplog.so")
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "libmm_oracle_qplog.so"], stdout=subprocess.DEVNULL)
    clib = abi.CLib(lib_path)
    clib.lib.orc_set_threads.argtypes = [C.c_int32]
    clib.lib.orc_set_threads(1)  # the log is written by one thread
    clib.lib.orc_set_math.argtypes = [C.c_int32]
    clib.lib.orc_set_math(1)
    clib.lib.orc_qplog_open.argtypes = [C.c_char_p]
    clib.lib.orc_qplog_open.restype = None
    cfg = {"safety_guarantee": shield, "HEADWAY_TIME": 0.5}
    env = BatchedMergeEnv(clib, E, N, env_id="merge-multi-agent-v1", config=cfg, device="cpu", cbf_eta=0.03125,
                          cbf_tau=0.5, seed=1000, auto_reset=True, qp_solver="ipm")
    env.reset()
    g = torch.Generator().manual_seed(123)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1])
    ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(16)]
    T = env.T
    ge = torch.arange(0, E, dtype=torch.int64)
    env.env_i32[abi.EP["STEPS"]] = ((ge * 37) % T).to(torch.int32)
    for t in range(T):
        env.step(ring[t % 16])
    clib.lib.orc_qplog_open(path.encode())
    for t in range(steps):
        env.step(ring[t % 16])
    clib.lib.orc_qplog_open(None)


def analyse(path, E, N, steps, G=8):
    # (env, launch-local sub-step) -> list of (veh, rank, iters, optimal, rows, dep_ol, dep_oa)
    per = collections.defaultdict(list)
    hist = collections.Counter()
    capped = total = 0
    # the log has no launch index: sub-step clocks restart at an auto-reset, so count launches per env by order of appearance
    seen = collections.defaultdict(int)   # env -> number of lines so far grouped into (launch, sub) by time changes
    last_time = {}
    sub_idx = {}
    for ln in open(path):
        e, tm, i, r, it, opt, rows, dol, doa = map(int, ln.split())
        if last_time.get(e) != tm:
            last_time[e] = tm
            sub_idx[e] = sub_idx.get(e, -1) + 1
        per[(e, sub_idx[e])].append((i, r, it, opt, rows, dol, doa))
        hist[it] += 1
        total += 1
        capped += 0 if opt else 1
    envs_per_wave = 64 // G
    n_sub = max(s for (_, s) in per) + 1
    waves = E // envs_per_wave

    def chained(extra, cap=None):
        """Trips per wave and policy step when a QP starts as soon as the decisions it reads are final.  extra = 1: the
        follower starts in the trip AFTER its input stopped; extra = 0 (what the kernel does): in the same trip.
        cap: pretend no QP runs longer than `cap` iterations (isolates what the iteration-capped QPs cost)."""
        tot = 0
        depth_hist.clear()
        for w in range(waves):
            for s in range(n_sub):
                worst = 0
                for e in range(w * envs_per_wave, (w + 1) * envs_per_wave):
                    done_at, depth = {}, {}
                    for (i, r, it, opt, rows, dol, doa) in sorted(per.get((e, s), ()), key=lambda q: q[1]):
                        it = min(it, cap) if cap else it
                        start = max([done_at.get(d, 0) for d in (dol, doa) if d >= 0] + [0])
                        done_at[i] = start + it + extra
                        depth[i] = max([depth.get(d, 0) for d in (dol, doa) if d >= 0] + [0]) + 1
                    if done_at:
                        worst = max(worst, max(done_at.values()) + (1 - extra))  # (+ the trip of the last stopping test)
                        depth_hist[max(depth.values())] += 1
                tot += worst
        return tot / (waves * steps)

    depth_hist = collections.Counter()
    lit = 0
    for w in range(waves):
        for s in range(n_sub):
            # literal: stage r lasts max over the wave's envs of that rank's iterations
            stage = collections.defaultdict(int)
            for e in range(w * envs_per_wave, (w + 1) * envs_per_wave):
                for (i, r, it, *_rest) in per.get((e, s), ()):
                    stage[r] = max(stage[r], it + 1)
            lit += sum(stage.values())
    trips = {"literal_sweep": lit / (waves * steps), "chained_next_trip": chained(1), "chained_same_trip": chained(0),
             "chained_same_trip_if_no_qp_ran_past_10_iterations": chained(0, cap=10)}
    lane_iters = sum(k * v for k, v in hist.items())
    trips["ideal_packed"] = lane_iters / 64 / (waves * steps)
    out = {
        "workload": "%d envs x %d CAVs, MASS, ipm, %d policy steps (stationary batch)" % (E, N, steps),
        "qps": total, "qps_per_env_step": total / (E * steps), "capped_share": capped / total,
        "iters_mean": lane_iters / total, "iters_hist": {str(k): hist[k] for k in sorted(hist)},
        "chain_depth_hist_per_env_substep": {str(k): depth_hist[k] for k in sorted(depth_hist)},
        "trips_per_wave_step": trips,
        "reading": "a trip = one execution of the interior-point body by a wave of 8 envs; the kernel runs chained_same_trip; "
                   "ideal_packed = lane-iterations / 64 is what perfect packing without dependencies would need",
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "qp.log")
        run_log(args.envs, args.agents, args.steps, path)
        res = analyse(path, args.envs, args.agents, args.steps)
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
