#!/usr/bin/env python3
"""GPU box: would the split interior-point step gain from stepping the batch as several env partitions on concurrent streams?

The sweep kernel's launch lasts as long as its slowest wave (one lane per env, one wave per SIMD); SIMDs whose wave is through
sit idle until the next phase kernel.  Partitions on their own streams drift apart and fill each other's tails.  This probe
needs no kernel change: P independent VecMergeEnv handles of E / P envs each (global env ids and seeds of the one big batch,
`first_env`), stepped (a) one after the other on one stream, (b) each on its own stream, against (c) the one-handle batch.

    python tools/split_streams_probe.py [--envs 65536] [--agents 8] [--parts 1 2 4 8] [--shield mass] -> gpurun_out/split_streams.json
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from marl_mass_amd import VecMergeEnv, _cabi as abi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--shield", default="mass")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = "cuda:0"
    safety = {"mass": "cbf-cav", "hss": "cbf-avs_cint"}[args.shield]
    cfg = {"safety_guarantee": safety, "HEADWAY_TIME": 0.5}
    kw = dict(cbf_eta=0.03125, cbf_tau=0.5, seed=1000, auto_reset=True, qp_solver="ipm", debug_flags=8)  # (split step at any size)
    E, N = args.envs, args.agents
    g = torch.Generator(device=dev).manual_seed(123)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device=dev)
    ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(16)]
    out = {"workload": "%d envs x %d CAVs, %s, qp_solver=ipm, split step forced" % (E, N, safety), "steps": args.steps, "rows": []}
    for P in args.parts:
        per = (E + P - 1) // P
        envs, rings, streams = [], [], []
        for i in range(P):
            e0, e1 = i * per, min(E, (i + 1) * per)
            env = VecMergeEnv(e1 - e0, N, env_id="merge-multi-agent-v1", config=cfg, device=dev, first_env=e0,
                              skip_outputs=("agents_info", "action_mask", "crashed"), **kw)
            env.enable_metrics(deferred=True)
            env.reset()
            ge = torch.arange(e0, e1, dtype=torch.int64, device=dev)
            env.env_i32[abi.EP["STEPS"]] = ((ge * 37) % env.T).to(torch.int32)
            envs.append(env)
            rings.append([r[e0:e1].contiguous() for r in ring])
            streams.append(torch.cuda.Stream(device=dev))
        for t in range(envs[0].T + 3):
            for env, r in zip(envs, rings):
                env.step(r[t % 16])
        torch.cuda.synchronize()
        res = {}
        for how in ("one stream", "own streams"):
            if how == "own streams" and P == 1:
                continue
            t0 = time.perf_counter()
            for t in range(args.steps):
                for env, r, s in zip(envs, rings, streams):
                    if how == "own streams":
                        with torch.cuda.stream(s):
                            env.step(r[t % 16])
                    else:
                        env.step(r[t % 16])
            torch.cuda.synchronize()
            res[how] = (time.perf_counter() - t0) / args.steps * 1e3
        for env in envs:
            env.poll_errors()
        out["rows"].append({"parts": P, "envs_per_part": per, "ms_per_step": res})
        print(P, res, flush=True)
        del envs, rings, streams
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "split_streams.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
