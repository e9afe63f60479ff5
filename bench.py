#!/usr/bin/env python3
"""Headline benchmark: agent-steps/sec of env.step with the MASS CBF shield on (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one fused HIP launch = one policy step (3 simulation sub-steps + rewards + obs +
auto-reset) of EVERY env of the rank.  Workload (config.workload): 65 536 envs x 8 CAVs per GPU,
merge-multi-agent-v1, safety_guarantee=cbf-cav (MASS), eta=0.03125, tau=0.5, synthetic episodes
from the device RNG and a pre-generated categorical action tape.  Envs are independent,
so ranks shard the batch with no data-path collective ("scaling": "weak", per-GPU work fixed);
the only RCCL traffic is the 8-double metric all-reduce after the timed region.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SHIELDS = {"mass": "cbf-cav", "hss": "cbf-avs_cint", "none": "none"}


def algorithmic_bytes_per_agent_step(env_id, N):
    """B_alg = 2S + 4*n_s + 21 + 41/N with S = 44 B (v0) / 53 B (v1), n_s = 25 / 30."""
    S, n_s = (53, 30) if env_id.endswith("v1") else (44, 25)
    return 2 * S + 4 * n_s + 21 + 41.0 / N


def cpu_baseline(args, env_id, cfg, kw):
    """Times the CPU oracle (the C restatement, `kind: port`) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_env
    threads = oracle_env.library().lib.orc_set_threads(args.cpu_threads or min(16, os.cpu_count() or 1))
    E, N, K = args.cpu_envs, args.agents, args.cpu_steps
    env = oracle_env.OracleEnv(E, N, env_id=env_id, config=cfg, **kw)
    env.reset()
    g = torch.Generator().manual_seed(123)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1])
    acts = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(8)]
    env.step(acts[0])
    t0 = time.perf_counter()
    for t in range(K):
        env.step(acts[t % 8])
    dt = time.perf_counter() - t0
    return {"value": E * N * K / dt, "unit": "agent-steps/s", "cores": int(threads), "kind": "port",
            "sample": "%d envs x %d CAVs x %d steps, same config, OpenMP over envs (%.1f s)" % (E, N, K, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs", type=int, default=65536, help="envs PER GPU")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--shield", choices=sorted(SHIELDS), default="mass")
    ap.add_argument("--env-id", default="merge-multi-agent-v1")
    ap.add_argument("--obs-f64", action="store_true")
    ap.add_argument("--hdv", type=int, default=0, help="mixed traffic: the last HDV of the --agents vehicles are IDM/MOBIL HDVs (not the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=32768)
    ap.add_argument("--cpu-steps", type=int, default=400)
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    here = os.path.dirname(os.path.abspath(__file__))
    if rank == 0 and not os.path.exists(os.path.join(here, "marl-mass_amd", "csrc", "libmm_hip.so")):
        import __graft_entry__  # fresh checkout: the library is a build artefact
        __graft_entry__.build()
    if world > 1:
        dist.barrier(device_ids=[local])
    from marl_mass_amd import VecMergeEnv, reduce_rollout_metrics
    E, N = args.envs, args.agents
    cfg = {"safety_guarantee": SHIELDS[args.shield], "HEADWAY_TIME": 0.5 if args.shield != "none" else 1.2}
    kw = dict(cbf_eta=0.03125 if args.shield != "none" else 0.0, cbf_tau=cfg["HEADWAY_TIME"], seed=1000,
              auto_reset=True, obs_f64=args.obs_f64, n_hdv=args.hdv)
    env = VecMergeEnv(E, N, env_id=args.env_id, config=cfg, device=dev, first_env=rank * E, **kw)
    metrics = env.enable_metrics()
    env.reset()
    g = torch.Generator(device=dev).manual_seed(123 + rank)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device=dev)
    ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(16)]

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local])

    for t in range(args.warmup):
        env.step(ring[t % 16])
    torch.cuda.synchronize()
    metrics.zero_()
    metrics[7] = float("inf")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        ev[t][0].record()
        env.step(ring[t % 16])  # one mm_step launch on torch's current stream
        ev[t][1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps  # HIP events on the launch stream

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # end-of-rollout metric reduction: the only collective of the path
        reduce_rollout_metrics(metrics)
    elapsed = float(tmax[0])
    m = metrics.cpu().tolist()

    if rank == 0:
        agent_steps = float(E) * N * args.steps * world
        b_alg = algorithmic_bytes_per_agent_step(args.env_id, N)
        achieved = E * N * b_alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (profiles/traffic.json, written from
        # tools/profile.sh on the same workload); null for any other workload
        traffic = None
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
            w = tj["workload"]
            if (w["envs_per_gpu"], w["agents"], w["shield"], w["env_id"]) == (E, N, args.shield, args.env_id):
                traffic = tj["bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        # secondary reading (the kernel is VALU-issue bound): instructions from the committed
        # SQ_INSTS_VALU pass x 4 issue cycles per wave64 instruction, against 1024 SIMDs at the 2.4 GHz peak clock
        valu = None
        try:
            sj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "step_kernel_summary.json")))
            if traffic is not None:
                n_valu = sj["pmc_per_launch"]["SQ_INSTS_VALU"] / sj["pmc_per_launch"]["SQ_WAVES"] * (E * 8 // 64 if N <= 8 else E * 16 // 64)
                ach = n_valu * 4 / (kern_ms * 1e-3) / 1e12
                valu = {"bound": "valu-issue", "achieved": ach, "peak": 1024 * 2.4e9 / 1e12, "unit": "T SIMD-cycles/s",
                        "frac": ach / (1024 * 2.4e9 / 1e12), "valu_insts_per_wave": sj["pmc_per_launch"]["SQ_INSTS_VALU"] / sj["pmc_per_launch"]["SQ_WAVES"]}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
        line = {
            "metric": "agent-steps/sec (whole node), MASS CBF shield on, 65536 envs x 8 CAVs" if (args.shield == "mass" and not args.hdv)
                      else "agent-steps/sec (whole node), shield=%s" % args.shield,
            "value": agent_steps / elapsed, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs x %d CAVs per GPU, %s, safety_guarantee=%s, eta=0.03125, tau=%.1f, "
                                   "100-step episodes with auto-reset, categorical action tape%s"
                                   % (E, N, args.env_id, cfg["safety_guarantee"], cfg["HEADWAY_TIME"],
                                      (", of which %d HDVs per env" % args.hdv) if args.hdv else ""),
                       "envs_per_gpu": E, "agents": N, "obs_dtype": "f64" if args.obs_f64 else "f32",
                       "parallelism": "env-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/)",
                         "alg_bytes_per_launch": E * N * b_alg,
                         "kernel": "step_kernel", "kernel_ms": kern_ms, "alg_bytes_per_agent_step": b_alg},
            "rollout_metrics": {"mean_reward": m[0] / max(m[4], 1), "crashed_episodes": m[1],
                                "mean_speed": m[2] / max(m[4], 1), "env_steps": m[4],
                                "mean_merge_percent": m[5] / max(m[6], 1), "episodes": m[6], "min_headway": m[7]},
        }
        if valu is not None:
            line["roofline_secondary"] = valu
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(args, args.env_id, cfg, kw)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
