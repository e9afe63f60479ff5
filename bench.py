#!/usr/bin/env python3
"""Headline benchmark: agent-steps/sec of env.step with the MASS CBF shield on (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one fused HIP launch = one policy step (3 simulation sub-steps + rewards + obs +
auto-reset) of EVERY env of the rank.  Workload (config.workload): 65 536 envs x 8 CAVs,
merge-multi-agent-v1, safety_guarantee=cbf-cav (MASS), eta=0.03125, tau=0.5, synthetic episodes
from the device RNG and a pre-generated categorical action tape.  Envs are independent,
so ranks shard the batch with no data-path collective; the only RCCL traffic is the 8-double metric
all-reduce after the timed region.

Scaling modes (the line says which one ran, in `scaling`, `metric` and `config.workload`):
  --scaling strong (default) --total-envs (65 536) envs are split over the ranks: BASELINE.json's metric on its own
                   configuration at every N -- one GPU holds the whole batch, 8 ranks are config 5 exactly (8 192 envs
                   per GPU); total work fixed.  With N > 1 the line also carries `weak_scaling`: the same kernel with
                   --envs (65 536) envs on EVERY rank, timed right after the headline region;
  --scaling weak   every rank holds --envs (65 536) envs (BASELINE's batch replicated per GPU) as the headline value.

The batch is made STATIONARY before anything is timed: episode phases are staggered over the batch
(env e starts at step (37 e) mod 100) and one untimed episode length is rolled, so every timed window --
however short -- samples all phases of an episode incl. the merge zone and the in-kernel auto-reset.
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
QP_MODE_NOTE = {"exact": "closed-form KKT point of the shield QP: outside north_star's 1e-5 against the reference's interior-point "
                         "iterate, see config.tolerance",
                "ipm": "cvxopt's interior-point iterate, the reference's own solver behaviour: inside north_star's 1e-5"}
SHIELDS = {"mass": "cbf-cav", "hss": "cbf-avs_cint", "none": "none"}


def algorithmic_bytes_per_agent_step(env_id, N):
    """B_alg = 2S + 4*n_s + 21 + 41/N with S = 44 B (v0) / 53 B (v1), n_s = 25 / 30."""
    S, n_s = (53, 30) if env_id.endswith("v1") else (44, 25)
    return 2 * S + 4 * n_s + 21 + 41.0 / N


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    out = {"value": E * N * K / dt, "unit": "agent-steps/s", "cores": int(threads), "kind": "port",
           "sample": "%d envs x %d CAVs x %d steps, same config, OpenMP over envs (%.1f s)" % (E, N, K, dt),
           "host": {"cpu_model": _cpu_model(), "logical_cpus": os.cpu_count()}}
    # the same port on ONE thread (SURVEY 8d asks for both): a smaller sample of the same workload
    oracle_env.library().lib.orc_set_threads(1)
    E1, K1 = max(E // 16, 64), max(K // 4, 20)
    env1 = oracle_env.OracleEnv(E1, N, env_id=env_id, config=cfg, **kw)
    env1.reset()
    env1.step(acts[0][:E1].contiguous())
    t0 = time.perf_counter()
    for t in range(K1):
        env1.step(acts[t % 8][:E1].contiguous())
    dt1 = time.perf_counter() - t0
    out["single_thread"] = {"value": E1 * N * K1 / dt1, "unit": "agent-steps/s", "cores": 1,
                            "sample": "%d envs x %d CAVs x %d steps (%.1f s)" % (E1, N, K1, dt1)}
    oracle_env.library().lib.orc_set_threads(int(threads))
    # the reference's OWN Python path cannot travel to this box; its timing is regenerated in the build container by
    # tools/time_reference.py and committed as profiles/reference_cpu.json (host, cores and stand-in caveats inside)
    try:
        ref = json.load(open(os.path.join(REPO, "profiles", "reference_cpu.json")))
        row = [r for r in ref["rows"] if r["safety_guarantee"] == cfg["safety_guarantee"] and r["n_cav"] == N and r["qp_stand_in"] == "exact"]
        if row:
            out["reference_python"] = {"value": row[0]["agent_steps_per_s"], "unit": "agent-steps/s", "cores": 1,
                                       "ms_per_env_step": row[0]["ms_per_env_step"], "host": ref["host"]["cpu_model"],
                                       "where": "build container (not this box), tools/time_reference.py -> profiles/reference_cpu.json",
                                       "caveat": "reference imported under stand-ins; QP by the closed form (cvxopt unavailable)"}
    except (OSError, KeyError, ValueError):
        pass
    return out


def ipm_mode_line(args, dev, E, N, cfg, kw, ring, steps=12):
    """The headline workload once more with qp_solver="ipm": every shield QP through the interior-point iteration the
    reference's cvxopt call runs (bit-identical to the reference-side restatement of coneqp), same stationary batch, a short
    timed window with per-launch HIP events.  Reported beside the headline as a first-class mode with its own roofline
    block: it is the mode that meets north_star's 1e-5 tolerance against the interior-point iterate; the exact mode is the
    true minimiser of the same QP and differs from that iterate by up to 3e-4 m/s (profiles/r02/qp_fidelity.json)."""
    from marl_mass_amd import VecMergeEnv, _cabi as abi
    env = VecMergeEnv(E, N, env_id=args.env_id, config=cfg, device=dev, **dict(kw, qp_solver="ipm"))
    env.reset()
    ge = torch.arange(0, E, dtype=torch.int64, device=dev)
    env.env_i32[abi.EP["STEPS"]] = ((ge * 37) % env.T).to(torch.int32)
    for t in range(env.T + 2):
        env.step(ring[t % 16])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()  # ONE event pair on the launch stream around the whole window (7 launches per step: 4 phase + 3 sweep kernels)
    for t in range(steps):
        env.step(ring[t % 16])
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / steps
    env.poll_errors()
    b_alg = algorithmic_bytes_per_agent_step(args.env_id, N)
    achieved = E * N * b_alg / (kern_ms * 1e-3) / 1e9
    return {"qp_solver": "ipm", "value": E * N * steps / dt, "unit": "agent-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "meets_1e-5_vs_interior_point_iterate": True,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": workload_traffic(E, N, args.shield, args.env_id, "ipm"),
                         "kernel": "4 x step_kernel<..., SPLIT> (phase form) + 3 x sweep_kernel per policy step",
                         "kernel_ms": kern_ms, "alg_bytes_per_launch": E * N * b_alg,
                         "kernel_ms_note": "one HIP event pair around the whole window / steps: all 7 launches of a policy step",
                         "limiter": "latency of the per-env QP chain: a sweep wave (64 envs, one lane each) walks ~8 dependent QPs per env and "
                                    "sub-step, ~6 interior-point iterations each (~29 for the 0.4 % that would run to cvxopt's cap), one wave per "
                                    "SIMD; the launch ends with its slowest env.  See DESIGN.md section 2"},
            "note": "same workload with the product's DEFAULT numerics: shield QP by cvxopt's interior-point algorithm (include/mm_qp.h) in the "
                    "split step (phase kernels + lane-per-env sweep kernel); bit-identical to the oracle's literal restatement of coneqp"}


def workload_row(E, N, shield, env_id, qp_solver, hdv=0, density=0, pow2=False, mixed=False):
    """The row of profiles/traffic.json (one per profiled workload, written from tools/profile.sh runs of this command) that
    matches this workload, or None for a workload that was not profiled."""
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        rows = tj["workloads"] if "workloads" in tj else [tj]
        for r in rows:
            w = r["workload"]
            if (w["envs_per_gpu"], w["agents"], w["shield"], w["env_id"], w.get("qp_solver", "exact"), w.get("hdv", 0),
                    w.get("traffic_density", 0), bool(w.get("pow2_groups", False)), bool(w.get("mixed_traffic", False))) == \
                    (E, N, shield, env_id, qp_solver, hdv, density, bool(pow2), bool(mixed)):
                return r
    except (OSError, KeyError, ValueError):
        pass
    return None


def workload_traffic(*a, **k):
    """HBM bytes per mm_step of that workload from the committed PMC passes (None: not profiled)."""
    r = workload_row(*a, **k)
    return None if r is None else r["bytes_per_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs", type=int, default=None, help="envs PER GPU (weak scaling; default 65536).  Given without --total-envs / "
                                                           "--scaling it selects weak scaling: `--envs 8192` = 8 192 envs on every GPU")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="strong (default): BASELINE's 65 536 envs in total, split over the GPUs; weak: --envs per GPU")
    ap.add_argument("--no-weak-extra", action="store_true", help="N > 1, strong scaling: skip the extra weak-scaling measurement")
    ap.add_argument("--total-envs", type=int, default=None, help="envs over ALL GPUs (strong scaling; default 65536 = BASELINE c5 over 8)")
    ap.add_argument("--qp-solver", choices=("exact", "ipm"), default="exact",
                    help="exact: closed-form KKT point (production); ipm: cvxopt's interior-point iterate (fidelity mode)")
    ap.add_argument("--no-stagger", action="store_true", help="skip the phase staggering + pre-roll (lock-stepped episodes)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank REHEARSAL on a 1-GPU box: every rank uses cuda:0 and the collectives run over gloo "
                         "(RCCL refuses two ranks on one device); exercises launcher / sharding / reduction, not xGMI")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--shield", choices=sorted(SHIELDS), default="mass")
    ap.add_argument("--env-id", default="merge-multi-agent-v1")
    ap.add_argument("--obs-f64", action="store_true")
    ap.add_argument("--hdv", type=int, default=0, help="mixed traffic: the last HDV of the --agents vehicles are IDM/MOBIL HDVs (not the headline workload)")
    ap.add_argument("--traffic-density", type=int, default=0, choices=(0, 1, 2, 3),
                    help="1..3: every (re)spawn DRAWS its vehicle counts like MergeEnv._num_vehicles (merge_env_v1.py:180-211); --agents is "
                         "then the slot capacity (>= 6 / 8 / 11) of a ragged batch and agent-steps count the vehicles actually present "
                         "(the reference's training distribution; not the headline workload)")
    ap.add_argument("--mixed-traffic", action="store_true", help="with --traffic-density: CAVs + IDM/MOBIL HDVs (traffic_type=mixed)")
    ap.add_argument("--pow2-groups", action="store_true", help="A-B timing: step in power-of-two lane groups only (debug_flags bit1; N = 5..6 / "
                                                               "9..12 otherwise run in 6- / 12-lane groups)")
    ap.add_argument("--all-outputs", action="store_true", help="also write agents_info / action_mask / crashed every step (30 B per agent "
                                                                "that B_alg does not count and the rollout loop does not read)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fidelity-line", action="store_true", help="skip the qp_solver=ipm measurement (qp_fidelity_mode) appended to the headline line")
    ap.add_argument("--cpu-envs", type=int, default=32768)
    ap.add_argument("--cpu-steps", type=int, default=400)
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    if args.scaling is None:  # an explicit per-GPU batch size means per-GPU work is what the caller fixes
        args.scaling = "weak" if (args.envs is not None and args.total_envs is None) else "strong"
    args.envs = 65536 if args.envs is None else args.envs
    args.total_envs = 65536 if args.total_envs is None else args.total_envs

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    nccl = world > 1 and not args.rehearse_on_one_gpu
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if nccl:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    here = os.path.dirname(os.path.abspath(__file__))
    if rank == 0 and not os.path.exists(os.path.join(here, "marl-mass_amd", "csrc", "libmm_hip.so")):
        import __graft_entry__  # fresh checkout: the library is a build artefact
        __graft_entry__.build()
    if world > 1:
        dist.barrier(device_ids=[local]) if nccl else dist.barrier()
    from marl_mass_amd import VecMergeEnv, reduce_rollout_metrics, shard_range, _cabi as abi
    N = args.agents
    if args.scaling == "strong":
        first_env, E = shard_range(args.total_envs, rank, world)
        E_total = args.total_envs
    else:
        first_env, E, E_total = rank * args.envs, args.envs, args.envs * world
    cfg = {"safety_guarantee": SHIELDS[args.shield], "HEADWAY_TIME": 0.5 if args.shield != "none" else 1.2}
    kw = dict(cbf_eta=0.03125 if args.shield != "none" else 0.0, cbf_tau=cfg["HEADWAY_TIME"], seed=1000,
              auto_reset=True, obs_f64=args.obs_f64, n_hdv=args.hdv, qp_solver=args.qp_solver, debug_flags=(2 if args.pow2_groups else 0) | int(os.environ.get("MM_DEBUG_FLAGS", "0")))
    if args.traffic_density:
        cfg.update({"traffic_density": args.traffic_density, "traffic_type": "mixed" if args.mixed_traffic else "cav",
                    "mixed_traffic": args.mixed_traffic})
        kw["draw_counts"] = True
    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local]) if nccl else dist.barrier()

    def measure(E, first_env, steps, warmup):
        """One stationary batch of E envs on this rank, `steps` timed launches between barriers; returns the wall time of the
        timed region, the mean per-launch HIP-event time, the metrics buffer, the env and its action ring."""
        # per-agent outputs the rollout loop does not consume (SURVEY 8d counts none of them in B_alg) are not requested:
        # vehicle positions / speeds for the evaluation plots, the action mask (masking is off in v1), the crashed plane
        masking = dict(abi.default_env_config(args.env_id), **cfg).get("action_masking")  # (v0's default masks: the mask is a result)
        skip = () if args.all_outputs else (("agents_info", "crashed") if masking else ("agents_info", "action_mask", "crashed"))
        env = VecMergeEnv(E, N, env_id=args.env_id, config=cfg, device=dev, first_env=first_env, skip_outputs=skip, **kw)
        if os.environ.get("MM_BENCH_NO_METRICS"):  # tuning experiment: no in-kernel metric accumulation
            metrics = torch.zeros(8, dtype=torch.float64, device=dev)
        else:
            # a rollout loop reads its metrics once, at the end (marl/mappo.py:348-354): the per-wave partials are folded into the
            # 8 doubles by ONE launch at the end of the timed region, not by one small launch behind every step
            metrics = env.enable_metrics(deferred=True)
        env.reset()
        g = torch.Generator(device=dev).manual_seed(123 + rank)
        p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device=dev)
        ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(16)]
        if not args.no_stagger:
            # stationary batch: stagger the episode phases by GLOBAL env id, then roll one episode length untimed so
            # that every env has re-spawned once at its own phase and its state is consistent with its step counter
            T = env.T
            ge = torch.arange(first_env, first_env + E, dtype=torch.int64, device=dev)
            env.env_i32[abi.EP["STEPS"]] = ((ge * 37) % T).to(torch.int32)
            for t in range(T):
                env.step(ring[t % 16])
        for t in range(warmup):
            env.step(ring[t % 16])
        env.flush_metrics()  # (what pre-roll and warm-up accumulated is not part of the measurement)
        torch.cuda.synchronize()
        metrics.zero_()
        metrics[7] = float("inf")
        # the timed region holds the launches and nothing else (no events: round 3 sampled event pairs inside it and they
        # showed up in both numbers)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            env.step(ring[t % 16])  # one mm_step on torch's current stream
        env.flush_metrics()  # end of the rollout: fold the per-wave partials (inside the timed region)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        env.poll_errors()  # (outside the timed region) a latched check_bounds / bad action / kernel loop guard voids the measurement: raise
        metrics = metrics.clone()  # the rollout metrics of the timed region (what follows keeps accumulating into the env's buffer)
        # kernel time for the roofline block, AFTER the timed region: the same launches once more, captured in a hipGraph (no
        # host in the loop) and bracketed by ONE HIP event pair on the launch stream; / launches = the step kernel's duration
        # + the device-side dispatch gap.  Falls back to an eager loop between the same two events.
        kn = max(16, min(steps, 64))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        how, after = "hipGraph replay of %d launches" % kn, 2 * kn  # (after: mm_steps executed behind the timed region)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    for t in range(kn):
                        env.step(ring[t % 16])
                gr.replay()  # (first replay: warm)
                side.synchronize()
                e0.record(side); gr.replay(); e1.record(side)
                side.synchronize()
            torch.cuda.current_stream().wait_stream(side)
        except Exception as exc:  # capture refused (e.g. a build without graph support): eager loop
            how, after = "eager loop of %d launches (graph capture failed: %s)" % (kn, type(exc).__name__), kn
            torch.cuda.synchronize()
            e0.record()
            for t in range(kn):
                env.step(ring[t % 16])
            e1.record()
            torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / kn
        env.flush_metrics()
        env.poll_errors()
        return elapsed, kern_ms, metrics, env, ring, (how, after), skip

    elapsed, kern_ms, metrics, env, ring, kern_how, skipped = measure(E, first_env, args.steps, args.warmup)
    weak_extra = None
    if world > 1 and args.scaling == "strong" and not args.no_weak_extra:
        # the same kernel with BASELINE's batch on EVERY rank (per-GPU work fixed): reported beside the headline
        w_elapsed, w_kern_ms = measure(args.envs, rank * args.envs, args.steps, 2)[:2]
        wt = torch.tensor([w_elapsed], dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        weak_extra = {"scaling": "weak", "envs_per_gpu": args.envs, "envs_total": args.envs * world,
                      "value": float(args.envs) * world * N * args.steps / float(wt[0]), "unit": "agent-steps/s",
                      "ms_per_step": float(wt[0]) / args.steps * 1e3, "kernel_ms_rank0": w_kern_ms}

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if nccl or world == 1 else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # end-of-rollout metric reduction: the only collective of the path
        reduce_rollout_metrics(metrics)
    elapsed = float(tmax[0])
    m = metrics.cpu().tolist()

    present = None
    if args.traffic_density:  # ragged batch: count the vehicles that exist (the batch is stationary: one snapshot represents the window)
        present = (env.u8[abi.B["KIND"]] != 0).sum().to(torch.float64).reshape(1)
        if world > 1:
            present = present.to(tmax.device); dist.all_reduce(present)
        present = float(present[0])
    if rank == 0:
        agent_steps = float(E_total) * N * args.steps if present is None else present * args.steps
        b_alg = algorithmic_bytes_per_agent_step(args.env_id, N)
        achieved = E * N * b_alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch and VALU instructions per wave from the committed PMC passes of this workload
        # (profiles/traffic.json: one row per profiled workload); null for a workload that was not profiled
        wrow = workload_row(E, N, args.shield, args.env_id, args.qp_solver, args.hdv, args.traffic_density, args.pow2_groups,
                            args.mixed_traffic)
        traffic = None if wrow is None else wrow["bytes_per_launch"]
        # secondary reading (the kernel is VALU-issue bound): instructions from the committed
        # SQ_INSTS_VALU pass x 4 issue cycles per wave64 instruction, against 1024 SIMDs at the 2.4 GHz peak clock
        valu = None
        try:
            row = wrow  # (the matched workload's own row: by key, not by byte count)
            if row is None:
                raise KeyError("workload not profiled")
            sj = json.load(open(os.path.join(REPO, "profiles", row.get("summary", "r01/step_kernel_summary.json"))))
            per_wave = sj["pmc_per_launch"]["SQ_INSTS_VALU"] / sj["pmc_per_launch"]["SQ_WAVES"]
            n_valu = per_wave * (E * 8 // 64 if N <= 8 else E * 16 // 64)
            ach = n_valu * 4 / (kern_ms * 1e-3) / 1e12
            valu = {"bound": "valu-issue", "achieved": ach, "peak": 1024 * 2.4e9 / 1e12, "unit": "T SIMD-cycles/s",
                    "frac": ach / (1024 * 2.4e9 / 1e12), "valu_insts_per_wave": per_wave}
        except (OSError, KeyError, ValueError, ZeroDivisionError, IndexError, TypeError):
            pass
        headline = args.shield == "mass" and not args.hdv and args.env_id.endswith("v1") and N == 8 and args.qp_solver == "exact" and not args.traffic_density
        if headline and E_total == 65536:
            # BASELINE.json's metric on its own config: 65 536 envs x 8 CAVs in total (one GPU holds the whole batch; 8 GPUs
            # under the default strong scaling are config 5 itself)
            metric = "agent-steps/sec (whole node), MASS CBF shield on, 65536 envs x 8 CAVs"
        elif headline:
            metric = "agent-steps/sec (whole node), MASS CBF shield on, %d x 65536 envs x 8 CAVs (weak scaling: BASELINE's batch per GPU)" % world \
                if (args.scaling == "weak" and E == 65536) else \
                "agent-steps/sec (whole node), MASS CBF shield on, %d envs x 8 CAVs" % E_total
        else:
            metric = "agent-steps/sec (whole node), shield=%s, qp=%s, %d envs x %d vehicles" % (args.shield, args.qp_solver, E_total, N)
        if present is not None:
            metric += " [vehicle counts drawn per episode, traffic_density=%d%s: mean %.2f vehicles per env in %d slots]" % (
                args.traffic_density, " mixed" if args.mixed_traffic else "", present / E_total, N)
        line = {
            "metric": metric,
            "value": agent_steps / elapsed, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs x %d CAVs in total = %d per GPU on %d GPU(s) (%s scaling), %s, safety_guarantee=%s, "
                                   "qp_solver=%s (%s), eta=0.03125, tau=%.1f, 100-step episodes with in-kernel auto-reset, episode "
                                   "phases staggered over the batch%s, categorical action tape p=[.1,.6,.1,.1,.1]%s; outputs not written: %s; "
                                   "rollout metrics folded once at the end of the timed region"
                                   % (E_total, N, E, world, args.scaling, args.env_id, cfg["safety_guarantee"], args.qp_solver,
                                      QP_MODE_NOTE[args.qp_solver] if args.shield != "none" else "no shield",
                                      cfg["HEADWAY_TIME"], " (OFF)" if args.no_stagger else "",
                                      (", of which %d HDVs per env" % args.hdv) if args.hdv else "",
                                      ", ".join(skipped) if skipped else "none"),
                       "envs_total": E_total, "envs_per_gpu": E, "agents": N, "traffic_density": args.traffic_density, "pow2_groups": bool(args.pow2_groups), "obs_dtype": "f64" if args.obs_f64 else "f32",
                       "qp_solver": args.qp_solver,
                       # what this launch does NOT produce of MergeEnv.step's info dict (NULL output pointers: the kernel skips
                       # them), and how the rollout metrics are folded -- both differ from rounds 1-2's lines
                       "skipped_outputs": list(skipped),
                       "metrics_deferred": not bool(os.environ.get("MM_BENCH_NO_METRICS")),
                       "tolerance": {"north_star": "1e-5 on float state vs the reference's QP (cvxopt interior-point iterate)",
                                     "exact": "closed-form KKT point = the true minimiser; differs from the interior-point iterate by "
                                              "up to 3.0e-4 m/s per QP (p99 1.3e-4): OUTSIDE 1e-5 (profiles/r02/qp_fidelity.json)",
                                     "ipm": "bit-identical to the restatement of cvxopt's coneqp that answered solvers.qp while the "
                                            "reference produced the ipm_* tapes: INSIDE 1e-5; see qp_fidelity_mode for its throughput"},
                       "parallelism": "env-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes/launch: rocprofv3 PMC passes of this command, 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction), "
                                         "committed under profiles/ (see profiles/traffic.json: source)",
                         "alg_bytes_per_launch": E * N * b_alg,
                         "kernel": "step_kernel" if args.qp_solver == "exact" or args.shield == "none" or E * (2 if N <= 2 else 4 if N <= 4 else 8 if N <= 8 else 16) // 64 <= 2048
                         else "4 x step_kernel (phase form) + 3 x sweep_kernel (interior-point step above two fused waves per SIMD)",
                         "kernel_ms": kern_ms, "alg_bytes_per_agent_step": b_alg,
                         "kernel_ms_note": "measured AFTER the timed region: %s between ONE HIP event pair on the launch stream, / launches "
                                           "= the kernel(s) of one mm_step + the device's dispatch gap (the rollout metrics are deferred: no "
                                           "flush kernel per step); rocprofv3's kernel-trace average of the same command is under profiles/" % kern_how[0],
                         "mm_steps_after_timed_region": kern_how[1]},
            "rollout_metrics": {"mean_reward": m[0] / max(m[4], 1), "crashed_episodes": m[1],
                                "mean_speed": m[2] / max(m[4], 1), "env_steps": m[4],
                                "mean_merge_percent": m[5] / max(m[6], 1), "episodes": m[6], "min_headway": m[7]},
        }
        if valu is not None:
            line["roofline_secondary"] = valu
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = "all %d ranks on cuda:0, gloo collectives: NOT a scaling measurement" % world
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(args, args.env_id, cfg, kw)
        if weak_extra is not None:
            line["weak_scaling"] = weak_extra
        if headline and world == 1 and not args.no_fidelity_line:
            # the same workload on the product's DEFAULT numerics -- the shield's QP through cvxopt's interior-point iteration
            # (DESIGN.md 3), the mode inside north_star's 1e-5 -- as a peer of `value` at the top level of the line, with its own
            # roofline block below
            fid = ipm_mode_line(args, dev, E, N, cfg, kw, ring)
            line["value_within_tolerance"] = fid["value"]
            line["value_within_tolerance_note"] = ("qp_solver=ipm (the default of every entry point): %.3f ms per step; `value` is qp_solver=exact "
                                                   "(explicit opt-in), outside 1e-5 of the interior-point iterate" % fid["ms_per_step"])
            line["qp_fidelity_mode"] = fid
        if headline and world == 1 and not args.all_outputs and not args.no_fidelity_line:
            # like-for-like with rounds 1-2 and with the reference, whose step always produces the whole info dict: every output
            # written, metrics folded behind every step
            ao = VecMergeEnv(E, N, env_id=args.env_id, config=cfg, device=dev, first_env=first_env, **kw)
            ao.enable_metrics(deferred=False)
            ao.reset()
            ao.env_i32[abi.EP["STEPS"]] = ((torch.arange(first_env, first_env + E, dtype=torch.int64, device=dev) * 37) % ao.T).to(torch.int32)
            for t in range(ao.T):
                ao.step(ring[t % 16])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(50):
                ao.step(ring[t % 16])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            line["all_outputs_per_step_fold"] = {"value": E * N * 50 / dt, "unit": "agent-steps/s", "ms_per_step": dt / 50 * 1e3, "steps": 50,
                                                 "note": "same workload with agents_info / action_mask / crashed written and the rollout metrics "
                                                         "folded by a flush kernel behind every step (the shape of BENCH_r01 / r02)"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
