"""N>1 path on CPU: two gloo ranks each own half of the env batch (oracle backend).  Sharding must
not change any env (RNG streams are keyed by global env id) and the metric all-reduce must equal
the single-process accumulator."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, N, STEPS = 48, 4, 104
KW = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
          cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=77, auto_reset=True)


def _tape():
    g = torch.Generator().manual_seed(5)
    p = torch.tensor([0.1, 0.5, 0.2, 0.1, 0.1])
    return [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(STEPS)]


def _run(env, actions):
    m = env.enable_metrics()
    env.reset()
    for a in actions:
        env.step(a)
    return m


def _worker(rank, world, port, out_dir):
    for p in (REPO, os.path.join(REPO, "oracle")):
        sys.path.insert(0, p)
    import oracle_env
    from marl_mass_amd import reduce_rollout_metrics, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle_env.library().lib.orc_set_threads(1)
    first, count = shard_range(E, rank, world)
    env = oracle_env.OracleEnv(count, N, first_env=first, **KW)
    m = _run(env, [a[first:first + count].contiguous() for a in _tape()])
    reduce_rollout_metrics(m)
    torch.save({"f64": env.f64.clone(), "u8": env.u8.clone(), "i32": env.env_i32.clone(), "metrics": m.clone(),
                "first": first, "count": count}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    import oracle_env
    oracle_env.library().lib.orc_set_threads(1)
    full = oracle_env.OracleEnv(E, N, **KW)
    m = _run(full, _tape())
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(2)]
    for part in parts:
        sl = slice(part["first"], part["first"] + part["count"])
        assert torch.equal(part["f64"].nan_to_num(), full.f64[:, sl].nan_to_num())
        assert torch.equal(part["u8"], full.u8[:, sl]) and torch.equal(part["i32"], full.env_i32[:, sl])
    # every rank holds the reduced metrics; sums differ from the single-process order only by fp reassociation
    for part in parts:
        assert torch.allclose(part["metrics"][:7], m[:7], rtol=1e-12, atol=0)
        assert float(part["metrics"][7]) == float(m[7])
    assert float(m[4]) == E * STEPS and float(m[6]) >= E  # env-steps counted, every episode finished once


def _worker_hip(rank, world, port, out_dir):
    """One rank of the HIP backend: its own process, its own handle on cuda:0 (the 1-GPU box has no second device; on the
    node every rank has its own GPU and the collective is RCCL), metrics reduced over gloo through host copies."""
    sys.path.insert(0, REPO)
    from marl_mass_amd import VecMergeEnv, reduce_rollout_metrics, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    EH = 4096
    first, count = shard_range(EH, rank, world)
    env = VecMergeEnv(count, 8, device="cuda:0", first_env=first, **dict(KW, seed=1000))
    m = env.enable_metrics()
    env.reset()
    g = torch.Generator().manual_seed(5)
    p = torch.tensor([0.1, 0.5, 0.2, 0.1, 0.1])
    for _ in range(STEPS):
        a = torch.multinomial(p, EH * 8, True, generator=g).view(EH, 8).int()
        env.step(a[first:first + count].contiguous().cuda())
    reduce_rollout_metrics(m)
    torch.save({"f64": env.f64.cpu(), "u8": env.u8.cpu(), "i32": env.env_i32.cpu(), "metrics": m.cpu(), "first": first, "count": count},
               os.path.join(out_dir, "hip_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_shards_equal_single_batch_hip(tmp_path):
    """The N > 1 path on the product backend: two rank processes (each with its own library handle, shard and
    `first_env`), metric all-reduce, against one process holding the whole batch -- every bit of state, and the metrics."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_hip, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from marl_mass_amd import VecMergeEnv
    EH = 4096
    full = VecMergeEnv(EH, 8, device="cuda:0", **dict(KW, seed=1000))
    m = full.enable_metrics()
    full.reset()
    g = torch.Generator().manual_seed(5)
    p = torch.tensor([0.1, 0.5, 0.2, 0.1, 0.1])
    for _ in range(STEPS):
        full.step(torch.multinomial(p, EH * 8, True, generator=g).view(EH, 8).int().cuda())
    parts = [torch.load(os.path.join(str(tmp_path), "hip_rank%d.pt" % r)) for r in range(2)]
    for part in parts:
        sl = slice(part["first"], part["first"] + part["count"])
        assert torch.equal(part["f64"].nan_to_num(), full.f64[:, sl].cpu().nan_to_num())
        assert torch.equal(part["u8"], full.u8[:, sl].cpu()) and torch.equal(part["i32"], full.env_i32[:, sl].cpu())
        assert torch.allclose(part["metrics"][:7], m.cpu()[:7], rtol=1e-12, atol=0)
        assert float(part["metrics"][7]) == float(m[7])
    assert float(m[4]) == EH * STEPS


def test_shard_range_covers_batch():
    from marl_mass_amd import shard_range
    for total, world in ((65536, 8), (10, 3), (7, 8)):
        got = [shard_range(total, r, world) for r in range(world)]
        assert sum(c for _, c in got) == total
        assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(world - 1))


def test_checkpoint_resume_is_bit_identical():
    """state_dict / load_state_dict of an env batch (oracle backend): resume reproduces the continuation."""
    import torch
    import oracle_env
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact",
              cbf_tau=0.5, seed=12, auto_reset=True, n_hdv=2)
    env = oracle_env.OracleEnv(32, 6, **kw)
    env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [torch.randint(0, 5, (32, 6), generator=g, dtype=torch.int32) for _ in range(30)]
    for a in acts[:10]:
        env.step(a)
    ck = env.state_dict()
    ref = [tuple(t.clone() for t in env.step(a)[:3]) for a in acts[10:]]
    fresh = oracle_env.OracleEnv(32, 6, **kw)
    fresh.load_state_dict(ck)
    for a, (o, r, d) in zip(acts[10:], ref):
        o2, r2, d2, _ = fresh.step(a)
        assert torch.equal(o2, o) and torch.equal(r2, r) and torch.equal(d2, d)
    assert torch.equal(fresh.state, env.state)
    import pytest
    with pytest.raises(ValueError):
        oracle_env.OracleEnv(16, 6, **kw).load_state_dict(ck)
