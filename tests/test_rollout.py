"""On-device rollout counterpart of MAPPO.interact (marl/mappo.py:102-158,364-370), on CPU with the
oracle backend: returns must equal the reference's per-agent `_discount_reward` applied episode by
episode to the same reward tape."""
import numpy as np
import pytest
import torch

import oracle_env
from marl_mass_amd.rollout import ActorNetwork, CriticNetwork, DeviceRollout, discount_rewards


def _discount_reward(rewards, final_value, gamma):  # marl/mappo.py:364-370 semantics
    out = np.zeros_like(rewards)
    running = final_value
    for t in reversed(range(len(rewards))):
        running = running * gamma + rewards[t]
        out[t] = running
    return out


def test_discount_matches_reference_per_episode():
    rs = np.random.RandomState(0)
    T, E, N = 37, 5, 3
    r = rs.randn(T, E, N)
    d = rs.rand(T, E) < 0.1
    fv = rs.randn(E, N)
    got = discount_rewards(torch.tensor(r), torch.tensor(d), torch.tensor(fv), 0.99).numpy()
    for e in range(E):
        for n in range(N):
            t0 = 0
            ends = [t for t in range(T) if d[t, e]] + ([T - 1] if not d[T - 1, e] else [])
            for t1 in ends:
                final = 0.0 if d[t1, e] else fv[e, n]
                np.testing.assert_allclose(got[t0:t1 + 1, e, n], _discount_reward(r[t0:t1 + 1, e, n], final, 0.99),
                                           rtol=1e-12, atol=1e-12)
                t0 = t1 + 1


@pytest.mark.parametrize("reward_type", ["regionalR", "global_R"])
def test_interact_shapes_and_bookkeeping(reward_type):
    torch.manual_seed(0)
    E, N = 6, 4
    env = oracle_env.OracleEnv(E, N, env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
                               cbf_eta=0.03125, cbf_tau=0.5, seed=5, auto_reset=True)
    actor, critic = ActorNetwork(env.n_s, 128, env.n_a), CriticNetwork(env.n_s, env.n_a, 128)
    g = torch.Generator().manual_seed(1)
    ro = DeviceRollout(env, actor, critic, roll_out_n_steps=110, reward_type=reward_type, generator=g)
    out = ro.interact()
    assert out["states"].shape == (110, E, N, env.n_s) and out["actions"].shape == (110, E, N)
    assert out["returns"].shape == (110, E, N) and torch.isfinite(out["returns"]).all()
    assert int(out["dones"].sum()) == E  # 100-step episodes under the shield: each env finished exactly once
    assert (out["actions"] >= 0).all() and (out["actions"] < 5).all()
    out2 = ro.interact()  # a second rollout continues from the carried observation
    assert not torch.equal(out2["states"][0], out["states"][0])


@pytest.mark.gpu
def test_interact_on_gpu():
    from marl_mass_amd import VecMergeEnv
    E, N = 4096, 8
    env = VecMergeEnv(E, N, config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, cbf_tau=0.5,
                      seed=5, auto_reset=True)
    actor = ActorNetwork(env.n_s, 128, env.n_a).cuda()
    critic = CriticNetwork(env.n_s, env.n_a, 128).cuda()
    ro = DeviceRollout(env, actor, critic, roll_out_n_steps=20)
    out = ro.interact()
    assert out["returns"].is_cuda and torch.isfinite(out["returns"]).all()
    assert out["states"].shape == (20, E, N, 30)


@pytest.mark.gpu
def test_graph_captured_rollout_equals_eager():
    """use_graph: the whole rollout replayed as one hipGraph gives the same tensors as the eager loop
    (a greedy policy makes the comparison independent of the RNG stream), twice in a row."""
    import time
    from marl_mass_amd import VecMergeEnv
    E, N, T = 8192, 8, 25
    kw = dict(config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, cbf_tau=0.5, seed=9, auto_reset=True)
    torch.manual_seed(3)

    class GreedyActor(ActorNetwork):  # exactly one-hot log-probabilities: sampling has a single outcome
        def forward(self, state):
            lp = super().forward(state)
            return torch.where(lp == lp.max(-1, keepdim=True).values, 0.0, -float("inf")).to(lp.dtype)

    actor = GreedyActor(30, 128, 5).cuda()
    critic = CriticNetwork(30, 5, 128).cuda()
    eager = DeviceRollout(VecMergeEnv(E, N, **kw), actor, critic, roll_out_n_steps=T)
    graph = DeviceRollout(VecMergeEnv(E, N, **kw), actor, critic, roll_out_n_steps=T, use_graph=True)
    graph.interact()  # warm-up + capture + first replay = 2 rollouts
    eager.interact()
    eager.interact()
    for _ in range(2):
        a, b = eager.interact(), graph.interact()
        torch.cuda.synchronize()
        for k in ("states", "actions", "returns", "dones", "average_speed", "min_headway"):
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(eager.env.f64.nan_to_num(), graph.env.f64.nan_to_num())
    t0 = time.perf_counter(); eager.interact(); torch.cuda.synchronize(); t1 = time.perf_counter()
    graph.interact(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rollout of %d steps x %d envs: eager %.2f ms, hipGraph %.2f ms" % (T, E, (t1 - t0) * 1e3, (t2 - t1) * 1e3))


def test_evaluate_matches_per_episode_loop():
    """DeviceRollout.evaluate vs the reference's evaluation loop (marl/mappo.py:255-361) run env by env
    on single-env batches with the same seeds and a greedy policy."""
    torch.manual_seed(0)
    E, N = 5, 4
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "none", "HEADWAY_TIME": 1.2}, seed=40)

    class GreedyActor(ActorNetwork):
        def forward(self, state):
            lp = super().forward(state)
            return torch.where(lp == lp.max(-1, keepdim=True).values, 0.0, -float("inf")).to(lp.dtype)

    actor = GreedyActor(30, 128, 5)
    ro = DeviceRollout(oracle_env.OracleEnv(E, N, auto_reset=True, **kw), actor, roll_out_n_steps=10)
    seeds = [7, 8, 9, 10, 11]
    rewards, (vs, vp), ext = ro.evaluate(seeds=seeds)
    assert ro.env.auto_reset is True  # restored
    for e, sd in enumerate(seeds):  # the reference's loop, one env at a time
        env1 = oracle_env.OracleEnv(1, N, auto_reset=False, **kw)
        obs, _ = env1.reset(seeds=torch.tensor([sd]))
        step, avg, tsp, rs, mh, done = 0, 0.0, 0.0, [], float("inf"), False
        while not done:
            lp = actor(obs.reshape(N, 30).float())
            obs, r, d, info = env1.step(lp.argmax(-1).view(1, N).int())
            step += 1
            avg += float(info["average_speed"][0]); tsp += float(info["traffic_speed"][0])
            mh = min(mh, float(info["min_headway"][0]))
            rs.append(float(r[0]))
            done = bool(d[0])
        assert int(ext["steps"][e]) == step
        np.testing.assert_allclose(rewards[:step, e].numpy(), np.array(rs), rtol=0, atol=0)
        assert torch.isnan(rewards[step:, e]).all()
        assert abs(float(ext["avg_speeds"][e]) - avg / step) <= 1e-12
        assert abs(float(ext["traffic_speeds"][e]) - tsp / step) <= 1e-12
        assert bool(ext["crash_count"][e]) == bool(info["crashed"][0].any())
        assert abs(float(ext["merge_percents"][e]) - float(info["merge_percent"][0])) <= 1e-12
        assert ext["min_headway"] <= mh + 1e-15
