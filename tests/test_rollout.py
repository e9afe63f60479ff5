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
                               cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=5, auto_reset=True)
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
    env = VecMergeEnv(E, N, config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5,
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
    kw = dict(config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=9, auto_reset=True)
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


def _sample(clib, logp, seed, counter):
    out = torch.empty(logp.shape[0], dtype=torch.int32, device=logp.device)
    stream = None
    if logp.is_cuda:
        import ctypes
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    clib.check(clib.lib.mm_sample_actions(logp.data_ptr(), logp.shape[0], logp.shape[1], seed, counter.data_ptr(),
                                          out.data_ptr(), stream))
    return out


def test_sample_actions_distribution_and_counter():
    """mm_sample_actions (oracle build): inverse-CDF sampling of softmax probabilities, fresh numbers per call."""
    clib = oracle_env.library()
    torch.manual_seed(1)
    n = 200000
    logp = torch.log_softmax(torch.tensor([[0.3, 1.2, -0.5, 0.0, 2.0]]), -1).repeat(n, 1).contiguous()
    ctr = torch.zeros(1, dtype=torch.int64)
    a0 = _sample(clib, logp, 5, ctr)
    a1 = _sample(clib, logp, 5, ctr)
    assert int(ctr) == 2 and not torch.equal(a0, a1)
    freq = torch.bincount(a0.long(), minlength=5).double() / n
    assert torch.allclose(freq, logp[0].exp().double(), atol=5e-3)
    ctr.zero_()
    assert torch.equal(_sample(clib, logp, 5, ctr), a0)          # same key -> same draw
    one_hot = torch.full((7, 5), -float("inf")); one_hot[torch.arange(7), torch.arange(7) % 5] = 0.0
    assert _sample(clib, one_hot.contiguous(), 9, ctr).tolist() == [0, 1, 2, 3, 4, 0, 1]
    with pytest.raises(ValueError):
        _sample(clib, torch.zeros(4, 9), 0, ctr)


@pytest.mark.gpu
def test_sample_actions_hip_equals_oracle():
    from marl_mass_amd import hip_library
    torch.manual_seed(2)
    n = 1 << 18
    logp = torch.log_softmax(torch.randn(n, 5) * 2, -1).contiguous()
    c_cpu, c_gpu = torch.tensor([41], dtype=torch.int64), torch.tensor([41], dtype=torch.int64, device="cuda")
    for _ in range(2):
        a_cpu = _sample(oracle_env.library(), logp, 1234567, c_cpu)
        a_gpu = _sample(hip_library(), logp.cuda(), 1234567, c_gpu)
        assert torch.equal(a_gpu.cpu(), a_cpu)
    assert int(c_gpu) == int(c_cpu) == 43


def _policy_act(clib, obs, actor, seed, counter, want_logp=True):
    import ctypes
    n, S = obs.shape
    acts = torch.empty(n, dtype=torch.int32, device=obs.device)
    logp = torch.empty(n, 5, dtype=torch.float32, device=obs.device) if want_logp else None
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if obs.is_cuda else None
    p = lambda t: t.detach().contiguous().data_ptr()  # noqa: E731
    clib.check(clib.lib.mm_policy_act(obs.data_ptr(), n, S, p(actor.fc1.weight), p(actor.fc1.bias), p(actor.fc2.weight),
                                      p(actor.fc2.bias), p(actor.fc3.weight), p(actor.fc3.bias), 128, 5, seed, counter.data_ptr(),
                                      acts.data_ptr(), logp.data_ptr() if want_logp else None, stream))
    return acts, logp


@pytest.mark.parametrize("n_s,n", [(30, 1000), (25, 77)])
def test_policy_act_oracle_matches_torch(n_s, n):
    """mm_policy_act (oracle build) == ActorNetwork forward (Model_common.py:5-22) + mm_sample_actions."""
    torch.manual_seed(4)
    actor = ActorNetwork(n_s, 128, 5)
    obs = torch.randn(n, n_s)
    ctr = torch.tensor([3], dtype=torch.int64)
    acts, logp = _policy_act(oracle_env.library(), obs, actor, 11, ctr)
    with torch.no_grad():
        ref = actor(obs)
    assert torch.allclose(logp, ref, atol=2e-5, rtol=0) and int(ctr) == 4
    ctr2 = torch.tensor([3], dtype=torch.int64)
    assert torch.equal(acts, _sample(oracle_env.library(), logp.contiguous(), 11, ctr2))


@pytest.mark.gpu
@pytest.mark.parametrize("n_s,n", [(30, 65536 * 8 + 13), (25, 1000), (30, 31)])
def test_policy_act_hip_mfma(n_s, n):
    """The fused f32-MFMA actor + sampling launch vs torch fp32 (log-probabilities to rounding) and vs the
    oracle's sampler on the kernel's own log-probabilities (actions exact); ragged n, both obs widths."""
    from marl_mass_amd import hip_library
    torch.manual_seed(5)
    actor = ActorNetwork(n_s, 128, 5).cuda()
    with torch.no_grad():  # asymmetric, non-trivial scales in every layer
        actor.fc1.bias.uniform_(-0.5, 0.5); actor.fc2.bias.uniform_(-0.5, 0.5); actor.fc3.bias.uniform_(-1, 1)
        actor.fc3.weight.mul_(3.0)
    obs = (torch.randn(n, n_s, device="cuda") * 1.5).contiguous()
    ctr = torch.tensor([8], dtype=torch.int64, device="cuda")
    acts, logp = _policy_act(hip_library(), obs, actor, 99, ctr)
    with torch.no_grad():
        ref = actor(obs)
    assert float((logp - ref).abs().max()) <= 2e-4
    assert int(ctr) == 9
    c2 = torch.tensor([8], dtype=torch.int64)
    assert torch.equal(acts.cpu(), _sample(oracle_env.library(), logp.cpu().contiguous(), 99, c2))
    acts2, _ = _policy_act(hip_library(), obs, actor, 99, ctr, want_logp=False)  # counter advanced: new draw
    assert not torch.equal(acts, acts2) or n < 64


def test_interact_fused_policy_path_is_reproducible():
    """Default path (no torch generator, reference ActorNetwork): DeviceRollout.act is one mm_policy_act call;
    two rollouts built the same way draw the same actions, and a different sample_seed draws others."""
    def make(seed):
        torch.manual_seed(0)
        env = oracle_env.OracleEnv(16, 4, env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-avs_cint", "HEADWAY_TIME": 0.5},
                                   cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=3, auto_reset=True)
        ro = DeviceRollout(env, ActorNetwork(30, 128, 5), CriticNetwork(30, 5, 128), roll_out_n_steps=12, sample_seed=seed)
        assert ro.fused_policy
        return ro.interact()
    a, b, c = make(1), make(1), make(2)
    assert torch.equal(a["actions"], b["actions"]) and torch.equal(a["returns"], b["returns"])
    assert not torch.equal(a["actions"], c["actions"])
    assert len(torch.unique(a["actions"])) > 1


def test_rollout_checkpoint_resume():
    """DeviceRollout.state_dict / load_state_dict: the resumed stream draws the same actions and returns."""
    def make():
        torch.manual_seed(0)
        env = oracle_env.OracleEnv(8, 4, env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
                                   cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=3, auto_reset=True)
        return DeviceRollout(env, ActorNetwork(30, 128, 5), CriticNetwork(30, 5, 128), roll_out_n_steps=7, sample_seed=5)
    a = make()
    a.interact()
    ck = a.state_dict()
    ref = a.interact()
    b = make()
    b.load_state_dict(ck)
    out = b.interact()
    for k in ("states", "actions", "returns", "dones"):
        assert torch.equal(out[k], ref[k]), k


@pytest.mark.parametrize("greedy", [True, False], ids=["greedy", "sampling"])
def test_evaluate_leaves_the_training_stream_untouched(greedy):
    """The reference evaluates on a separate env_eval (run_mappo.py:300-306): the training env keeps its seed sequence
    and its in-progress episodes.  DeviceRollout.evaluate borrows the batch, so afterwards every plane, counter, per-env
    RNG seed, the carried observation and the SAMPLER's counter must be exactly what they were -- and the next rollout
    must be the one a twin that never evaluated produces, with a deterministic and with a sampling actor."""
    def make():
        torch.manual_seed(0)
        env = oracle_env.OracleEnv(4, 4, env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
                                   cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=3, auto_reset=True)

        class GreedyActor(ActorNetwork):  # deterministic actions: this leg does not depend on the sampler's counter
            def forward(self, state):
                lp = super().forward(state)
                return torch.where(lp == lp.max(-1, keepdim=True).values, 0.0, -float("inf")).to(lp.dtype)
        actor = GreedyActor(30, 128, 5) if greedy else ActorNetwork(30, 128, 5)  # (sampling: Philox keyed by the counter)
        return DeviceRollout(env, actor, CriticNetwork(30, 5, 128), roll_out_n_steps=15, sample_seed=11)
    a, b = make(), make()
    a.interact(); b.interact()
    before = a.env.state.clone()
    a.evaluate(seeds=[7, 8, 9, 10])
    assert torch.equal(a.env.state, before), "state planes / counters / seeds changed by evaluate()"
    assert torch.equal(a.env.seeds, b.env.seeds) and torch.equal(a.env.env_i32, b.env.env_i32)
    assert torch.equal(a.obs, b.obs) and torch.equal(a._sample_counter, b._sample_counter)
    ra, rb = a.interact(), b.interact()
    for k in ("states", "actions", "returns", "dones"):
        assert torch.equal(ra[k], rb[k]), k
    a.evaluate(seeds=[7, 8, 9, 10])  # and a second evaluation does not replay the first post-evaluation spawn either
    ra, rb = a.interact(), b.interact()
    assert torch.equal(ra["states"], rb["states"]) and torch.equal(ra["actions"], rb["actions"])


@pytest.mark.gpu
def test_graph_rollout_survives_evaluate_and_reseed():
    """hipGraph replay interleaved with evaluate() and a load_state_dict that changes the sampler seed: the static carry
    buffer is refreshed in place and the graph is re-captured when a baked-in argument changed (eager twin == graph)."""
    from marl_mass_amd import VecMergeEnv
    E, N, T = 2048, 8, 12
    kw = dict(config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=9, auto_reset=True)
    torch.manual_seed(3)
    actor, critic = ActorNetwork(30, 128, 5).cuda(), CriticNetwork(30, 5, 128).cuda()
    eager = DeviceRollout(VecMergeEnv(E, N, **kw), actor, critic, roll_out_n_steps=T, sample_seed=4)
    graph = DeviceRollout(VecMergeEnv(E, N, **kw), actor, critic, roll_out_n_steps=T, sample_seed=4, use_graph=True)
    graph.interact()                    # warm-up + capture + first replay = 2 rollouts
    eager.interact(); eager.interact()
    seeds = list(range(100, 100 + E))
    re_, rg = eager.evaluate(seeds=seeds), graph.evaluate(seeds=seeds)
    assert torch.equal(re_[0].nan_to_num(), rg[0].nan_to_num())
    a, b = eager.interact(), graph.interact()
    torch.cuda.synchronize()
    for k in ("states", "actions", "returns", "dones"):
        assert torch.equal(a[k], b[k]), ("after evaluate", k)
    ck = eager.state_dict()
    ck["sample_seed"] = 77              # a new sampler seed must reach the captured kernels too
    eager.load_state_dict(ck); graph.load_state_dict(ck)
    a, b = eager.interact(), graph.interact()
    torch.cuda.synchronize()
    for k in ("states", "actions", "returns", "dones"):
        assert torch.equal(a[k], b[k]), ("after reseed", k)


def test_evaluate_refuses_an_env_without_the_outputs_it_reads():
    """MAPPO.evaluation reads the vehicles' speed / position and the crash flag of every step (marl/mappo.py:300-330): a
    batch built with those outputs skipped cannot be evaluated, and says so before it touches the batch."""
    env = oracle_env.OracleEnv(2, 4, env_id="merge-multi-agent-v1", config={"safety_guarantee": "none"}, seed=3,
                               auto_reset=True, skip_outputs=("agents_info",))
    ro = DeviceRollout(env, ActorNetwork(30, 128, 5), CriticNetwork(30, 5, 128), roll_out_n_steps=5, sample_seed=1)
    before = env.state.clone()
    with pytest.raises(ValueError, match="agents_info"):
        ro.evaluate(seeds=[1, 2])
    assert torch.equal(env.state, before) and env.auto_reset


def test_step_writes_requested_outputs_into_caller_slots():
    """step(out={key: tensor}): that step writes those outputs into the caller's tensors (a rollout's rewards[t], dones[t],
    ...) and nowhere else; everything else -- state, the other outputs, later steps -- is what a twin without slots gives."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact",
              cbf_tau=0.5, seed=4, auto_reset=True)
    a_env, b_env = oracle_env.OracleEnv(6, 4, **kw), oracle_env.OracleEnv(6, 4, **kw)
    a_env.reset(); b_env.reset()
    g = torch.Generator().manual_seed(1)
    T = 5
    rew = torch.full((T, 6, 4), -9.0, dtype=torch.float64)
    don = torch.full((T, 6), 7, dtype=torch.uint8)
    for t in range(T):
        act = torch.randint(0, 5, (6, 4), generator=g, dtype=torch.int32)
        before = b_env.out["regional_rewards"].clone()
        _, ra, da, ia = a_env.step(act)
        _, rb, db, ib = b_env.step(act, out={"regional_rewards": rew[t], "done": don[t]})
        assert torch.equal(rew[t], ia["regional_rewards"])  # what went through step(out=...) equals the twin's own outputs
        assert torch.equal(don[t], da)
        assert db.data_ptr() == don[t].data_ptr()
        assert ib["regional_rewards"].data_ptr() == rew[t].data_ptr()
        assert torch.equal(b_env.out["regional_rewards"], before), "the env's own buffer was written although a slot was given"
        assert torch.equal(ra, rb) and torch.equal(a_env.state, b_env.state) and torch.equal(ia["agents_rewards"], ib["agents_rewards"])
    _, _, d2, i2 = b_env.step(act)  # without slots the env's own buffers are the outputs again
    assert d2.data_ptr() == b_env.out["done"].data_ptr() and i2["regional_rewards"].data_ptr() == b_env.out["regional_rewards"].data_ptr()
    with pytest.raises(KeyError):
        b_env.step(act, out={"no_such_output": rew[0]})
    with pytest.raises(AssertionError):
        b_env.step(act, out={"done": torch.zeros(6, dtype=torch.float64)})


def _discount_case(seed, T=37, E=9, N=5):
    g = torch.Generator().manual_seed(seed)
    rewards = (torch.rand(T, E, N, dtype=torch.float64, generator=g) - 0.3) * 7
    dones = (torch.rand(T, E, generator=g) < 0.08).to(torch.uint8)
    final = torch.randn(E, N, dtype=torch.float64, generator=g)
    return rewards, dones, final


@pytest.mark.parametrize("scale", [20.0, 0.0])
def test_discount_returns_entry_equals_the_torch_chain(scale):
    """mm_discount_returns (CPU twin here, HIP in the GPU leg) == the reward scaling + discount_rewards chain of torch ops, bit
    for bit (same operations in the same order: r / scale, running * gamma + r), incl. a `done` at the last step."""
    from marl_mass_amd.rollout import discount_rewards
    rewards, dones, final = _discount_case(3)
    dones[-1, 0] = 1
    want = discount_rewards(rewards / scale if scale > 0 else rewards, dones, final, 0.99)
    clib = oracle_env.library()
    got = torch.empty_like(rewards)
    assert clib.lib.mm_discount_returns(rewards.data_ptr(), dones.data_ptr(), final.data_ptr(), rewards.shape[0], rewards.shape[1],
                                        rewards.shape[2], 0.99, scale, got.data_ptr(), None) == 0
    assert torch.equal(got, want)
    assert clib.lib.mm_discount_returns(None, dones.data_ptr(), final.data_ptr(), 1, 1, 1, 0.99, scale, got.data_ptr(), None) != 0


@pytest.mark.gpu
def test_discount_returns_entry_on_gpu():
    from marl_mass_amd import hip_library
    from marl_mass_amd.rollout import discount_rewards
    rewards, dones, final = _discount_case(5, T=100, E=777, N=8)
    want = discount_rewards(rewards / 20.0, dones, final, 0.99)
    r, d, f = rewards.cuda(), dones.cuda(), final.cuda()
    got = torch.empty_like(r)
    assert hip_library().lib.mm_discount_returns(r.data_ptr(), d.data_ptr(), f.data_ptr(), 100, 777, 8, 0.99, 20.0, got.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    # in place (returns may alias rewards)
    assert hip_library().lib.mm_discount_returns(r.data_ptr(), d.data_ptr(), f.data_ptr(), 100, 777, 8, 0.99, 20.0, r.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(r.cpu(), want)
