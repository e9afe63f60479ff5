"""Drop-in proof, replayed: what the reference's OWN MAPPO (marl/mappo.py) produced on the reference env is
reproduced by this repo's `compat.make(env_id)` object driven the way MAPPO drives an env.

The fixtures (tests/golden/mappo_dropin_*.npz) were recorded by tools/gen_mappo_dropin.py, which imports the reference's
marl/mappo.py in the build container, runs `MAPPO.interact()` x K and `MAPPO.evaluation()` on `gym.make(env_id)`, and
-- before writing anything -- runs the SAME MAPPO object code on `marl_mass_amd.compat.make(env_id)` and asserts that
both produce the identical action sequence and states / returns / ext_info equal to rounding (meta:
dropin_vs_reference_max_abs).  Here, without the reference, the loop below restates `interact` / `evaluation`
(marl/mappo.py:102-158, 255-361) and replays the fixture on the oracle backend (CPU) and on the HIP backend (GPU):
the f1 pins of SURVEY 8f-1 -- the actor's log-probabilities on a fixed checkpoint (Model_common.py:5-22), the
sampled actions (`np.random.choice` on the global stream that reset() seeds, :220-236) and the discounted returns
with the critic bootstrap (`_discount_reward`, :364-370).
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle_env
from golden_util import GOLDEN
from marl_mass_amd import compat
from marl_mass_amd.rollout import ActorNetwork, CriticNetwork

CASES = ["mass", "hss", "v0none"]


def _load(tag):
    z = np.load(os.path.join(GOLDEN, "mappo_dropin_%s.npz" % tag))
    return z, json.loads(str(z["meta"])), json.loads(str(z["ext"]))


def _nets(z, meta):
    actor, critic = ActorNetwork(meta["n_s"], 128, meta["n_a"]), CriticNetwork(meta["n_s"], meta["n_a"], 128)
    actor.load_state_dict({k[len("w_actor."):]: torch.tensor(z[k]) for k in z.files if k.startswith("w_actor.")})
    critic.load_state_dict({k[len("w_critic."):]: torch.tensor(z[k]) for k in z.files if k.startswith("w_critic.")})
    return actor, critic


def _make_env(meta, factory):
    compat.CBFType.GAMMA_B, compat.CBFType.TAU, compat.CBFType.QP_SOLVER = meta["eta"], meta["headway_time"], "exact"
    env = compat.make(meta["env_id"], **({"backend_factory": factory} if factory else {}))
    for k, v in meta["env_config"].items():  # run_mappo.py:145-171: written after construction
        env.config[k] = v
    env.config.update({"HEADWAY_TIME": meta["headway_time"], "safety_guarantee": meta["shield"], "seed": meta["env_seed"]})
    env.seed = meta["env_seed"]
    return env


def _softmax_actions(actor, state):
    """_softmax_action + exploration_action (marl/mappo.py:208-236): per-agent forward, np.random.choice on the GLOBAL stream."""
    with torch.no_grad():
        p = torch.exp(actor(torch.tensor(np.asarray(state), dtype=torch.float32))).numpy()
    return [int(np.random.choice(np.arange(len(pi)), p=pi)) for pi in p], p


def _replay(tag, factory):
    z, meta, ext_ref = _load(tag)
    actor, critic = _nets(z, meta)
    env = _make_env(meta, factory)
    env_state, _ = env.reset()  # MAPPO.__init__ (marl/mappo.py:44)
    gamma, scale, T = meta["reward_gamma"], meta["reward_scale"], meta["roll_out_n_steps"]
    drawn_equal = drawn_total = 0
    for k in range(meta["K"]):
        st_ref, ac_ref, ret_ref = z["ro%d_states" % k], z["ro%d_actions" % k], z["ro%d_returns" % k]
        n_agents = len(env.controlled_vehicles)
        states, rewards, done = [], [], True
        for i in range(T):  # interact(), marl/mappo.py:114-135
            states.append(env_state)
            drawn, _ = _softmax_actions(actor, env_state)
            drawn_equal += int(np.sum(np.array(drawn) == ac_ref[i])); drawn_total += n_agents
            next_state, global_reward, done, info = env.step(tuple(int(a) for a in ac_ref[i]))
            rewards.append(info["regional_rewards"])
            env_state = final_state = next_state
            if done:
                env_state, _ = env.reset()
                break
        assert len(states) == st_ref.shape[0] and done == bool(z["ro%d_done" % k]), (k, len(states), done)
        np.testing.assert_allclose(np.array(states), st_ref, rtol=0, atol=1e-9)
        if done:
            final_value = np.zeros(n_agents)
        else:  # bootstrap: action(final_state) then value (marl/mappo.py:147-150, :238-252)
            fa, _ = _softmax_actions(actor, final_state)
            one_hot = np.eye(meta["n_a"], dtype=np.float32)[fa]
            with torch.no_grad():
                final_value = critic(torch.tensor(np.asarray(final_state), dtype=torch.float32), torch.tensor(one_hot)).numpy()[:, 0]
        r = np.array(rewards) / scale
        ret = np.zeros_like(r)
        for a in range(n_agents):  # _discount_reward (marl/mappo.py:364-370)
            run = final_value[a]
            for t in reversed(range(len(r))):
                run = run * gamma + r[t, a]
                ret[t, a] = run
        np.testing.assert_allclose(ret, ret_ref, rtol=0, atol=2e-6)  # float32 critic value in the bootstrap
        with torch.no_grad():  # f1 pin: log-probabilities of the fixed checkpoint on the recorded states
            lp = actor(torch.tensor(st_ref, dtype=torch.float32).reshape(-1, meta["n_s"])).numpy().reshape(z["ro%d_logp" % k].shape)
        np.testing.assert_allclose(lp, z["ro%d_logp" % k], rtol=0, atol=2e-5)
    assert drawn_equal >= 0.98 * drawn_total, (drawn_equal, drawn_total)  # same global stream, same probabilities -> same draws
    # evaluation(), marl/mappo.py:255-361 on a second env object (run_mappo.py uses env_eval)
    ev = _make_env(meta, factory)
    steps, avg_speeds, crash, merge, tspeeds, min_headway = [], [], [], [], [], float("inf")
    for i, seed in enumerate(meta["test_seeds"]):
        state, _ = ev.reset(is_training=False, testing_seeds=seed)
        rew_ref = z["ev%d_rewards" % i]
        step, avg, tsp, done = 0, 0.0, 0.0, False
        while not done:
            acts, _ = _softmax_actions(actor, state)  # advances the global stream as the reference's evaluation does
            # the recorded evaluation drew from the same stream; stepping with our own draws keeps the test honest about
            # the whole chain (reset RNG replay -> obs -> probabilities -> draw -> step)
            state, reward, done, info = ev.step(acts)
            assert abs(reward - rew_ref[step]) <= 1e-9, (i, step, reward, rew_ref[step])
            step += 1
            avg += info["average_speed"]; tsp += info["traffic_speed"]
            min_headway = min(min_headway, info["min_headway"])
        steps.append(step); avg_speeds.append(avg / step); tspeeds.append(tsp / step); crash.append(float(ev.is_crashed()))
        merge.append(info["merge_percent"])
    assert steps == [int(s) for s in ext_ref["steps"]] and crash == [float(c) for c in ext_ref["crash_count"]]
    np.testing.assert_allclose(avg_speeds, ext_ref["avg_speeds"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(tspeeds, ext_ref["traffic_speeds"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(merge, ext_ref["merge_percents"], rtol=0, atol=1e-9)
    assert abs(min_headway - ext_ref["min_headway"]) <= 1e-9
    return meta


@pytest.mark.parametrize("tag", CASES)
def test_mappo_loop_on_dropin_oracle_backend(tag):
    meta = _replay(tag, lambda **kw: oracle_env.OracleEnv(**kw))
    assert meta["dropin_vs_reference_max_abs"] <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_mappo_loop_on_dropin_hip_backend(tag):
    """The same replay with the product backend: `env = marl_mass_amd.make(env_id)` exactly as a maintainer would swap it in."""
    _replay(tag, None)


@pytest.mark.gpu
def test_policy_kernel_logprobs_match_reference_checkpoint():
    """mm_policy_act (f32 MFMA) on the recorded states of the reference's rollout: log-probabilities of the fixed
    checkpoint as the reference's ActorNetwork computed them (Model_common.py:5-22), <= 2e-4."""
    import ctypes
    from marl_mass_amd import hip_library
    z, meta, _ = _load("mass")
    actor, _ = _nets(z, meta)
    actor = actor.cuda()
    clib = hip_library()
    for k in range(meta["K"]):
        st = torch.tensor(z["ro%d_states" % k], dtype=torch.float32).reshape(-1, meta["n_s"]).cuda().contiguous()
        n = st.shape[0]
        acts = torch.empty(n, dtype=torch.int32, device="cuda")
        logp = torch.empty(n, meta["n_a"], dtype=torch.float32, device="cuda")
        ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
        p = lambda t: t.detach().contiguous().data_ptr()  # noqa: E731
        clib.check(clib.lib.mm_policy_act(st.data_ptr(), n, meta["n_s"], p(actor.fc1.weight), p(actor.fc1.bias), p(actor.fc2.weight),
                                          p(actor.fc2.bias), p(actor.fc3.weight), p(actor.fc3.bias), 128, meta["n_a"], 1, ctr.data_ptr(),
                                          acts.data_ptr(), logp.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ref = z["ro%d_logp" % k].reshape(n, meta["n_a"])
        assert float(np.abs(logp.cpu().numpy() - ref).max()) <= 2e-4
