#!/usr/bin/env python3
"""Drop-in proof with the reference's OWN caller (build container only; the reference cannot travel).

Imports the reference's marl/mappo.py (MAPPO, ActorNetwork / CriticNetwork) from /root/reference under the stand-ins
of tools/refshim and drives `MAPPO.interact()` x K + `MAPPO.evaluation()` under fixed seeds twice:

  (a) on the reference's env:  env = gym.make(env_id)               + the config writes of run_mappo.py:137-171
  (b) on this repo's drop-in:  env = marl_mass_amd.compat.make(...)  + the SAME writes (CPU oracle backend here)

with the same network initialisation, and asserts that the two runs are the same rollout: identical action sequence
(the policy samples from the global numpy stream that reset() seeds, so this holds only if every reset consumed the
same draws and every state the actor saw was the same), states / rewards / returns / ext_info equal to rounding.
Then it writes what run (a) produced as a fixture (tests/golden/mappo_dropin_*.npz):

  * actor / critic weights, the states MAPPO stored, the actions it took, the discounted returns it pushed to memory
    (marl/mappo.py:152-158 via _discount_reward :364-370, with the critic bootstrap), episode boundaries;
  * the actor's log-probabilities on those states (Model_common.py:5-22) -- the f1 pin of SURVEY 8f-1;
  * evaluation(): rewards per step, ext_info (steps, avg_speeds, crash_count, min_headway, traffic_speeds, merge_percents).

tests/test_mappo_dropin.py replays the fixture on the drop-in (oracle backend on CPU, HIP backend on the GPU box).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference", REPO, os.path.join(REPO, "oracle")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import _refcompat  # noqa: E402,F401
import gym  # noqa: E402
import highway_env  # noqa: E402,F401
import cvxopt  # noqa: E402
from highway_env.vehicle.safety.cbf import CBFType as RefCBFType  # noqa: E402
from marl.mappo import MAPPO  # noqa: E402  (the reference's caller)

import oracle_env  # noqa: E402
from marl_mass_amd import compat  # noqa: E402

OUT = os.environ.get("MM_GOLDEN_OUT", os.path.join(REPO, "tests", "golden"))
ENV_KEYS = dict(simulation_frequency=15, duration=20, policy_frequency=5, COLLISION_REWARD=200, HIGH_SPEED_REWARD=1,
                HEADWAY_COST=4, MERGING_LANE_COST=4, traffic_density=1, action_masking=False, lateral_control="steer",
                mixed_traffic=False, traffic_type="cav", agent_reward="default")


def configure(env, shield, tau, seed):
    """run_mappo.py:137-171, 282-283: config written AFTER construction, seed set on the env object."""
    for k, v in ENV_KEYS.items():
        env.config[k] = v
    env.config["HEADWAY_TIME"] = tau
    env.config["safety_guarantee"] = shield
    env.config["seed"] = seed
    env.seed = seed
    if hasattr(env, "unwrapped"):
        env.unwrapped.seed = seed
    return env


def drive(make_env, shield, tau, eta, env_id, K, T, test_seeds):
    """The reference's training / evaluation calls, verbatim in order (run_mappo.py:233,290-306)."""
    RefCBFType.GAMMA_B, RefCBFType.TAU = eta, tau
    compat.CBFType.GAMMA_B, compat.CBFType.TAU = eta, tau
    cvxopt.solvers.mode = "exact"
    env = configure(make_env(env_id), shield, tau, seed=0)
    env_eval = configure(make_env(env_id), shield, tau, seed=0)
    torch.manual_seed(1234)
    mappo = MAPPO(env=env, state_dim=env.n_s, action_dim=env.n_a, memory_capacity=10000, roll_out_n_steps=T,
                  reward_gamma=0.99, reward_scale=20.0, use_cuda=False, traffic_density=1, reward_type="regionalR",
                  test_seeds=",".join(str(s) for s in test_seeds), max_steps=None)
    rollouts = []
    for _ in range(K):
        n_before = len(mappo.memory.memory)
        mappo.interact()
        exps = mappo.memory.memory[n_before:]
        rollouts.append(dict(states=np.array([e.states for e in exps], dtype=np.float64),
                             actions=np.array([e.actions for e in exps], dtype=np.float64).argmax(-1).astype(np.int32),
                             returns=np.array([e.rewards for e in exps], dtype=np.float64),
                             episode_done=bool(mappo.episode_done), n_agents=int(mappo.n_agents)))
    rewards, (vspeed, vpos), ext = mappo.evaluation(env_eval, None, eval_episodes=len(test_seeds), is_train=False)
    weights = {("actor." + k): v.numpy().copy() for k, v in mappo.actor.state_dict().items()}
    weights.update({("critic." + k): v.numpy().copy() for k, v in mappo.critic.state_dict().items()})
    with torch.no_grad():
        logp = [mappo.actor(torch.tensor(r["states"], dtype=torch.float32).reshape(-1, env.n_s)).numpy().reshape(
            r["states"].shape[0], r["n_agents"], env.n_a) for r in rollouts]
    return dict(rollouts=rollouts, logp=logp, eval_rewards=[np.array(r, dtype=np.float64) for r in rewards],
                eval_vspeed=[np.array(v[-1] if len(np.shape(v)) > 1 else v, dtype=np.float64) for v in vspeed],
                ext={k: (float(v) if np.isscalar(v) else [float(x) for x in v]) for k, v in ext.items() if k != "step_time"},
                weights=weights, n_s=int(env.n_s), n_a=int(env.n_a))


def compare(a, b):
    worst = 0.0
    assert len(a["rollouts"]) == len(b["rollouts"])
    for ra, rb in zip(a["rollouts"], b["rollouts"]):
        assert ra["n_agents"] == rb["n_agents"] and ra["episode_done"] == rb["episode_done"], "different episode structure"
        assert ra["states"].shape == rb["states"].shape, (ra["states"].shape, rb["states"].shape)
        assert np.array_equal(ra["actions"], rb["actions"]), "the policy drew different actions: the RNG streams / states diverged"
        worst = max(worst, float(np.abs(ra["states"] - rb["states"]).max()), float(np.abs(ra["returns"] - rb["returns"]).max()))
    for x, y in zip(a["eval_rewards"], b["eval_rewards"]):
        assert x.shape == y.shape
        worst = max(worst, float(np.abs(x - y).max()))
    for k in a["ext"]:
        worst = max(worst, float(np.abs(np.array(a["ext"][k], dtype=float) - np.array(b["ext"][k], dtype=float)).max()))
    return worst


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = [("mass", "merge-multi-agent-v1", "cbf-cav", 0.5, 0.03125), ("hss", "merge-multi-agent-v1", "cbf-avs_cint", 0.5, 0.03125),
             ("v0none", "merge-multi-agent-v0", "none", 1.2, 0.0)]
    for tag, env_id, shield, tau, eta in cases:
        K, T, test_seeds = 6, 40, [0, 25, 50]
        ref = drive(lambda eid: gym.make(eid), shield, tau, eta, env_id, K, T, test_seeds)
        dropin = drive(lambda eid: compat.make(eid, backend_factory=lambda **kw: oracle_env.OracleEnv(**kw)), shield, tau, eta,
                       env_id, K, T, test_seeds)
        worst = compare(ref, dropin)
        assert worst <= 1e-9, worst
        meta = dict(env_id=env_id, shield=shield, headway_time=tau, eta=eta, K=K, roll_out_n_steps=T, test_seeds=test_seeds,
                    torch_seed=1234, env_seed=0, reward_type="regionalR", reward_scale=20.0, reward_gamma=0.99,
                    n_s=ref["n_s"], n_a=ref["n_a"], env_config=ENV_KEYS,
                    dropin_vs_reference_max_abs=worst,
                    note="recorded from the reference's MAPPO on the reference env; at generation time the same MAPPO on "
                         "marl_mass_amd.compat (oracle backend) drew the identical action sequence and matched to %.1e" % worst)
        arrays = {}
        for k, r in enumerate(ref["rollouts"]):
            arrays["ro%d_states" % k], arrays["ro%d_actions" % k], arrays["ro%d_returns" % k] = r["states"], r["actions"], r["returns"]
            arrays["ro%d_logp" % k] = ref["logp"][k].astype(np.float32)
            arrays["ro%d_done" % k] = np.array(r["episode_done"])
        for k, r in enumerate(ref["eval_rewards"]):
            arrays["ev%d_rewards" % k] = r
        for k, v in ref["weights"].items():
            arrays["w_" + k] = v
        np.savez_compressed(os.path.join(OUT, "mappo_dropin_%s.npz" % tag), meta=json.dumps(meta), ext=json.dumps(ref["ext"]), **arrays)
        print("%-8s %s %s: %d rollouts (agents %s), eval steps %s, drop-in == reference to %.2e, identical actions"
              % (tag, env_id, shield, K, [r["n_agents"] for r in ref["rollouts"]], ref["ext"]["steps"], worst))


if __name__ == "__main__":
    main()
