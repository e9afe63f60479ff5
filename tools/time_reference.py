#!/usr/bin/env python3
"""Times the REFERENCE's own CPU path (build container only: /root/reference does not exist on the GPU box).

Regenerates BASELINE.md section 2 as data: the reference Python imported from /root/reference under the stand-ins of
tools/refshim (gym / pygame stubs, numpy+pandas compat, the cvxopt stand-in -- see its docstring), one Python thread
(the reference is single-threaded), timed around env.step exactly where the reference times it itself
(marl/mappo.py:308-311).  Writes profiles/reference_cpu.json; bench.py attaches it to `cpu_baseline.reference_python`.

    python tools/time_reference.py
"""
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference"]

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import _refcompat  # noqa: E402,F401
import gym  # noqa: E402
import highway_env  # noqa: E402,F401
import cvxopt  # noqa: E402
from highway_env.vehicle.safety.cbf import CBFType  # noqa: E402


def run(env_id, shield, n_cav, tau, eta, qp, episodes, budget_s=25.0):
    CBFType.GAMMA_B, CBFType.TAU = eta, tau
    cvxopt.solvers.mode = qp
    env = gym.make(env_id)
    env.config.update({"simulation_frequency": 15, "duration": 20, "policy_frequency": 5, "COLLISION_REWARD": 200,
                       "HIGH_SPEED_REWARD": 1, "HEADWAY_COST": 4, "HEADWAY_TIME": tau, "MERGING_LANE_COST": 4,
                       "traffic_density": 1, "action_masking": False, "safety_guarantee": shield, "lateral_control": "steer",
                       "mixed_traffic": False, "traffic_type": "cav", "agent_reward": "default"})
    env._num_vehicles = lambda num_CAV=0: (n_cav, 0)  # BASELINE configs fix the vehicle count
    rng = np.random.RandomState(123)
    steps, crashes, t_step, t0 = 0, 0, 0.0, time.perf_counter()
    for ep in range(episodes):
        env.reset(is_training=False, testing_seeds=ep)
        done = False
        while not done:
            a = tuple(int(x) for x in rng.choice(5, size=n_cav, p=[0.1, 0.6, 0.1, 0.1, 0.1]))
            t1 = time.perf_counter()
            _, _, done, _ = env.step(a)
            t_step += time.perf_counter() - t1
            steps += 1
        crashes += int(env.is_crashed())
        if time.perf_counter() - t0 > budget_s:
            episodes = ep + 1
            break
    cvxopt.solvers.mode = "exact"
    return dict(env_id=env_id, safety_guarantee=shield, n_cav=n_cav, qp_stand_in=qp, episodes=episodes, env_steps=steps,
                crashed_episodes=crashes, ms_per_env_step=1e3 * t_step / steps, agent_steps_per_s=steps * n_cav / t_step)


def main():
    rows = [run("merge-multi-agent-v0", "none", 4, 1.2, 0.0, "exact", 10),
            run("merge-multi-agent-v1", "cbf-avs_cint", 4, 0.5, 0.03125, "exact", 10),
            run("merge-multi-agent-v1", "cbf-cav", 8, 0.5, 0.03125, "exact", 5),
            run("merge-multi-agent-v1", "cbf-cav", 8, 0.5, 0.03125, "coneqp", 5)]
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    out = dict(
        what="reference hkbharath/MARL-MASS env.step on the CPU, reset excluded, random action tape p=[.1,.6,.1,.1,.1] seed 123",
        host=dict(cpu_model=cpu[0] if cpu else platform.processor(), logical_cpus=os.cpu_count(), threads_used=1,
                  python=platform.python_version(), numpy=np.__version__, pandas=pd.__version__),
        caveats=["runs under tools/refshim: stub gym / pygame, numpy+pandas API compat (the reference pins numpy 1.19 / pandas 1.1)",
                 "cvxopt 1.2.7 is not installable here: solvers.qp is the closed-form KKT point ('exact') or the pure-Python "
                 "restatement of coneqp ('coneqp', slower than the real C/BLAS solver would be) -- shield rows are indicative",
                 "measured in the build container, not on the GPU box: the reference cannot travel"],
        rows=rows)
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", "reference_cpu.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for r in rows:
        print("%-24s %-14s N=%d qp=%-7s %6.1f ms/env-step  %6.1f agent-steps/s  (%d steps, %d crashed)" % (
            r["env_id"], r["safety_guarantee"], r["n_cav"], r["qp_stand_in"], r["ms_per_env_step"], r["agent_steps_per_s"],
            r["env_steps"], r["crashed_episodes"]))


if __name__ == "__main__":
    main()
