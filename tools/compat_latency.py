"""Latency of the drop-in object API (MergeEnvCompat, E = 1) on the GPU: what `env = make(env_id)` costs per env.step."""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_mass_amd import compat
import numpy as np
compat.CBFType.GAMMA_B, compat.CBFType.TAU = 0.03125, 0.5
env = compat.make("merge-multi-agent-v1")
env.config.update({"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5, "traffic_type": "cav", "mixed_traffic": False})
env._num_vehicles = lambda num_CAV=0: (8, 0)
obs, _ = env.reset(is_training=False, testing_seeds=0)
rs = np.random.RandomState(0)
n = 0; t0 = time.perf_counter()
for ep in range(3):
    done = False
    while not done:
        obs, r, done, info = env.step(tuple(rs.choice(5, size=8, p=[.1,.6,.1,.1,.1])))
        n += 1
    env.reset(is_training=False, testing_seeds=ep + 1)
dt = time.perf_counter() - t0
print("compat adapter (E=1, 8 CAVs, MASS): %.2f ms per env.step over %d steps" % (dt / n * 1e3, n))
