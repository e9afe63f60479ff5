"""Rollout-counterpart timing: policy forward + sampling + mm_step + bookkeeping per policy step."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from marl_mass_amd import VecMergeEnv
from marl_mass_amd.rollout import ActorNetwork, CriticNetwork, DeviceRollout

E, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 8, 50
kw = dict(config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=9, auto_reset=True)
torch.manual_seed(0)
actor, critic = ActorNetwork(30, 128, 5).cuda(), CriticNetwork(30, 5, 128).cuda()
for graph in (False, True):
    ro = DeviceRollout(VecMergeEnv(E, N, **kw), actor, critic, roll_out_n_steps=T, use_graph=graph)
    ro.interact(); ro.interact(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ro.interact(); ro.interact(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print("graph=%d: rollout %d steps x %d envs x %d: %.2f ms = %.3f ms/step, %.3g agent-steps/s" % (graph, T, E, N, dt * 1e3, dt * 1e3 / T, E * N * T / dt))
env = VecMergeEnv(E, N, **kw); obs, _ = env.reset()
x = obs.reshape(E * N, 30).float()
for name, fn in (("actor forward", lambda: actor(x)), ("act (forward + sample)", lambda: ro.act(obs)),
                 ("mm_step", lambda: env.step(torch.ones(E, N, dtype=torch.int32, device="cuda")))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print("%-24s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
