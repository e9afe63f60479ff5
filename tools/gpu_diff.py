"""Debug helper (GPU box): run HIP vs oracle on a random rollout and print the first divergence."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import torch, numpy as np
import oracle_env
from marl_mass_amd import VecMergeEnv, _cabi as abi

env_id, safety, N, E, steps, eta, tau = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), float(sys.argv[7])
n_hdv = int(sys.argv[8]) if len(sys.argv) > 8 else 0
oracle_env.set_math_mode(1)
kw = dict(env_id=env_id, config={"safety_guarantee": safety, "HEADWAY_TIME": tau}, cbf_eta=eta, cbf_tau=tau,
          obs_f64=True, seed=1000, auto_reset=True, trace=True, n_hdv=n_hdv, qp_solver=os.environ.get("MM_QP_SOLVER", "exact"))
import os
if os.environ.get("MM_DENSITY"):  # ragged batch: vehicle counts drawn per episode
    kw["config"].update({"traffic_density": int(os.environ["MM_DENSITY"]), "traffic_type": "cav", "mixed_traffic": False})
    kw["draw_counts"] = True
gpu, cpu = VecMergeEnv(E, N, device="cuda:0", debug_flags=int(os.environ.get("MM_DEBUG_FLAGS", "0")), **kw), oracle_env.OracleEnv(E, N, **kw)
gpu.reset(); cpu.reset()
g = torch.Generator().manual_seed(123)
p = torch.tensor([float(x) for x in os.environ.get("MM_ACTION_P", "0.1,0.6,0.1,0.1,0.1").split(",")])
for t in range(steps):
    a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
    sg, sc = gpu.f64.cpu().clone(), cpu.f64.clone()
    gpu.step(a.cuda()); cpu.step(a)
    du = (gpu.u8.cpu() != cpu.u8)
    df = (gpu.f64.cpu() - cpu.f64).nan_to_num().abs()
    if du.any() or df.max() > 0:
        idx = du.nonzero() if du.any() else (df > 0).nonzero()
        print("step", t, "first mismatches (plane, env, agent):", idx[:8].tolist(), "max float diff", float(df.max()))
        e = int(idx[0][1])
        print("planes:", [abi.B_PLANES[int(i[0])] if du.any() else abi.F_PLANES[int(i[0])] for i in idx[:8]])
        np.set_printoptions(precision=12, linewidth=200)
        tg, tc = gpu.trace[:, :, e].cpu().numpy(), cpu.trace[:, :, e].numpy()
        print("actions", a[e].tolist())
        print("pre-step state diff max", float((sg[:, e] - sc[:, e]).nan_to_num().abs().max()))
        for k in range(3):
            for name in abi.T_PLANES:
                gg, cc = tg[k, abi.T[name]], tc[k, abi.T[name]]
                if not np.array_equal(np.nan_to_num(gg), np.nan_to_num(cc)):
                    print(" sub", k, name, "\n   gpu", gg, "\n   cpu", cc)
        print("x (cpu) per substep:\n", tc[:, abi.T["X"]], "\nlane\n", tc[:, abi.T["LANE"]], "\ny\n", tc[:, abi.T["Y"]])
        break
else:
    print("no divergence in", steps, "steps")
