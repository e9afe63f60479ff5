import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import torch, numpy as np
import oracle_env
from marl_mass_amd import VecMergeEnv, _cabi as abi
oracle_env.set_math_mode(1)
kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
          cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=31, auto_reset=True, n_hdv=3)
E, N = 512, 8
gpu, cpu = VecMergeEnv(E, N, device="cuda:0", **kw), oracle_env.OracleEnv(E, N, **kw)
gpu.reset(); cpu.reset()
g = torch.Generator().manual_seed(3)
np.set_printoptions(precision=10, linewidth=200)
for t in range(30):
    a = torch.randint(0, 5, (E, N), generator=g, dtype=torch.int32)
    gpu.step(a.cuda()); cpu.step(a)
    if t % 5 == 4:
        steer = (torch.rand(E, N, dtype=torch.float64, generator=g) - 0.5) * 0.2
        acc = (torch.rand(E, N, dtype=torch.float64, generator=g) - 0.5) * 12
        rg, rc = gpu.shield_actions(steer, acc), cpu.shield_actions(steer, acc)
        for name, x, y in zip(("steer", "acc", "status", "margin"), rg, rc):
            d = (x.cpu().nan_to_num() != y.nan_to_num())
            if d.any():
                idx = d.nonzero()[:5]
                print("t", t, name, "mismatches", int(d.sum()), idx.tolist())
                e = int(idx[0][0])
                print(" gpu", x[e].cpu().numpy(), "\n cpu", y[e].numpy())
                print(" kind", cpu.u8[abi.B["KIND"], e].numpy(), "lane", cpu.u8[abi.B["LANE"], e].numpy(), "crashed", cpu.u8[abi.B["CRASHED"], e].numpy(), "hist", cpu.u8[abi.B["HIST_LEN"], e].numpy())
                print(" x", cpu.f64[abi.F["X"], e].numpy(), "\n y", cpu.f64[abi.F["Y"], e].numpy())
                print(" status gpu", rg[2][e].cpu().numpy(), "cpu", rc[2][e].numpy())
        if any((x.cpu().nan_to_num() != y.nan_to_num()).any() for x, y in zip(rg, rc)):
            break
