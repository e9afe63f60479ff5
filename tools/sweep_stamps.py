"""Diagnostic: where a wave of the lane-per-env sweep kernel spends its cycles (build: `make tuning EXTRA="-DMM_STAMPS -DMM_ONLY_G=8"`,
copied to libmm_hip_stamps.so).  Per wave and sweep launch: cycles in the stopping test, post / publish, setup of the next ego,
the interior-point iteration; trips of the wave loop and trips that ran a setup."""
import ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["MM_HIP_LIB"] = os.path.join(REPO, "marl-mass_amd", "csrc", "libmm_hip_stamps.so")
import torch
from marl_mass_amd import VecMergeEnv, hip_library, _cabi as abi
shield = sys.argv[1] if len(sys.argv) > 1 else "cbf-cav"
E, N = int(os.environ.get("MM_STAMPS_E", "65536")), 8
env = VecMergeEnv(E, N, config={"safety_guarantee": shield, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, cbf_tau=0.5, seed=1000, auto_reset=True, qp_solver="ipm", debug_flags=int(os.environ.get("MM_DEBUG_FLAGS", "0")))
env.reset()
env.env_i32[abi.EP["STEPS"]] = ((torch.arange(E, device="cuda:0") * 37) % 100).to(torch.int32)
g = torch.Generator(device="cuda:0").manual_seed(123)
p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device="cuda:0")
ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(8)]
lib = hip_library().lib
buf = (ctypes.c_ulonglong * 8)()
for t in range(100): env.step(ring[t % 8])
torch.cuda.synchronize(); lib.mm_debug_read_sweep_stamps(buf, 1)
K = 20
for t in range(K): env.step(ring[t % 8])
torch.cuda.synchronize(); lib.mm_debug_read_sweep_stamps(buf, 1)
launches = (E // 64) * K * 3
names = ["stopping test", "post + publish", "setup", "iteration", "trips", "setup trips", "re-swept envs (failed assumption)", "QPs handed to the verification queue"]
out = {n: buf[k] / launches for k, n in enumerate(names)}
out["unit"] = "s_memtime cycles (100 MHz ticks) resp. counts, per wave and sweep launch"
print(json.dumps(out, indent=1))
