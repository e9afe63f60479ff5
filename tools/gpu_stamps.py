"""Diagnostic: per-phase cycle shares of step_kernel (build with `make tuning EXTRA="-DMM_STAMPS -DMM_ONLY_G=8 -DMM_ONLY_MIXED=false"`, copy to libmm_hip_stamps.so)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["MM_HIP_LIB"] = os.path.join(REPO, "marl-mass_amd", "csrc", "libmm_hip_stamps.so")
import torch
from marl_mass_amd import VecMergeEnv, hip_library
shield = sys.argv[1] if len(sys.argv) > 1 else "cbf-cav"
E, N = 65536, 8
env = VecMergeEnv(E, N, config={"safety_guarantee": shield, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, cbf_tau=0.5, seed=1000, auto_reset=True)
env.reset()
g = torch.Generator(device="cuda:0").manual_seed(123)
p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device="cuda:0")
ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(8)]
lib = hip_library().lib
buf = (ctypes.c_ulonglong * 16)()
for t in range(10): env.step(ring[t % 8])
torch.cuda.synchronize(); lib.mm_debug_read_stamps(buf, 1)
K = 50
for t in range(K): env.step(ring[t % 8])
torch.cuda.synchronize(); lib.mm_debug_read_stamps(buf, 1)
names = ["load+setup", "act", "predict A", "S1 classify", "select+rounds", "lazy B", "sweep exit/serial", "commit", "collisions", "trace+terminal", "rewards+outputs", "respawn+store", "observation"]
tot = sum(buf[:13])
waves = E * 8 / 64
for k, nme in enumerate(names):
    print("%-20s %6.2f %%   %8.0f cycles/wave/step" % (nme, 100.0 * buf[k] / tot, buf[k] / waves / K))
print("total %.0f cycles/wave/step" % (tot / waves / K))
print("shielded wave-sub-steps %d, of which serial fallback %d (%.2f %%); irregular lanes %d" % (buf[13], buf[14], 100.0 * buf[14] / max(buf[13], 1), buf[15]))
