"""Diagnostic: per-phase cycle shares of step_kernel (build with `make tuning EXTRA="-DMM_STAMPS -DMM_ONLY_G=8 -DMM_ONLY_MIXED=false"`, copy to libmm_hip_stamps.so)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MM_HIP_LIB", os.path.join(REPO, "marl-mass_amd", "csrc", "libmm_hip_stamps.so"))
import torch
from marl_mass_amd import VecMergeEnv, hip_library
shield = sys.argv[1] if len(sys.argv) > 1 else "cbf-cav"
n_hdv = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # mixed traffic: build the stamps library with -DMM_ONLY_MIXED=true
E, N = int(os.environ.get("MM_STAMPS_E", "65536")), 8   # (MM_STAMPS_E=8192: one wave per SIMD, the latency-bound case)
metrics_on = not os.environ.get("MM_BENCH_NO_METRICS")
env = VecMergeEnv(E, N, config={"safety_guarantee": shield, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=1000, auto_reset=True, n_hdv=n_hdv)
if metrics_on: env.enable_metrics()
env.reset()
from marl_mass_amd import _cabi as abi
env.env_i32[abi.EP["STEPS"]] = ((torch.arange(E, device="cuda:0") * 37) % 100).to(torch.int32)  # stationary batch, as bench.py
g = torch.Generator(device="cuda:0").manual_seed(123)
p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device="cuda:0")
ring = [torch.multinomial(p, E * N, True, generator=g).view(E, N).int() for _ in range(8)]
lib = hip_library().lib
buf = (ctypes.c_ulonglong * 16)()
for t in range(100): env.step(ring[t % 8])
torch.cuda.synchronize(); lib.mm_debug_read_stamps(buf, 1)
K = 50
for t in range(K): env.step(ring[t % 8])
torch.cuda.synchronize(); lib.mm_debug_read_stamps(buf, 1)
names = ["load+setup", "act", "predict A", "S1 classify", "select+rounds", "lazy B", "sweep exit/serial", "commit", "collisions", "trace+terminal", "rewards+outputs", "respawn+store", "observation", "(count)", "barrier before obs", "metrics"]
tot = sum(buf[k] for k in range(16) if k != 13)
waves = E * 8 / 64
for k, nme in enumerate(names):
    if k == 13: continue
    print("%-20s %6.2f %%   %8.0f cycles/wave/step" % (nme, 100.0 * buf[k] / tot, buf[k] / waves / K))
print("total %.0f cycles/wave/step" % (tot / waves / K))
import json
out = {"workload": "%d envs x 8 CAVs, %s, stationary batch (staggered phases + 100-step pre-roll), %d steps" % (E, shield, K),
       "build": "-DMM_STAMPS -DMM_ONLY_G=8 -DMM_ONLY_MIXED=false (s_memtime stamps cost ~10 %% themselves)",
       "cycles_per_wave_step": {n: buf[k] / waves / K for k, n in enumerate(names) if k != 13}, "total_cycles_per_wave_step": tot / waves / K,
       "share": {n: buf[k] / tot for k, n in enumerate(names) if k != 13},
       "shielded_wave_substeps": int(buf[13]) >> 32, "veto_passes": int(buf[13]) & 0xFFFFFFFF,
       "veto_passes_per_wave_substep": (int(buf[13]) & 0xFFFFFFFF) / max(int(buf[13]) >> 32, 1)}
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "phase_cycles_%s%s%s.json" % (shield, "_hdv%d" % n_hdv if n_hdv else "", "" if E == 65536 else "_E%d" % E)), "w"), indent=1)
print("shielded wave-sub-steps %d, veto passes per wave-sub-step %.3f" % (int(buf[13]) >> 32, (int(buf[13]) & 0xFFFFFFFF) / max(int(buf[13]) >> 32, 1)))
