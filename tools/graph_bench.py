"""GPU box: per-step wall time of env.step launched eagerly vs replayed from a captured hipGraph (16 steps per graph)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from marl_mass_amd import VecMergeEnv, _cabi as abi
for E in (65536, 16384, 8192):
    env = VecMergeEnv(E, 8, config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=1000, auto_reset=True,
                      skip_outputs=("agents_info", "action_mask", "crashed"))
    env.enable_metrics(); env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(123)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device="cuda:0")
    ring = [torch.multinomial(p, E * 8, True, generator=g).view(E, 8).int() for _ in range(16)]
    env.env_i32[abi.EP["STEPS"]] = ((torch.arange(E, device="cuda:0") * 37) % 100).to(torch.int32)
    for t in range(110): env.step(ring[t % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(192): env.step(ring[t % 16])
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 192
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for t in range(16): env.step(ring[t])   # warm-up on the side stream
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for t in range(16): env.step(ring[t])
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(12): gr.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 192
    print("E=%6d  eager %.4f ms/step   graph %.4f ms/step" % (E, eager * 1e3, graph * 1e3))
