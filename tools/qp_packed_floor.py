#!/usr/bin/env python3
"""GPU box: what would the interior-point QPs of ONE policy step of the bench workload cost if they could be packed?

The fused step kernel runs about 8 of 64 lanes per interior-point iteration (an env has about one QP ready at a time:
tools/ipm_sched_model.py).  This measures the other end: the same NUMBER of QPs (65 536 envs x 23.7 per env and step,
profiles/r03/ipm_sched_model.json) solved by the stand-alone batch entry mm_shield_qp -- one QP per lane, every lane
busy, no dependencies between QPs, nothing of the env around them -- on the QPs the reference assembled in the golden
tapes (77 916, 0.46 % of them run to cvxopt's iteration cap), in two orders:

  shuffled   a wave holds whatever comes: one capped QP holds its wave for 100 iterations
  by length  QPs sorted by their iteration count: lanes of a wave finish together (perfect packing)

The second number + the exact-mode step (everything that is not the QP) is the floor of ANY schedule of this mode with
this solver; DESIGN.md section 6 quotes it.  Design tool, not part of the product or the tests.
"""
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from marl_mass_amd import VecMergeEnv  # noqa: E402


def main():
    Gs, hs, rs = [], [], []
    for f in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "*.npz"))):
        z = np.load(f)
        if "qp_rows" in z.files and len(z["qp_rows"]):
            Gs.append(z["qp_G"]); hs.append(np.nan_to_num(z["qp_h"])); rs.append(z["qp_rows"])
    G, h, rows = np.concatenate(Gs), np.concatenate(hs), np.concatenate(rs).astype(np.int32)
    env = VecMergeEnv(1, 2, config={"safety_guarantee": "none"})
    _, st, it = env.shield_qp(G, h, rows, solver="ipm", with_iters=True)
    it = it.cpu().numpy()
    per_step = int(65536 * 23.68)
    rng = np.random.default_rng(5)
    pick = rng.integers(0, len(rows), per_step)
    res = {"qps_recorded": int(len(rows)), "capped_share": float((it == 100).mean()), "iters_mean": float(it.mean()),
           "qps_per_policy_step": per_step, "lane_iterations_per_policy_step": int(it[pick].sum())}
    for name, order in (("shuffled", pick), ("by_length", pick[np.argsort(it[pick], kind="stable")])):
        Gd = torch.as_tensor(G[order], device="cuda:0"); hd = torch.as_tensor(h[order], device="cuda:0")
        rd = torch.as_tensor(rows[order], device="cuda:0")
        for solver in ("ipm", "exact"):
            env.shield_qp(Gd, hd, rd, solver=solver)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                env.shield_qp(Gd, hd, rd, solver=solver)  # (polls the latch: synchronous)
            res["%s_%s_ms" % (name, solver)] = (time.perf_counter() - t0) / 5 * 1e3
    res["reading"] = ("by_length_ipm_ms = the interior-point work of one policy step of the bench workload with every lane busy "
                      "(includes 128 B of operands read per QP and the allocation of the outputs, see *_exact_ms for that overhead)")
    print(json.dumps(res, indent=1))
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "qp_packed_floor.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
