#!/usr/bin/env python3
"""Copies the summaries tools/profile_all.sh left under gpurun_out/prof_<tag>/ into profiles/<round>/ and rebuilds
profiles/traffic.json (one row per profiled workload: what bench.py reports as roofline.traffic for that workload).

    python tools/collect_profiles.py r03
"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
dst = os.path.join(REPO, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
rows = []
for tag in sorted(os.listdir(os.path.join(REPO, "gpurun_out"))):
    if not tag.startswith("prof_"):
        continue
    src = os.path.join(REPO, "gpurun_out", tag)
    sj = os.path.join(src, "summary.json")
    if not os.path.exists(sj):
        continue
    name = tag[5:]
    shutil.copy(sj, os.path.join(dst, name + "_summary.json"))
    if os.path.exists(os.path.join(src, "bench_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))
    d = json.load(open(sj))
    b = d.get("bench_line_under_rocprof")
    pmc = d.get("pmc_per_launch", {})
    if not isinstance(b, dict) or "FETCH_SIZE" not in pmc:
        continue
    c = b["config"]
    shield = {"cbf-cav": "mass", "cbf-avs_cint": "hss", "none": "none"}[c["workload"].split("safety_guarantee=")[1].split(",")[0]]
    rows.append({
        "workload": {"envs_per_gpu": c["envs_per_gpu"], "agents": c["agents"], "shield": shield,
                     "env_id": "merge-multi-agent-v1" if "v1" in c["workload"] else "merge-multi-agent-v0",
                     "qp_solver": c["qp_solver"], "hdv": int(c["workload"].split("of which ")[1].split(" HDVs")[0]) if "of which" in c["workload"] else 0,
                     "traffic_density": c.get("traffic_density", 0), "pow2_groups": bool(c.get("pow2_groups", False)),
                     "mixed_traffic": " mixed" in b["metric"]},
        "tag": name, "kernel": d["kernel"].split("(")[0].replace("void ", ""),
        "kernel_avg_ns": d["avg_ns"], "bytes_per_launch": d["hbm_traffic_bytes_per_launch"],
        "fetch_bytes": 2 * pmc["FETCH_SIZE"] * 1024, "write_bytes": pmc["WRITE_SIZE"] * 1024,
        "alg_bytes_per_launch": b["roofline"]["alg_bytes_per_launch"],
        "traffic_over_algorithmic": d["hbm_traffic_bytes_per_launch"] / b["roofline"]["alg_bytes_per_launch"],
        "valu_per_wave": pmc.get("SQ_INSTS_VALU", 0) / max(pmc.get("SQ_WAVES", 1), 1),
        "salu_per_wave": pmc.get("SQ_INSTS_SALU", 0) / max(pmc.get("SQ_WAVES", 1), 1),
        "summary": "%s/%s_summary.json" % (rnd, name)})
out = {"format": "one row per profiled bench.py workload; bench.py matches (envs_per_gpu, agents, shield, env_id, qp_solver, hdv, traffic_density, pow2_groups, mixed_traffic) and "
                 "reports bytes_per_launch as roofline.traffic",
       "source": "tools/profile_all.sh -> tools/profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the bench.py command "
                 "(KiB units); bytes = 2 x FETCH_SIZE + WRITE_SIZE: the x 2 is the guide's gfx950 correction, confirmed on this path's access "
                 "widths by tools/ubench/fetch_calib.hip in the same runs (counter_calibration in each summary)",
       "workloads": rows}
json.dump(out, open(os.path.join(REPO, "profiles", "traffic.json"), "w"), indent=1)
for r in rows:
    print("%-10s %-50s %9.1f us  traffic %.2fx alg  VALU/wave %8.0f  SALU/wave %7.0f" % (
        r["tag"], r["kernel"], r["kernel_avg_ns"] / 1e3, r["traffic_over_algorithmic"], r["valu_per_wave"], r["salu_per_wave"]))
