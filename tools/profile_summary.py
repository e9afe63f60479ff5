"""Condense the rocprofv3 output of tools/profile.sh into one JSON (kernel stats + per-launch PMC means)."""
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
out = {"command": "tools/profile.sh " + " ".join(sys.argv[2:]) + " (rocprofv3 --kernel-trace --stats; counters in separate --pmc passes)"}
stats = glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(stats[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
out["kernel_stats_top"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:6]]
step = [r for r in rows if "step_kernel" in r["Name"]][0]
out["kernel"], out["calls"], out["avg_ns"], out["pct_of_gpu_time"] = step["Name"], int(step["Calls"]), float(step["AverageNs"]), float(step["Percentage"])
pmc = {}
for f in glob.glob(os.path.join(O, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "step_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)
out["pmc_per_launch"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # both counters are in KiB; this kernel's accesses are 8 B / 4 B / 1 B per lane, not the 16 B/lane
    # stream for which the guide's x2 FETCH correction was measured -> reported both ways
    out["hbm_traffic_bytes_per_launch"] = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
    out["hbm_traffic_bytes_per_launch_fetch_x2"] = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
try:
    out["bench_line_under_rocprof"] = json.loads(open(os.path.join(O, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa
    out["bench_line_under_rocprof"] = str(e)
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
import shutil
shutil.copy(stats[0], os.path.join(O, "bench_kernel_stats.csv"))
print(json.dumps({k: out[k] for k in ("kernel", "calls", "avg_ns", "pmc_per_launch") if k in out}, indent=1))
