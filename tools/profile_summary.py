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
# the kernels of one mm_step: one fused step_kernel launch, or -- interior-point mode of a CAV-only batch -- nsub + 1 launches of
# the phase form of step_kernel with a sweep_kernel launch after each act half (7 launches at 3 sub-steps per policy step)
is_step = lambda name: "step_kernel" in name or "sweep_kernel" in name
srows = [r for r in rows if is_step(r["Name"])]
step = srows[0]
split = any("sweep_kernel" in r["Name"] for r in srows)
n_sweep = sum(int(r["Calls"]) for r in srows if "sweep_kernel" in r["Name"])
n_phase = sum(int(r["Calls"]) for r in srows if "step_kernel" in r["Name"])
per_step = (n_phase + n_sweep) // (n_phase - n_sweep) if split else 1  # launches per mm_step
out["kernel"], out["calls"], out["avg_ns_all_calls"], out["pct_of_gpu_time"] = step["Name"], int(step["Calls"]), float(step["AverageNs"]), float(step["Percentage"])
if split:
    out["kernel"] = " + ".join(r["Name"].split("(")[0].replace("void ", "") for r in srows)
    out["launches_per_mm_step"] = per_step
    out["kernels"] = [{"name": r["Name"].split("(")[0].replace("void ", ""), "calls": int(r["Calls"]), "avg_ns_all_calls": float(r["AverageNs"])} for r in srows]
    out["pct_of_gpu_time"] = sum(float(r["Percentage"]) for r in srows)
# bench.py rolls 100 untimed steps + 10 warm-up steps before the 200 timed ones: the stats average above covers all 310
# launches (the first ones run on a not-yet-stationary batch).  The kernel trace gives the timed region on its own.
trace = glob.glob(os.path.join(O, "stats", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    tr = [r for r in csv.DictReader(open(trace[0])) if is_step(r["Kernel_Name"])]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the timed region = the launches of the last --steps mm_steps that precede bench.py's kernel-time loop (a hipGraph replay
    # of min(steps, 64) launches, itself preceded by one warm replay: 2 x kn mm_steps after the timed region)
    steps = int(os.environ.get("PROF_STEPS", "200"))
    try:  # (the bench line says how many mm_steps its kernel-time loop ran behind the timed region)
        bl = json.loads(open(os.path.join(O, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        tail = int(bl["roofline"]["mm_steps_after_timed_region"]) * per_step
    except Exception:  # noqa
        tail = 0
    last = tr[len(tr) - tail - steps * per_step: len(tr) - tail]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
    out["avg_ns"] = sum(d) / (len(d) / per_step)  # per mm_step (= per launch in the fused form)
    out["timed_region"] = {"launches": len(d), "mm_steps": len(d) // per_step, "avg_ns": out["avg_ns"], "min_ns": min(d), "max_ns": max(d)}
    if split:
        by = {}
        for r in last:
            by.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        out["timed_region"]["per_kernel_avg_ns"] = {k: sum(v) / len(v) for k, v in by.items()}
else:
    out["avg_ns"] = out["avg_ns_all_calls"]
pmc = {}
for f in glob.glob(os.path.join(O, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if not is_step(r["Kernel_Name"]):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc[k] = sum(v) / (len(v) / per_step)  # per mm_step: the sum over its launches (fused form: per launch)
out["pmc_per_launch"] = pmc
# calibration: counters against known byte counts for this path's access widths (tools/ubench/fetch_calib.hip)
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(O, "calib_" + c, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            calib.setdefault(k, {})[c + "_KiB"] = sum(v) / len(v)
n = 64 << 20
truth = {"copy_f64": (n * 8, n * 8), "copy_u8": (n, n), "copy_f32x30": (n * 8, n * 8)}
for k, (rd, wr) in truth.items():
    if k in calib:
        calib[k]["bytes_read"], calib[k]["bytes_written"] = rd, wr
        if "FETCH_SIZE_KiB" in calib[k]:
            calib[k]["true_over_reported_fetch"] = rd / (calib[k]["FETCH_SIZE_KiB"] * 1024)
        if "WRITE_SIZE_KiB" in calib[k]:
            calib[k]["true_over_reported_write"] = wr / (calib[k]["WRITE_SIZE_KiB"] * 1024)
out["counter_calibration"] = calib
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # counters are in KiB.  MI355X guide: on gfx950 FETCH_SIZE reports half the bytes of a coalesced stream -> x2; the
    # calibration above checks that factor on 8 B/lane and 1 B/lane accesses (this kernel's widths)
    out["hbm_traffic_bytes_per_launch_uncorrected"] = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
    out["hbm_traffic_bytes_per_launch"] = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
try:
    out["bench_line_under_rocprof"] = json.loads(open(os.path.join(O, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa
    out["bench_line_under_rocprof"] = str(e)
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
import shutil
shutil.copy(stats[0], os.path.join(O, "bench_kernel_stats.csv"))
print(json.dumps({k: out[k] for k in ("kernel", "calls", "avg_ns", "pmc_per_launch") if k in out}, indent=1))
