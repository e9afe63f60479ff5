#!/bin/bash
# Runs on the GPU box: the bench.py variants behind DESIGN.md's measurement table -> gpurun_out/variants.jsonl
# (one bench line each; the first is the default headline line incl. cpu_baseline and qp_fidelity_mode).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/variants.jsonl
: > "$OUT"
cd "$R"
python3 bench.py | tail -1 >> "$OUT"
for v in "--shield hss" "--shield none" "--envs 131072 --agents 4" "--envs 32768 --agents 12" "--envs 8192" \
         "--envs 16384" "--hdv 4" "--hdv 4 --shield hss" "--hdv 4 --shield none" "--env-id merge-multi-agent-v0 --shield none" \
         "--traffic-density 1 --agents 6" "--traffic-density 2 --agents 8" "--traffic-density 3 --agents 11" \
         "--traffic-density 1 --agents 6 --mixed-traffic" "--traffic-density 3 --agents 11 --mixed-traffic"; do
  python3 bench.py --no-cpu-baseline --no-fidelity-line $v | tail -1 >> "$OUT"
done
python3 - "$OUT" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    c = d["config"]
    print("%-8s N=%-2d E=%-6d %-40s %.4f ms  %.3e /s" % (c["workload"].split("safety_guarantee=")[1].split(",")[0], c["agents"], c["envs_per_gpu"],
          ("hdv" if "HDV" in c["workload"] else "") + (" v0" if "v0" in c["workload"] else "") + (" drawn" + d["metric"].split("traffic_density=")[1][:8] if "traffic_density=" in d["metric"] else ""), d["ms_per_step"], d["value"]))
    if "qp_fidelity_mode" in d:
        f = d["qp_fidelity_mode"]; print("   ipm fidelity mode: %.3f ms  %.3e /s" % (f["ms_per_step"], f["value"]))
PY
