#!/bin/bash
# Usage (GPU box): bash tools/pmc_once.sh "CTR1 CTR2 ..." [bench args]  -> mean per step_kernel launch
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C="$1"; shift
O=$R/gpurun_out/pmc_once; rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $C --output-format csv -d "$O" -o pmc -- python3 "$R/bench.py" --no-cpu-baseline --no-fidelity-line --steps 20 --warmup 5 "$@" > /dev/null 2> "$O/err.txt" || { tail -5 "$O/err.txt"; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, os
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s %16.1f" % (k, sum(v) / len(v)))
PY
