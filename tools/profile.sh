#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats + separate PMC passes for bench.py's default
# workload, then tools/profile_summary.py condenses them into gpurun_out/prof/summary.json.
# Usage: bash tools/profile.sh [extra bench.py args]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof${PROF_TAG:+_$PROF_TAG}   # PROF_TAG=name: one output directory per profiled workload
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-fidelity-line --steps ${PROF_STEPS:-200} --warmup 10 "$@" > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d "$O/pmc$i" -o pmc -- python3 "$R/bench.py" --no-cpu-baseline --no-fidelity-line --steps 20 --warmup 5 "$@" > /dev/null 2> "$O/pmc$i.err"
  echo "pmc pass $i ($c) done"
done
# counter calibration on known byte counts (8 B/lane doubles, 1 B/lane bytes, float2 runs)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$O/calib_$c" -o pmc -- "$R/tools/ubench/fetch_calib" > "$O/calib_$c.txt" 2> "$O/calib_$c.err" || true
done
python3 "$R/tools/profile_summary.py" "$O" "$@"
