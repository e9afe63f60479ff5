#!/bin/bash
# GPU box: the default bench line (no CPU baseline, no second mode) for every tuning build marl-mass_amd/csrc/tune/libmm_s_*.so
# (single-instantiation builds of the headline kernel with different compiler options; loaded through MM_HIP_LIB).
# -> gpurun_out/lib_sweep.txt : name, ms per step, kernel ms (HIP events)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
: > gpurun_out/lib_sweep.txt
for rep in 1 2; do
for so in marl-mass_amd/csrc/tune/libmm_s_*.so; do
  n=$(basename "$so" .so)
  MM_HIP_LIB=$R/$so timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-fidelity-line --steps 400 --warmup 20 "$@" 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s %.4f ms/step  kernel %.4f ms' % ('$n', d['ms_per_step'], d['roofline']['kernel_ms']))" >> gpurun_out/lib_sweep.txt || echo "$n failed" >> gpurun_out/lib_sweep.txt
done
done
sort gpurun_out/lib_sweep.txt
