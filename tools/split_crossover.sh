#!/bin/bash
# GPU box: interior-point step, fused kernel (debug_flags bit 2) vs split step (bit 3) by batch size -> gpurun_out/split_crossover.txt
# (where mm_step's own rule `steps_split` should put the boundary).   bash tools/split_crossover.sh [bench.py args, e.g. --shield hss]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out; : > gpurun_out/split_crossover.txt
for E in ${CROSS_ENVS:-8192 12288 16384 24576 32768}; do
  for f in 4 8; do
    MM_DEBUG_FLAGS=$f timeout -k 10 200 python3 bench.py --qp-solver ipm --envs $E --steps 20 --warmup 3 --no-cpu-baseline --no-fidelity-line "$@" 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%6d envs  %s  %.4f ms/step' % ($E, 'fused' if $f == 4 else 'split', d['ms_per_step']))" >> gpurun_out/split_crossover.txt || echo "$E $f failed" >> gpurun_out/split_crossover.txt
  done
done
cat gpurun_out/split_crossover.txt
