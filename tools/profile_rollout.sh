#!/bin/bash
# rocprofv3 kernel stats of the rollout counterpart (policy_kernel + step_kernel + bookkeeping) on the GPU box
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_rollout; rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o rollout -- python3 "$R/tools/rollout_bench.py" > "$O/out.txt" 2> "$O/err.txt"
f=$(find "$O" -name "*kernel_stats.csv" | head -1)
[ "$f" = "$O/rollout_kernel_stats.csv" ] || cp "$f" "$O/rollout_kernel_stats.csv"
head -8 "$O/rollout_kernel_stats.csv" | cut -c1-220
