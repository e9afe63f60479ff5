#!/bin/bash
# 1/2/4/8-GPU scaling of the headline bench on ONE node (run it on an 8-GPU box; the driver does the same at round end).
# The launcher is started before anything touches a GPU; one rank per GPU, RCCL only for the end-of-rollout metric all-reduce.
#   bash tools/scale.sh [strong|weak] [extra bench.py args]      -> gpurun_out/scale_<mode>.jsonl
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=${1:-strong}; shift || true
OUT=$R/gpurun_out/scale_$MODE.jsonl
mkdir -p "$R/gpurun_out"; : > "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python3 -c 'import torch; print(torch.cuda.device_count())')
for n in 1 2 4 8; do
  [ "$n" -le "$NG" ] || { echo "skip N=$n (only $NG GPUs)"; continue; }
  if [ "$n" -eq 1 ]; then
    python3 "$R/bench.py" --gpus 1 --steps 200 --warmup 10 --scaling "$MODE" --no-cpu-baseline "$@" | tail -1 >> "$OUT"
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      "$R/bench.py" --gpus "$n" --steps 200 --warmup 10 --scaling "$MODE" --no-cpu-baseline "$@" | tail -1 >> "$OUT"
  fi
  tail -1 "$OUT" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("N=%d  %.3e agent-steps/s  %.3f ms/step  (%s)" % (d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"]))'
done
