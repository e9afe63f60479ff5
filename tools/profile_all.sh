#!/bin/bash
# Runs on the GPU box: tools/profile.sh (rocprofv3 kernel stats + separate PMC passes) for the workloads DESIGN.md makes
# limiter claims about -> gpurun_out/prof_<tag>/summary.json each; copy the summaries to profiles/rNN/<tag>_summary.json.
#   bash tools/profile_all.sh [tag ...]      (default: all)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
declare -A ARGS=(
  [headline]=""
  [ipm]="--qp-solver ipm"
  [hss_ipm]="--qp-solver ipm --shield hss"
  [mixed44]="--hdv 4"
  [mixed44_ipm]="--hdv 4 --qp-solver ipm"
  [density3_ipm]="--traffic-density 3 --agents 11 --qp-solver ipm"
  [g16]="--envs 32768 --agents 12 --pow2-groups"
  [lanes12]="--envs 32768 --agents 12"
  [density3]="--traffic-density 3 --agents 11"
  [density1]="--traffic-density 1 --agents 6"
  [density3mixed]="--traffic-density 3 --agents 11 --mixed-traffic"
  [small8192]="--envs 8192"
)
TAGS=("$@"); [ ${#TAGS[@]} -gt 0 ] || TAGS=(headline ipm hss_ipm mixed44 mixed44_ipm lanes12 density1 density3 density3_ipm density3mixed g16 small8192)
for t in "${TAGS[@]}"; do
  steps=200; case "$t" in *ipm) steps=30;; esac
  echo "== profile $t: bench.py ${ARGS[$t]}"
  PROF_TAG=$t PROF_STEPS=$steps bash "$R/tools/profile.sh" ${ARGS[$t]} | tail -3
done
