#!/bin/bash
# GPU box: bench lines of the workloads that run in 6- / 12-lane env groups (N = 5..6, 9..12), with --pow2-groups for A-B.
set -o pipefail
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --no-fidelity-line "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %.4f ms  %.3e /s' % ('$*', d['ms_per_step'], d['value']))"; }
run --traffic-density 1 --agents 6
run --traffic-density 3 --agents 11
run --envs 32768 --agents 12
run --envs 65536 --agents 6
run --envs 65536 --agents 6 --shield hss
run --envs 65536 --agents 6 --shield none
run --envs 32768 --agents 12 --shield none
run --envs 32768 --agents 10
run --traffic-density 1 --agents 6 --mixed-traffic
run --traffic-density 3 --agents 11 --mixed-traffic
run --envs 65536 --agents 6 --hdv 3
run --traffic-density 1 --agents 6 --qp-solver ipm --steps 30 --warmup 3
run --traffic-density 1 --agents 6 --qp-solver ipm --steps 30 --warmup 3 --pow2-groups
run --traffic-density 3 --agents 11 --qp-solver ipm --steps 20 --warmup 3
run --traffic-density 3 --agents 11 --qp-solver ipm --steps 20 --warmup 3 --pow2-groups
