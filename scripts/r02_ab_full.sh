#!/bin/bash
# A/B of runtime knobs on the full default bench (200 steps), variants interleaved on one box:
#   bash scripts/r02_ab_full.sh 2 "A=1" "BFF_MERGE_LIVE=0"
export TMPDIR=/tmp
reps=$1; shift
for r in $(seq $reps); do
for v in "$@"; do
  env $v python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'scenes/s', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4), d['kernels_ms'], 'alone', round(d['roofline_merge']['alone_on_chip_ms'], 4), round(d['roofline']['alone_on_chip']['avg_launch_ms'], 4), d['host_ms'])
"
done; done
