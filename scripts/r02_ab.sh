#!/bin/bash
# A/B of a runtime knob on the sequential config-2 step: kernel timeline of one default scene + one "many" scene.
export TMPDIR=/tmp
for v in "$@"; do
  name=$(echo "$v" | tr '= ' '__')
  env $v python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'scenes/s', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3), d['kernels_ms'], d['host_ms'])
"
done
