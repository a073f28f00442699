#!/bin/bash
# Development helper: bench with extra arguments, one summary line per run.
# usage: tools/sweep.sh "--pool 600000" "--pool 1800000" ...
for A in "$@"; do
  python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$A', round(d['value']), d['iterations_per_step_rank0'], {k: round(v,1) for k,v in d['kernel_ms_per_step_rank0'].items()})"
done
