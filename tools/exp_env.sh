#!/bin/bash
# Development helper: short bench under several values of one environment variable.
# usage: tools/exp_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp_$v.log 2> gpurun_out/exp_$v.err || { tail -5 gpurun_out/exp_$v.err; exit 1; }
  python - "$var=$v" gpurun_out/exp_$v.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "Msamples/s %.1f ms/step %.1f iters %d kernels %s" % (d["value"], d["ms_per_step"], d["iterations_per_step_rank0"], {k: round(v,1) for k,v in d["kernel_ms_per_step_rank0"].items()}))
PY
done
