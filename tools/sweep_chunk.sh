#!/bin/bash
# Development helper: short bench for several values of --chunk.
for c in "$@"; do
  timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk $c > gpurun_out/sw_$c.log 2> gpurun_out/sw_$c.err || { tail -5 gpurun_out/sw_$c.err; exit 1; }
  python - "$c" gpurun_out/sw_$c.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("chunk", sys.argv[1], "Msamples/s %.1f ms/step %.1f" % (d["value"], d["ms_per_step"]))
PY
done
