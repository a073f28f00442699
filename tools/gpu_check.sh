#!/bin/bash
# GPU check used during development: parity tests, then a short bench (no CPU baseline).
set -o pipefail
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/gpu_tests.log | head -20; exit $rc; }
timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dev.log 2> gpurun_out/bench_dev.err || { tail -20 gpurun_out/bench_dev.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_dev.log").read().strip().splitlines()[-1])
print("Msamples/s %.1f  ms/step %.1f iters %d kernels %s  trace avg ms %.4f  frac %.4f" % (d["value"], d["ms_per_step"], d["iterations_per_step_rank0"], {k: round(v,1) for k,v in d["kernel_ms_per_step_rank0"].items()}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
