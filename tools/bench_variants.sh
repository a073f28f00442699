#!/bin/bash
# Times the headline render with variant builds of the library (build/variants/librtr_hip_<name>.so,
# experiment switches that price parts of the fused kernel; their images are wrong on purpose).
# usage (on the GPU box): tools/bench_variants.sh name...
for v in "$@"; do
  so=$GRAFT_REPO_ROOT/build/variants/librtr_hip_$v.so
  [ "$v" = base ] && so=$GRAFT_REPO_ROOT/raytracingrenderer_amd/librtr_hip.so
  RTR_HIP_SO=$so timeout -k 10 120 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms')" || exit 1
done
