#!/bin/bash
# Development helper: short bench with alternative builds of the HIP library
# (raytracingrenderer_amd/librtr_hip_<name>.so), restoring the default one afterwards.
cd $GRAFT_REPO_ROOT
cp raytracingrenderer_amd/librtr_hip.so /tmp/librtr_hip_keep.so
for n in "$@"; do
  cp raytracingrenderer_amd/librtr_hip_$n.so raytracingrenderer_amd/librtr_hip.so
  timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/var_$n.log 2> gpurun_out/var_$n.err || { tail -3 gpurun_out/var_$n.err; }
  python - "$n" gpurun_out/var_$n.log <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(sys.argv[1], "Msamples/s %.1f ms/step %.1f vertices/sample %.3f rays/sample %.3f ns/vertex-round %.3f" % (d["value"], d["ms_per_step"], d["vertices_per_sample"], d["rays_per_sample"], d["ms_per_step"] * 1e6 / (d["vertices_per_sample"] * d["config"]["width"] * d["config"]["height"] * d["config"]["spp"])))
except Exception as e: print(sys.argv[1], "failed", e)
PY
done
cp /tmp/librtr_hip_keep.so raytracingrenderer_amd/librtr_hip.so
