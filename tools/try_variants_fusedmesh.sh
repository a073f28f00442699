#!/bin/bash
# Development helper: fused-kernel render of the height-field scenes with alternative builds
# of the HIP library (raytracingrenderer_amd/librtr_hip_<name>.so), restoring the default.
cd $GRAFT_REPO_ROOT
cp raytracingrenderer_amd/librtr_hip.so /tmp/librtr_hip_keep.so
for n in "$@"; do
  cp raytracingrenderer_amd/librtr_hip_$n.so raytracingrenderer_amd/librtr_hip.so
  echo "== $n"
  timeout -k 10 120 python -u tools/bench_mesh_render.py 40 64 2>&1 | tail -1
  timeout -k 10 120 python -u tools/bench_mesh_render.py 708 64 2>&1 | tail -1
done
cp /tmp/librtr_hip_keep.so raytracingrenderer_amd/librtr_hip.so
