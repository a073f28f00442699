#!/bin/bash
# Development helper: compare fused and wavefront images for alternative builds of the library.
cd $GRAFT_REPO_ROOT
cp raytracingrenderer_amd/librtr_hip.so /tmp/librtr_hip_keep.so
cat > /tmp/cmp.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import raytracingrenderer_amd as rt
sg = rt.default_scene()
a, sa = sg.render(64, 48, 8, seed=1, mode=2, chunk_spp=2)
b, sb = sg.render(64, 48, 8, seed=1, mode=1, chunk_spp=2)
print(sys.argv[1], "differing pixels", int((a.view(np.uint32) != b.view(np.uint32)).any(1).sum()), "rays", sa.rays, sb.rays)
PY
for n in "$@"; do
  cp raytracingrenderer_amd/librtr_hip_$n.so raytracingrenderer_amd/librtr_hip.so
  timeout -k 10 100 python -u /tmp/cmp.py $n 2>&1 | tail -1
done
cp /tmp/librtr_hip_keep.so raytracingrenderer_amd/librtr_hip.so
