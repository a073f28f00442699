"""rtr_build_bvh on the boxes of a height-field mesh, twice (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracingrenderer_amd as rt  # noqa: E402
from raytracingrenderer_amd.scenes import heightfield_spec  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 708
_, _, verts, idx = heightfield_spec(cells)["objects"][-1]
tri = verts[idx.reshape(-1, 3)]
lo, hi = tri.min(1), tri.max(1)
for rep in range(3):
    t0 = time.perf_counter()
    rt.build_bvh(lo, hi)
    print("rtr_build_bvh %d boxes: %.2f ms" % (len(lo), (time.perf_counter() - t0) * 1e3), flush=True)
