"""Time of the tree build (reference BVH::selfBuildBVH) for the large-scene meshes: the host
mirror's std::sort builder against rtr_build_bvh on the device, and the raw call on boxes."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracingrenderer_amd as rt  # noqa: E402
from raytracingrenderer_amd.scenes import heightfield_spec  # noqa: E402

for cells in [int(a) for a in sys.argv[1:]] or [708, 1500]:
    spec = heightfield_spec(cells)
    _, _, verts, idx = spec["objects"][-1]
    tri = verts[idx.reshape(-1, 3)]
    lo, hi = tri.min(1), tri.max(1)
    rec = {"triangles": len(lo)}
    for rep in range(3):
        t0 = time.perf_counter()
        out = rt.build_bvh(lo, hi)
        rec["rtr_build_bvh_ms"] = (time.perf_counter() - t0) * 1e3
    for how in ("host", "device"):
        os.environ["RTR_TREE_BUILDER"] = how
        best = 1e9
        for rep in range(2):
            s = rt.HostScene(spec["camera"])
            for name, kind, e, kd, ks, ior, rough in spec["materials"]:
                s.add_material(name, kind, e, kd, ks, ior, rough)
            t0 = time.perf_counter()
            s.add_mesh("white", verts, idx)
            best = min(best, (time.perf_counter() - t0) * 1e3)
            s.close()
        rec["add_mesh_%s_ms" % how] = best
    print(json.dumps(rec), flush=True)
