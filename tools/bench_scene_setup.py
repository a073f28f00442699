"""Scene set-up times of the large-scene cases, phase by phase, with the tree built on the host
(std::sort) and on the device (rtr_build_bvh): HostScene.from_spec (triangles + trees), flatten,
rtr_scene_create."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracingrenderer_amd as rt  # noqa: E402
from raytracingrenderer_amd.scenes import heightfield_spec  # noqa: E402

for cells in [int(a) for a in sys.argv[1:]] or [708, 1500]:
    spec = heightfield_spec(cells)
    for how in ("host", "device", "device"):
        os.environ["RTR_TREE_BUILDER"] = how
        t0 = time.perf_counter()
        s = rt.HostScene.from_spec(spec)
        t1 = time.perf_counter()
        s.flatten_for_upload()
        t2 = time.perf_counter()
        s.device_scene()
        t3 = time.perf_counter()
        print(json.dumps({"triangles": 2 * cells * cells, "builder": how, "trees_ms": (t1 - t0) * 1e3,
                          "flatten_ms": (t2 - t1) * 1e3, "rtr_scene_create_ms": (t3 - t2) * 1e3}), flush=True)
        s.close()
