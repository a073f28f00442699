"""Development helper: rtr_scene_create alone (arrays already flattened on the host) for the
1 M-triangle height-field case, a few repetitions."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
from raytracingrenderer_amd import scenes
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
rt.default_scene().device_scene()
sg = rt.HostScene.from_spec(scenes.heightfield_spec(n))
sg.flatten_for_upload()
for rep in range(4):
    sg._invalidate_device()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sg.device_scene()
    torch.cuda.synchronize()
    print("%d triangles: rtr_scene_create %.1f ms" % (2 * n * n, (time.perf_counter() - t0) * 1e3), flush=True)
