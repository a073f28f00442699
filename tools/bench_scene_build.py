"""Development helper: where the time goes between mesh arrays and a scene resident on
the GPU for the 1 M-triangle height-field case."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
import _rays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
spec = _rays.heightfield_spec(n)
rt.default_scene().device_scene()          # HIP start-up is not part of this
for rep in range(2):
    t0 = time.time(); sg = rt.HostScene.from_spec(spec); t1 = time.time()
    sg.flatten(); t2 = time.time()
    sg.device_scene(); t3 = time.time()
    print("%d triangles: meshes+trees %.3f s, flatten %.3f s, flatten again + rtr_scene_create %.3f s"
          % (2 * n * n, t1 - t0, t2 - t1, t3 - t2), flush=True)
