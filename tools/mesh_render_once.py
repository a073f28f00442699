"""Development helper: one render of the 1 M-triangle height-field scene (after a one-sample
warm-up), for profiling.  usage: mesh_render_once.py [spp] [mode] [streams] [pool]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
from raytracingrenderer_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pool = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sg = rt.HostScene.from_spec(scenes.heightfield_spec(708))
sg.render(784, 784, 1, mode=mode, streams=streams, pool_size=pool)
t0 = time.time()
rgb, st = sg.render(784, 784, spp, mode=mode, streams=streams, pool_size=pool)
dt = time.time() - t0
print("mode %d streams %d pool %d: %.1f ms  %.1f Msamples/s  %.1f Mrays/s  iterations %d"
      % (st.mode, streams, st.pool_size, dt * 1e3, 784 * 784 * spp / dt / 1e6, st.rays / dt / 1e6, st.iterations))
