"""Development helper: the fused sampling kernel alone on a height-field scene."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
import _rays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sg = rt.HostScene.from_spec(_rays.heightfield_spec(n))
sg.render(784, 784, 1, mode=2)
t0 = time.time(); rgb, st = sg.render(784, 784, spp, mode=2); dt = time.time() - t0
print("%d^2 field, fused: %.1f ms  %.1f Msamples/s  %.1f Mrays/s" % (n, dt * 1e3, 784 * 784 * spp / dt / 1e6, st.rays / dt / 1e6))
