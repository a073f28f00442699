"""Development helper: render the 1 M-triangle height-field scene with both sampling loops."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracingrenderer_amd as rt
import _rays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sg = rt.HostScene.from_spec(_rays.heightfield_spec(n))
ref = None
for how in (dict(mode=1, streams=1), dict(mode=1, streams=2), dict(mode=2)):
    sg.render(784, 784, 1, **how)
    t0 = time.time()
    rgb, st = sg.render(784, 784, spp, profile=True, **how)
    dt = time.time() - t0
    same = "" if ref is None else (" same image: %s" % bool((rgb.view(np.uint32) == ref.view(np.uint32)).all()))
    ref = rgb if ref is None else ref
    print(how, "%.1f ms  %.1f Msamples/s  %.1f Mrays/s  trace %.1f logic %.1f fused %.1f other %.1f ms%s"
          % (dt * 1e3, 784 * 784 * spp / dt / 1e6, st.rays / dt / 1e6, st.ms_trace, st.ms_logic, st.ms_fused, st.ms_other, same), flush=True)
