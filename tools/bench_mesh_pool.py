"""Development helper: the 1 M-triangle height-field render against the path pool's size."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracingrenderer_amd as rt
import _rays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sg = rt.HostScene.from_spec(_rays.heightfield_spec(n))
sg.render(784, 784, 1, mode=1)
for pool in (1 << 21, 1 << 22, 1 << 23, 1 << 24):
    for streams in (1, 2):
        sg.render(784, 784, 2, mode=1, streams=streams, pool_size=pool)
        t0 = time.time()
        rgb, st = sg.render(784, 784, spp, mode=1, streams=streams, pool_size=pool, profile=True)
        dt = time.time() - t0
        print("pool 2^%d streams %d: %.1f ms  %.1f Msamples/s  %.1f Mrays/s  iterations %d trace %.1f logic %.1f ms"
              % (int(np.log2(pool)), streams, dt * 1e3, 784 * 784 * spp / dt / 1e6, st.rays / dt / 1e6, st.iterations, st.ms_trace, st.ms_logic), flush=True)
