"""Development helper: very large and very lopsided images, and the multi-device call at size."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import raytracingrenderer_amd as rt
sg = rt.default_scene()
for (W, H, spp) in ((6000, 6000, 4), (8192, 4096, 2), (20000, 64, 8), (64, 20000, 8)):
    buf = torch.empty(W * H * 3, dtype=torch.float32, device="cuda:0")
    t0 = time.time()
    st = sg.render_device(W, H, spp, buf.data_ptr(), seed=1)
    torch.cuda.synchronize()
    m = float(buf.mean())
    print(W, H, spp, "chunk", st.chunk_spp, "mode", st.mode, "%.1f ms" % ((time.time() - t0) * 1e3), "mean %.4f" % m, "finite", bool(torch.isfinite(buf).all()), flush=True)
    del buf
# multi on the biggest
rgb, st = sg.render_multi(4096, 4096, 2, [0, 0, 0, 0], seed=1)
one, s1 = sg.render(4096, 4096, 2, seed=1)
print("multi == single:", bool((rgb.view(np.uint32) == one.view(np.uint32)).all()), st.chunk_spp, s1.chunk_spp, "gather ms %.2f" % st.ms_gather)
