"""A larger bit-for-bit comparison of GPU renders with the CPU checker than the test suite affords:
the default scene (mirrored pairs, the bounded flavour of the fused kernel) and the specular /
microfacet variant, a few hundred million rays each, all sampling loops.  The checker runs on the
host's cores.  usage: big_bitexact.py [size] [spp] [default|variant]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _rays  # noqa: E402
from _oracle import Oracle, default_scene_spec  # noqa: E402
import raytracingrenderer_amd as rt  # noqa: E402
from raytracingrenderer_amd.scenes import variant_scene_spec  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
only = sys.argv[3] if len(sys.argv) > 3 else ""
O = Oracle()


def heartbeat(stop):
    """a line a minute while the checker renders (a silent job is taken to be hung)"""
    while not stop.wait(60):
        print("  ... checker still rendering", flush=True)


for name, spec, kw in (("default", default_scene_spec(), {}),
                       ("variant", variant_scene_spec(), dict(max_depth=16, rr_prob=0.95))):
    if only and only != name:
        continue
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    t0 = time.time()
    stop = threading.Event()
    threading.Thread(target=heartbeat, args=(stop,), daemon=True).start()
    img, stats = O.render(so, size, size, spp, seed=77, chunk_spp=8, threads=16, **kw)
    stop.set()
    print("%s: checker %dx%dx%d in %.1f s, %d rays" % (name, size, size, spp, time.time() - t0, stats[1]), flush=True)
    for how in (dict(), dict(mode=1)):
        rgb, st = sg.render(size, size, spp, seed=77, chunk_spp=8, **kw, **how)
        same = _rays.same_bits(rgb, img)
        assert same.all(), "%s %s: %d pixels differ" % (name, how, (~same.reshape(-1, 3).all(1)).sum())
        assert st.rays == stats[1] and st.vertices == stats[3]
    print(name, "bit-exact", flush=True)
print("BIG BITEXACT OK")
