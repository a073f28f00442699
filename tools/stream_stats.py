"""Development helper: counters of the streaming traversal (library built with
-DRTR_STREAM_STATS as raytracingrenderer_amd/librtr_hip_stats.so, swapped in by the caller)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import raytracingrenderer_amd as rt
import _rays
lib = rt.hip_lib()
def stats(label):
    out = (C.c_ulonglong * 8)()
    assert lib.rtr_debug_stream_stats(out) == 0
    it, vis, ev, rays, rounds, pairs, splits = [int(x) for x in out[:7]]
    if rays:
        print("%s: rays %d  visits/ray %.1f  lanes alive per iteration %.1f  iterations/refill %.1f  rays/refill %.1f  pairs/ray %.2f  pairs/round %.1f  splits/ray %.3f"
              % (label, rays, vis / rays, vis / max(it, 1), it / max(ev, 1), rays / max(ev, 1), pairs / rays, pairs / max(rounds, 1), splits / rays), flush=True)
sg = rt.HostScene.from_spec(_rays.heightfield_spec(708))
nr = 1 << 22
org, d = _rays.interior_rays(nr, seed=5)
prim, t, pos, nrm = sg.intersect(org, d)
stats("interior rays")
sg.render(784, 784, 16, mode=1, streams=1)
stats("render 16 spp")
