"""Development helper: the two sampling loops against each other on several scenes at sizes
the CPU checker cannot reach (bit-exact images and equal counters expected)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import raytracingrenderer_amd as rt
from raytracingrenderer_amd.api import default_scene_spec
import _rays
f32 = np.float32

def variant():
    spec = default_scene_spec(); objs = []
    for ob in spec["objects"]:
        if ob[0] == "sphere" and ob[1] in ("red", "blue"): ob = ("sphere", "spec", ob[2], ob[3])
        if ob[0] == "sphere" and ob[1] == "green": ob = ("sphere", "micro", ob[2], ob[3])
        objs.append(ob)
    spec["objects"] = objs
    return spec

def lights():
    spec = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    spec["objects"] = [("sphere", "light", (100, 300, -200), 15)] + spec["objects"] + \
                      [("mesh", "light", plate, [0, 1, 2, 3, 2, 1])]
    return spec

cases = [("default", default_scene_spec(), dict()), ("variant d16 rr.95", variant(), dict(max_depth=16, rr_prob=0.95)),
         ("three lights", lights(), dict()), ("heightfield 64", _rays.heightfield_spec(64), dict()),
         ("heightfield 300", _rays.heightfield_spec(300), dict()),
         ("heightfield 300 d16 rr.95", _rays.heightfield_spec(300), dict(max_depth=16, rr_prob=0.95))]
ok = True
for name, spec, kw in cases:
    sg = rt.HostScene.from_spec(spec)
    for seed in (1, 2):
        a, sa = sg.render(320, 240, 512, seed=seed, mode=2, **kw)
        sg.render(320, 240, 2, seed=99, mode=2, **kw)
        b, sb = sg.render(320, 240, 512, seed=seed, mode=1, **kw)
        same = bool((a.view(np.uint32) == b.view(np.uint32)).all()) or bool(np.array_equal(np.nan_to_num(a), np.nan_to_num(b)))
        cnt = (sa.rays, sa.vertices) == (sb.rays, sb.vertices)
        ok &= same and cnt
        print(name, "seed", seed, "image", same, "counters", cnt, "rays %.3g" % sa.rays, "nan px", int(np.isnan(a).any(1).sum()), flush=True)
print("ALL OK" if ok else "MISMATCH")
