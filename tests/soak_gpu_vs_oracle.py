"""Not collected by pytest: a longer GPU-vs-checker comparison than the test suite affords.
Run by hand on a GPU box: python tests/soak_gpu_vs_oracle.py [rounds].
Scenes x seeds x loop variants, images and counters bit for bit; closest hits on fresh
ray batches; photon maps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _rays  # noqa: E402
from _oracle import Oracle, default_scene_spec  # noqa: E402
import raytracingrenderer_amd as rt  # noqa: E402
from test_gpu_parity import _random_scene_spec, _assert_hits_equal, _assert_images_equal  # noqa: E402
from test_golden import _multi_light_spec  # noqa: E402

f32 = np.float32


def variant():
    spec = default_scene_spec()
    objs = []
    for ob in spec["objects"]:
        if ob[0] == "sphere" and ob[1] in ("red", "blue"):
            ob = ("sphere", "spec", ob[2], ob[3])
        if ob[0] == "sphere" and ob[1] == "green":
            ob = ("sphere", "micro", ob[2], ob[3])
        objs.append(ob)
    spec["objects"] = objs
    return spec


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    O = Oracle()
    cases = [("default", default_scene_spec(), {}), ("variant", variant(), dict(max_depth=12, rr_prob=0.9)),
             ("three lights", _multi_light_spec(), {}), ("random 3000", _random_scene_spec(21), {}),
             ("heightfield 90", _rays.heightfield_spec(90), {}), ("heightfield 380", _rays.heightfield_spec(380), {})]
    for name, spec, kw in cases:
        so = O.build_scene(spec)
        O.set_math(so, portable=True)
        sg = rt.HostScene.from_spec(spec)
        for rnd in range(rounds):
            seed = 100 + rnd
            org, d = _rays.interior_rays(200000, seed=seed)
            _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))
            W, H, spp = 72 + 8 * rnd, 56, 24
            img, stats = O.render(so, W, H, spp, seed=seed, chunk_spp=5, **kw)
            for how in (dict(), dict(mode=1, streams=1), dict(mode=1, streams=2), dict(mode=2)):
                rgb, st = sg.render(W, H, spp, seed=seed, chunk_spp=5, **kw, **how)
                _assert_images_equal(rgb, img)
                assert st.rays == stats[1] and st.vertices == stats[3], (name, how)
            want = O.create_photon_map(so, 20000, seed)
            got = sg.create_photon_map(20000, seed)
            assert got[3] == want[3]
            for x, y in zip(got[:3], want[:3]):
                assert x.shape == y.shape and _rays.same_bits(x, y).all()
            a, b = _rays.connection_pairs(so, 20000, seed)
            assert np.array_equal(sg.check_connectable(a, b), so.check_connectable(a, b))
            print(name, "round", rnd, "ok: %d rays, %d samples, 20000 photons, %d pairs" % (len(org), W * H * spp, len(a)), flush=True)
    print("SOAK OK")


if __name__ == "__main__":
    main()
