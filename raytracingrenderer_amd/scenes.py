"""Scene descriptions as data: the reference's default scene and the synthetic variants the
measurements use.  A spec is a dict {materials, objects, camera} that
``HostScene.from_spec`` (and the test-side checkers) turn into a scene through the
reference's own construction calls; nothing here computes anything.
"""
import numpy as np

DIFFUSE, SPECULAR, MICROFACET = 0, 1, 2
f32 = np.float32


def default_scene_spec():
    """The scene of the reference's src/main.cpp:7-84 as data (materials, objects, camera)."""
    def v(*a):
        return np.array(a, dtype=f32)
    le = (f32(8.0) * v(f32(0.747) + f32(0.058), f32(0.747) + f32(0.258), f32(0.747))
          + f32(15.6) * v(f32(0.740) + f32(0.287), f32(0.740) + f32(0.160), f32(0.740))
          + f32(18.4) * v(f32(0.737) + f32(0.642), f32(0.737) + f32(0.159), f32(0.737)))
    z = v(0, 0, 0)
    d = 1e-5
    mats = [
        ("light", DIFFUSE, le, z, z, d, d),
        ("white", DIFFUSE, z, v(.6, .6, .6), z, d, d),
        ("red", DIFFUSE, z, v(.6, .01, .01), z, d, d),
        ("green", DIFFUSE, z, v(.01, .6, .01), z, d, d),
        ("blue", DIFFUSE, z, v(.01, .01, .6), z, d, d),
        ("yellow", DIFFUSE, z, v(.6, .6, .01), z, d, d),
        ("pink", DIFFUSE, z, v(.01, .6, .6), z, d, d),
        ("cyan", DIFFUSE, z, v(.6, .01, .6), z, d, d),
        ("spec", SPECULAR, z, v(1, 1, 1), z, d, d),
        ("micro", MICROFACET, z, v(.6, .01, .01), v(.6, .01, .01), 0.0, 0.5),
    ]
    cube = np.array([[0, 0, -512], [512, 0, -512], [0, 512, -512], [512, 512, -512],
                     [0, 0, 0], [512, 0, 0], [0, 512, 0], [512, 512, 0]], dtype=f32)
    objs = [
        ("sphere", "light", (256, 256, -256), 20),
        ("mesh", "red", cube, [0, 1, 2, 3, 2, 1]),
        ("mesh", "white", cube, [0, 4, 1, 5, 1, 4]),
        ("mesh", "yellow", cube, [2, 3, 6, 7, 6, 3]),
        ("mesh", "pink", cube, [0, 2, 4, 6, 4, 2]),
        ("mesh", "cyan", cube, [1, 5, 3, 7, 3, 5]),
        ("sphere", "red", (256, 512 - 64, -256), 32),
        ("sphere", "blue", (64, 256, -256), 32),
        ("sphere", "green", (512 - 64, 256, -256), 32),
        ("sphere", "micro", (256, 64, -256), 32),
        ("sphere", "spec", (64, 64, -512 + 64), 64),
        ("sphere", "spec", (64, 512 - 64, -512 + 64), 64),
        ("sphere", "spec", (512 - 64, 64, -512 + 64), 64),
        ("sphere", "spec", (512 - 64, 512 - 64, -512 + 64), 64),
    ]
    cam = dict(pos=(256, 256, 512), front=(0, 0, -1), up=(0, 1, 0), fov=45.0)
    return dict(materials=mats, objects=objs, camera=cam)


def variant_scene_spec():
    """The default geometry with its diffuse spheres switched to SPECULAR (red, blue) and
    MICROFACET (green): long specular chains for max_depth / rr_prob runs."""
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


def heightfield_spec(n, seed=1234):
    """The default scene plus one mesh of 2*n*n triangles: a jittered height field inside
    the box (n = 708 gives the ~1 M-triangle case, a nested tree about 20 levels deep)."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(40, 472, n + 1)
    zs = np.linspace(-472, -40, n + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = 60 + 25 * np.sin(X / 37.0) * np.cos(Z / 29.0) + rng.uniform(-1.5, 1.5, X.shape)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(f32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).reshape(-1)
    b, c = a + 1, a + (n + 1)
    d = c + 1
    idx = np.stack([a, b, c, b, d, c], 1).reshape(-1).astype(np.uint32)  # faces up
    spec = default_scene_spec()
    spec["objects"] = spec["objects"] + [("mesh", "white", verts, idx)]
    return spec


def interior_rays(n, seed, lo=(0, 0, -512), hi=(512, 512, 0)):
    """n rays with origins uniform in the box and isotropic, un-normalised directions."""
    rng = np.random.default_rng(seed)
    org = (rng.random((n, 3)) * (np.array(hi) - np.array(lo)) + np.array(lo)).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    return org, d  # deliberately NOT normalised: the path normalises internally
