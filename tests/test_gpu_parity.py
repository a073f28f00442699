"""GPU parity: the HIP path (through the C ABI of include/rtr.h) against the CPU
checker on the same seeded inputs.

Bar: closest-hit prim id, distance, position and normal are bit-exact; material
eval / pdf / sample, light sampling and the random stream are bit-exact; rendered
images are bit-exact against the checker's forward-accumulation renderer run
with the same counter-based stream and the fixed-sequence sin/cos (any NaN
equals any NaN).  The checker itself is pinned to the reference by
tests/test_oracle_vs_ref.py and tests/test_golden.py.
"""
import numpy as np
import pytest

import _rays
from _oracle import Oracle, default_scene_spec

pytestmark = pytest.mark.gpu

f32 = np.float32


@pytest.fixture(scope="module")
def rt():
    import raytracingrenderer_amd as rt
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return rt


@pytest.fixture(scope="module")
def scenes(rt):
    O = Oracle()
    so = O.default_scene()
    O.set_math(so, portable=True)
    sg = rt.default_scene()
    return O, so, sg


def _assert_hits_equal(got, want):
    pg, tg, posg, ng = got
    pw, tw, posw, nw = want
    assert np.array_equal(pg, pw), np.nonzero(pg != pw)[0][:10]
    hit = pw >= 0
    assert _rays.same_bits(tg, tw)[hit].all()
    assert _rays.same_bits(posg, posw)[hit].all()
    assert _rays.same_bits(ng, nw)[hit].all()


def test_closest_hit_default_scene(scenes):
    O, so, sg = scenes
    org, d = _rays.default_batch(O, so, n_interior=200000, prim_step=4)
    _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))


def test_closest_hit_all_primary_rays_784(scenes):
    O, so, sg = scenes
    org, d = _rays.primary_rays(O, so, 784, 784, 3)
    got = sg.intersect(org, d)
    _assert_hits_equal(got, so.intersect(org, d))
    # known answers recorded from the compiled reference
    o1, d1 = O.primary_ray(so, 784, 784, 392, 392)
    p, t, pos, n = sg.intersect([o1], [d1])
    assert p[0] == 0 and t[0] == f32(748.0) and tuple(pos[0]) == (256.0, 256.0, -236.0)
    o1, d1 = O.primary_ray(so, 784, 784, 0, 0)
    assert sg.intersect([o1], [d1])[1][0] == f32(653.986755)
    o1, d1 = O.primary_ray(so, 784, 784, 200, 200)
    assert sg.intersect([o1], [d1])[1][0] == f32(946.167053)


def _random_scene_spec(seed, n_tri=3000, n_sph=60):
    rng = np.random.default_rng(seed)
    spec = default_scene_spec()
    pts = (rng.uniform(0, 512, (n_tri, 3)) * np.array([1, 1, -1])).astype(f32)
    tri = np.stack([np.arange(n_tri), (np.arange(n_tri) + 1) % n_tri, (np.arange(n_tri) + 7) % n_tri], 1)
    # small triangles: move 2nd/3rd vertex near the first
    pts2 = pts.copy()
    verts = np.concatenate([pts, (pts + rng.normal(0, 12, pts.shape)).astype(f32),
                            (pts2 + rng.normal(0, 12, pts.shape)).astype(f32)]).astype(f32)
    idx = np.stack([np.arange(n_tri), np.arange(n_tri) + n_tri, np.arange(n_tri) + 2 * n_tri], 1)
    del tri
    objs = [("mesh", "white", verts, idx.reshape(-1).astype(np.uint32))]
    for k in range(n_sph):
        c = rng.uniform(0, 512, 3) * np.array([1, 1, -1])
        objs.append(("sphere", ["red", "spec", "micro", "light"][k % 4], tuple(c),
                     float(rng.uniform(2, 40))))
    spec["objects"] = objs
    return spec


def test_closest_hit_random_scene_deep_tree(rt):
    O = Oracle()
    spec = _random_scene_spec(9)
    so = O.build_scene(spec)
    sg = rt.HostScene.from_spec(spec)
    org, d = _rays.interior_rays(300000, seed=31)
    _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))
    # ragged ends of the ray stream: fewer rays than a wave, one more than a wave, ...
    for n in (1, 63, 64, 65, 257, 4097):
        _assert_hits_equal(sg.intersect(org[:n], d[:n]), so.intersect(org[:n], d[:n]))


def test_million_triangle_mesh(rt):
    """Default scene + a 708 x 708 jittered height field (1 002 528 triangles, 2 M nodes):
    closest hits of 400 000 rays and a small image, bit for bit; the general traversal
    (per-lane walk, geometry gathered from HBM) and both sampling loops."""
    O = Oracle()
    spec = _rays.heightfield_spec(708)
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    org, d = _rays.interior_rays(400000, seed=5)
    want = so.intersect(org, d)
    _assert_hits_equal(sg.intersect(org, d), want)
    assert (want[0] >= 19).mean() > 0.1  # tens of thousands of rays end on the mesh
    # a handful of rays that skim the field and walk thousands of nodes each: with so few
    # rays in the wave their node ranges are split over its idle lanes
    rng = np.random.default_rng(3)
    so_ = np.stack([rng.uniform(30, 60, 150), rng.uniform(50, 95, 150), rng.uniform(-60, -30, 150)], 1).astype(f32)
    sd_ = np.stack([rng.uniform(0.5, 1, 150), rng.uniform(-0.03, 0.03, 150), rng.uniform(-1, -0.5, 150)], 1).astype(f32)
    _assert_hits_equal(sg.intersect(so_, sd_), so.intersect(so_, sd_))
    img, stats = O.render(so, 32, 32, 4, seed=9, chunk_spp=2)
    for how in (dict(), dict(mode=1), dict(mode=2)):
        rgb, st = sg.render(32, 32, 4, seed=9, chunk_spp=2, **how)
        _assert_images_equal(rgb, img)
        assert st.rays == stats[1] and st.vertices == stats[3]
        assert st.mode == (how.get("mode") or rt.MODE_WAVEFRONT)  # a tree this large: wavefront by default


def test_closest_hit_degenerate_triangles_nan_distance(rt):
    """Zero-area and edge-on triangles give inf/NaN distances that still count as
    hits in the reference; the merge order then decides.  Must match bit for bit."""
    O = Oracle()
    spec = default_scene_spec()
    verts = np.array([[100, 100, -100], [200, 100, -100], [100, 200, -100],   # regular
                      [100, 100, -200], [200, 100, -200], [300, 100, -200],   # collinear
                      [50, 50, -50], [50, 50, -50], [50, 50, -50],            # a point
                      [0, 300, -300], [512, 300, -300], [256, 300, -100],     # in a y-plane
                      [256, 0, -300], [256, 512, -300], [256, 256, -100]], f32)  # in an x-plane
    idx = np.arange(15, dtype=np.uint32)
    spec["objects"] = spec["objects"] + [("mesh", "white", verts, idx)]
    so = O.build_scene(spec)
    sg = rt.HostScene.from_spec(spec)
    ao, ad = _rays.adversarial_rays()
    # rays lying inside the planes of the edge-on triangles
    extra_o = np.array([[0, 300, -200], [256, 100, -200], [0, 300, -300], [256, 0, -300],
                        [600, 300, -250], [256, 600, -250]], f32)
    extra_d = np.array([[1, 0, 0], [0, 1, 0], [1, 0, 1], [0, 1, 1], [-1, 0, 0.2], [0, -1, 0.2]], f32)
    io, idr = _rays.interior_rays(50000, seed=77)
    org = np.concatenate([ao, extra_o, io])
    d = np.concatenate([ad, extra_d, idr])
    want = so.intersect(org, d)
    _assert_hits_equal(sg.intersect(org, d), want)


def test_degenerate_triangles_in_a_large_scene(rt):
    """The same inf/NaN distances where traversal streams rays through a tree in memory:
    rays that met a NaN are re-traced with the literal merge order."""
    O = Oracle()
    spec = _random_scene_spec(5)
    verts = np.array([[100, 100, -200], [200, 100, -200], [300, 100, -200],   # collinear
                      [50, 50, -50], [50, 50, -50], [50, 50, -50],            # a point
                      [0, 300, -300], [512, 300, -300], [256, 300, -100],     # in a y-plane
                      [256, 0, -300], [256, 512, -300], [256, 256, -100]], f32)  # in an x-plane
    spec["objects"] = spec["objects"] + [("mesh", "white", verts, np.arange(12, dtype=np.uint32))]
    so = O.build_scene(spec)
    sg = rt.HostScene.from_spec(spec)
    ao, ad = _rays.adversarial_rays()
    extra_o = np.array([[0, 300, -200], [256, 100, -200], [0, 300, -300], [256, 0, -300],
                        [600, 300, -250], [256, 600, -250]], f32)
    extra_d = np.array([[1, 0, 0], [0, 1, 0], [1, 0, 1], [0, 1, 1], [-1, 0, 0.2], [0, -1, 0.2]], f32)
    io, idr = _rays.interior_rays(20000, seed=78)
    # rays inside the planes of the edge-on triangles, many of them, from all around
    rng = np.random.default_rng(8)
    k = 3000
    ang = rng.uniform(0, 2 * np.pi, k)
    po = np.stack([rng.uniform(0, 512, k), np.full(k, 300.0), rng.uniform(-512, 0, k)], 1).astype(f32)
    pd = np.stack([np.cos(ang), np.zeros(k), np.sin(ang)], 1).astype(f32)
    org = np.concatenate([ao, extra_o, io, po])
    d = np.concatenate([ad, extra_d, idr, pd])
    want = so.intersect(org, d)
    _assert_hits_equal(sg.intersect(org, d), want)
    assert np.isnan(want[1][want[0] >= 0]).any() or np.isinf(want[1][want[0] >= 0]).any()


def test_photon_map_large_scene_and_no_lights(rt):
    """Render::createPhotonMap where traversal walks a tree in memory, on the scene whose
    spheres are mirrors and microfacet (photons pass through non-diffuse vertices without
    being stored), and on a scene without lights (the reference leaves its map empty)."""
    O = Oracle()
    for spec, n in ((_random_scene_spec(9), 30000), (_rays.heightfield_spec(40), 8000)):
        so = O.build_scene(spec)
        O.set_math(so, portable=True)
        sg = rt.HostScene.from_spec(spec)
        want = O.create_photon_map(so, n, 11)
        got = sg.create_photon_map(n, 11)
        assert got[3] == want[3]
        for a, b in zip(got[:3], want[:3]):
            assert a.shape == b.shape and _rays.same_bits(a, b).all()
    spec = default_scene_spec()
    spec["objects"] = [ob for ob in spec["objects"] if ob[1] != "light"]
    sg = rt.HostScene.from_spec(spec)
    got = sg.create_photon_map(100, 1)
    assert len(got[0]) == 0 and got[3] == 0


def _scaled(spec, k):
    out = dict(spec)
    cam = dict(spec["camera"]) if isinstance(spec["camera"], dict) else spec["camera"]
    objs = []
    for ob in spec["objects"]:
        if ob[0] == "sphere":
            objs.append(("sphere", ob[1], tuple(float(c) * k for c in ob[2]), float(ob[3]) * k))
        else:
            objs.append(("mesh", ob[1], (np.asarray(ob[2], f32) * f32(k)).astype(f32), ob[3]))
    out["objects"] = objs
    out["camera"] = cam
    return out


def test_scenes_outside_the_division_free_ranges(rt):
    """The division-free box test and sphere roots are only used while box planes, origins
    and quotients stay inside fixed ranges.  Scenes scaled by 2^23 (coordinates up to 2^32)
    and by 2^-27 (down to 2^-21) leave them, partly or wholly; every ray then takes the
    IEEE divisions.  Same hits as the checker, bit for bit."""
    O = Oracle()
    for base in (default_scene_spec(), _random_scene_spec(4, n_tri=800, n_sph=20)):
        for k in (2.0 ** 23, 2.0 ** -27, 2.0 ** 19, 2.0 ** -12):
            spec = _scaled(base, k)
            so = O.build_scene(spec)
            sg = rt.HostScene.from_spec(spec)
            org, d = _rays.interior_rays(40000, seed=9)
            org = (org * f32(k)).astype(f32)
            want = so.intersect(org, d)
            _assert_hits_equal(sg.intersect(org, d), want)
            assert (want[0] >= 0).mean() > 0.02   # the reference itself loses most hits at 2^23


def _unnest(O, so, sg, seed, share):
    """Shrinks and shifts a share of the nodes' boxes (the same in the checker's and the
    library's scene), so that boxes no longer contain their children's."""
    rng = np.random.default_rng(seed)
    lo, hi, child, prim = sg.flatten()
    n = len(prim)
    for k in rng.choice(n, max(1, int(share * n)), replace=False):
        ext = hi[k] - lo[k]
        nlo = (lo[k] + rng.uniform(-0.2, 0.6, 3) * ext).astype(f32)
        nhi = np.maximum(nlo, (hi[k] - rng.uniform(-0.2, 0.6, 3) * ext).astype(f32))
        sg.set_node_box(k, nlo, nhi)
        O.set_node_box(so, k, nlo, nhi)


def test_trees_whose_boxes_are_not_nested(rt):
    """The reference tests whatever box a node carries, and reaches a child only through
    its ancestors' boxes.  With boxes that do not contain their children's, the short cuts
    that rely on nesting (leaf box alone decides; a walk may start at any node) must be off:
    closest hits, visibility and images still equal the checker's, on a scene staged in LDS
    and on one walked in memory."""
    O = Oracle()
    for spec, share in ((default_scene_spec(), 0.3), (_random_scene_spec(3), 0.1)):
        so = O.build_scene(spec)
        O.set_math(so, portable=True)
        sg = rt.HostScene.from_spec(spec)
        _unnest(O, so, sg, 17, share)
        org, d = _rays.interior_rays(150000, seed=41)
        want = so.intersect(org, d)
        _assert_hits_equal(sg.intersect(org, d), want)
        nested = rt.HostScene.from_spec(spec).intersect(org, d)
        assert (nested[0] != want[0]).mean() > 0.01    # the boxes do cut hits away
        a, b = _rays.connection_pairs(so, 20000, 5)
        assert np.array_equal(sg.check_connectable(a, b), so.check_connectable(a, b))
        img, stats = O.render(so, 40, 30, 4, seed=2, chunk_spp=2)
        for how in MODES:
            rgb, st = sg.render(40, 30, 4, seed=2, chunk_spp=2, **how)
            _assert_images_equal(rgb, img)
            assert st.rays == stats[1] and st.vertices == stats[3]
        want = O.create_photon_map(so, 3000, 6)
        got = sg.create_photon_map(3000, 6)
        for x, y in zip(got[:3], want[:3]):
            assert x.shape == y.shape and _rays.same_bits(x, y).all()


def test_primitives_beyond_their_boxes(rt):
    """The reference never checks that a leaf's box holds its primitive.  A triangle with a vertex
    at 3e30 under boxes clamped into the division-free test's range: planes in range, boxes nested,
    so only the primitives' own coordinates tell that this scene's arithmetic can overflow (products
    of such coordinates are infinite).  The fused kernel then has to run the flavour that watches
    for infinite and undefined distances; hits, images and counters equal the checker's."""
    spec = default_scene_spec()
    far = np.array([[100, 100, -100], [3e30, 120, -150], [140, 3e30, -120]], f32)
    spec["objects"] = spec["objects"] + [("mesh", "white", far, [0, 1, 2])]
    O = Oracle()
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    lo, hi, child, prim = sg.flatten()
    for k in range(len(prim)):
        nlo, nhi = np.clip(lo[k], -1000, 1000), np.clip(hi[k], -1000, 1000)
        sg.set_node_box(k, nlo, nhi)
        O.set_node_box(so, k, nlo, nhi)
    nodes, parent, leaves, (fast, nested) = sg.read_tree()
    assert fast == 1 and nested == 1
    org, d = _rays.default_batch(O, so, n_interior=30000, prim_step=16)
    want = so.intersect(org, d)
    _assert_hits_equal(sg.intersect(org, d), want)
    assert (want[0] == 19).sum() > 50      # the far triangle is hit, where it crosses the box
    img, stats = O.render(so, 48, 48, 8, seed=3, chunk_spp=4)
    for how in MODES:
        rgb, st = sg.render(48, 48, 8, seed=3, chunk_spp=4, **how)
        _assert_images_equal(rgb, img)
        assert st.rays == stats[1] and st.vertices == stats[3]


def test_check_connectable(rt, scenes):
    """Scene::checkConnectable on the default scene (all-leaves traversal) and on a scene
    with thousands of primitives (tree walk): same booleans as the checker."""
    O, so, sg = scenes
    a, b = _rays.connection_pairs(so, 100000, 11)
    want = so.check_connectable(a, b)
    assert np.array_equal(sg.check_connectable(a, b), want)
    assert 0.2 < want.mean() < 0.8
    spec = _random_scene_spec(9)
    so2 = O.build_scene(spec)
    sg2 = rt.HostScene.from_spec(spec)
    a, b = _rays.connection_pairs(so2, 50000, 12)
    assert np.array_equal(sg2.check_connectable(a, b), so2.check_connectable(a, b))
    assert len(sg.check_connectable(np.zeros((0, 3), f32), np.zeros((0, 3), f32))) == 0


def test_empty_and_tiny_batches(scenes):
    O, so, sg = scenes
    p, t, pos, n = sg.intersect(np.zeros((0, 3), f32), np.zeros((0, 3), f32))
    assert len(p) == 0
    org, d = _rays.interior_rays(1, seed=1)
    _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))
    org, d = _rays.interior_rays(257, seed=2)  # ragged: one full block + one lane
    _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))


def test_scene_without_objects(rt):
    spec = default_scene_spec()
    spec["objects"] = []
    sg = rt.HostScene.from_spec(spec)
    org, d = _rays.interior_rays(100, seed=1)
    p, t, pos, n = sg.intersect(org, d)
    assert (p == -1).all()
    rgb, st = sg.render(16, 16, 2)
    assert (rgb == 0).all()


def _dirs(n, seed):
    return np.random.default_rng(seed).normal(size=(n, 3)).astype(f32)


def test_material_eval_pdf_sample(scenes):
    O, so, sg = scenes
    n = 100000
    wi, wo, nn = _dirs(n, 1), _dirs(n, 2), _dirs(n, 3)
    nu = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    wiu = wi / np.linalg.norm(wi, axis=1, keepdims=True)
    wo[: n // 4] = (2 * (nu * wiu).sum(1, keepdims=True) * nu - wiu)[: n // 4].astype(f32)
    u2 = np.random.default_rng(4).random((n, 2)).astype(f32)
    for m in ["white", "red", "spec", "micro", "light"]:
        f, pdf = sg.material_eval(m, wi, wo, nn)
        assert _rays.same_bits(f, so.material_eval(m, wi, wo, nn)).all(), m
        assert _rays.same_bits(pdf, so.material_pdf(m, wi, wo, nn)).all(), m
        got = sg.material_sample(m, wi, nu.astype(f32), u2)
        assert _rays.same_bits(got, O.material_sample(so, m, wi, nu.astype(f32), u2)).all(), m


def test_stream_is_philox4x32_10(scenes):
    O, so, sg = scenes
    rng = np.random.default_rng(8)
    n = 5000
    px, sa, bo = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for _ in range(3)]
    dim = rng.integers(0, 40, n).astype(np.uint32)
    got = sg.stream_uniform(1234567, px, sa, bo, dim)
    want = np.array([O.lib.orc_stream_uniform(1234567, int(px[k]), int(sa[k]), int(bo[k]), int(dim[k]))
                     for k in range(n)], f32)
    assert np.array_equal(got, want)
    assert (got >= 0).all() and (got < 1).all()


def test_sample_light_default_and_multi_light(rt, scenes):
    O, so, sg = scenes
    rng = np.random.default_rng(3)
    u = rng.random((4000, 12)).astype(f32)
    u[:50, 0] = 0  # pick < 1e-3: the loop's first branch
    pos, nrm, pdf, prim, used = sg.sample_light(u)
    for k in range(len(u)):
        po, no, pdo, primo, usedo = O.sample_light(so, u[k])
        assert primo == prim[k] and usedo == used[k]
        assert _rays.same_bits(po, pos[k]).all() and _rays.same_bits(no, nrm[k]).all()
        assert f32(pdo).view(np.uint32) == pdf[k].view(np.uint32)
    # two sphere lights + an emissive two-triangle plate (src/main.cpp:42-44's commented light)
    spec = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    spec["objects"] = [("sphere", "light", (100, 300, -200), 15)] + spec["objects"] + \
                      [("mesh", "light", plate, [0, 1, 2, 3, 2, 1])]
    so2 = O.build_scene(spec)
    O.set_math(so2, portable=True)
    sg2 = rt.HostScene.from_spec(spec)
    u = rng.random((3000, 16)).astype(f32)
    u[:100, 0] *= 1e-7
    pos, nrm, pdf, prim, used = sg2.sample_light(u)
    seen = set()
    for k in range(len(u)):
        po, no, pdo, primo, usedo = O.sample_light(so2, u[k])
        assert primo == prim[k] and usedo == used[k], (k, primo, prim[k], usedo, used[k])
        assert _rays.same_bits(po, pos[k]).all() and _rays.same_bits(no, nrm[k]).all()
        assert f32(pdo).view(np.uint32) == pdf[k].view(np.uint32)
        seen.add(primo)
    assert len(seen) >= 2


def _assert_images_equal(got, want):
    eq = _rays.same_bits(got, want)
    assert eq.all(), (np.nonzero(~eq.all(1))[0][:10], got[~eq.all(1)][:3], want[~eq.all(1)][:3])


# the sampling loop has two implementations (rtr_render_params.mode); the wavefront
# one can drive its pool on one stream or two
MODES = [dict(mode=2), dict(mode=1, streams=1), dict(mode=1, streams=2)]
MODE_IDS = ["fused", "wavefront-1-stream", "wavefront-2-streams"]


@pytest.mark.parametrize("how", MODES, ids=MODE_IDS)
@pytest.mark.parametrize("W,H,spp,chunk", [(64, 48, 8, 3), (33, 17, 5, 0), (96, 96, 16, 4)])
def test_render_bit_exact_vs_checker(scenes, W, H, spp, chunk, how):
    O, so, sg = scenes
    rgb, st = sg.render(W, H, spp, seed=7, chunk_spp=chunk, **how)
    assert st.mode == how["mode"]
    want, stats = O.render(so, W, H, spp, seed=7, chunk_spp=st.chunk_spp)
    _assert_images_equal(rgb, want)
    assert st.samples == W * H * spp == stats[0]
    assert st.rays == stats[1]          # same rays traced, one by one
    assert st.vertices == stats[3]
    assert rgb.mean() > 0.05


def test_full_resolution_image_bit_exact(scenes):
    """The benchmark's image size, 784 x 784, at 48 samples per pixel (about 100 M rays): every pixel
    and both counters equal to the checker's, fused kernel and wavefront loop.  (The full 1024 spp
    take the checker a minute on sixteen cores: tools/big_bitexact.py 784 1024 default.)"""
    O, so, sg = scenes
    want, stats = O.render(so, 784, 784, 48, seed=3, chunk_spp=8, threads=16)
    for how in (dict(mode=2), dict(mode=1)):
        rgb, st = sg.render(784, 784, 48, seed=3, chunk_spp=8, **how)
        _assert_images_equal(rgb, want)
        assert st.rays == stats[1] and st.vertices == stats[3]


@pytest.mark.parametrize("how", MODES, ids=MODE_IDS)
def test_render_edge_parameters(rt, scenes, how):
    """One pixel, one sample; a chunk longer than the sample count; paths cut at the first
    vertex; roulette that never stops a path; a partition that owns no column."""
    O, so, sg = scenes
    cases = [dict(W=1, H=1, spp=1), dict(W=8, H=8, spp=5, chunk_spp=7),
             dict(W=24, H=16, spp=4, max_depth=1), dict(W=24, H=16, spp=3, max_depth=4, rr_prob=1.0)]
    for cs in cases:
        cs = dict(cs)
        W, H, spp = cs.pop("W"), cs.pop("H"), cs.pop("spp")
        rgb, st = sg.render(W, H, spp, seed=13, **cs, **how)
        okw = dict(cs)
        okw["chunk_spp"] = st.chunk_spp
        want, stats = O.render(so, W, H, spp, seed=13, **okw)
        _assert_images_equal(rgb, want)
        assert (st.rays, st.vertices) == (stats[1], stats[3])
    rgb, st = sg.render(8, 8, 2, part_index=3, part_count=4, tile_cols=8, **how)
    assert rgb.shape == (0, 3) and st.samples == 0


def test_render_independent_of_pool_and_partition(rt, scenes):
    O, so, sg = scenes
    W, H, spp = 80, 40, 6
    full, st = sg.render(W, H, spp, seed=3, chunk_spp=2)
    assert st.mode == rt.MODE_FUSED  # what the default scene gets by default
    for pool in (512, 4096):
        small_pool, st = sg.render(W, H, spp, seed=3, chunk_spp=2, pool_size=pool, mode=rt.MODE_WAVEFRONT)
        assert st.mode == rt.MODE_WAVEFRONT
        _assert_images_equal(small_pool, full)
    R = 3
    out = np.zeros_like(full).reshape(W, H, 3)
    for r in range(R):
        part, st = sg.render(W, H, spp, seed=3, chunk_spp=2, part_index=r, part_count=R, tile_cols=8)
        cols = rt.partition_columns(W, r, R, 8)
        assert st.n_columns == len(cols)
        out[cols] = part.reshape(len(cols), H, 3)
    _assert_images_equal(out.reshape(-1, 3), full)


def test_render_variant_scene_depth_cap_and_rr(rt):
    """Specular / microfacet heavy variant, max depth 16, continue probability 0.95."""
    O = Oracle()
    spec = default_scene_spec()
    objs = []
    for ob in spec["objects"]:
        if ob[0] == "sphere" and ob[1] in ("red", "blue"):
            ob = ("sphere", "spec", ob[2], ob[3])
        if ob[0] == "sphere" and ob[1] == "green":
            ob = ("sphere", "micro", ob[2], ob[3])
        objs.append(ob)
    spec["objects"] = objs
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    want, stats = O.render(so, 48, 48, 6, seed=11, max_depth=16, rr_prob=0.95, chunk_spp=2)
    for how in MODES:
        rgb, st = sg.render(48, 48, 6, seed=11, max_depth=16, rr_prob=0.95, chunk_spp=2, **how)
        _assert_images_equal(rgb, want)
        assert st.vertices == stats[3]
        assert st.vertices / st.samples > 4  # long paths


def test_render_with_mesh_light(rt):
    O = Oracle()
    spec = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    back = np.array([[220, 21, -288], [292, 21, -288], [220, 21, -224], [292, 21, -224]], f32)
    spec["objects"] = spec["objects"][1:] + [("mesh", "light", plate, [0, 1, 2, 3, 2, 1]),
                                             ("mesh", "white", back, [0, 2, 1, 3, 1, 2])]
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    want, stats = O.render(so, 40, 40, 8, seed=2, chunk_spp=4)
    for how in MODES:
        rgb, st = sg.render(40, 40, 8, seed=2, chunk_spp=4, **how)
        _assert_images_equal(rgb, want)
    assert rgb.mean() > 0.001


def test_render_scene_too_large_for_lds(rt):
    """A scene with a few thousand triangles takes the general traversal (per-lane tree
    walk, geometry from memory); its tree is small enough that the fused kernel is the
    default.  Every loop variant must give the same image."""
    O = Oracle()
    spec = _random_scene_spec(5)
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    want, stats = O.render(so, 40, 32, 6, seed=4, chunk_spp=3)
    rgb, st = sg.render(40, 32, 6, seed=4, chunk_spp=3)
    assert st.mode == rt.MODE_FUSED
    _assert_images_equal(rgb, want)
    assert st.rays == stats[1] and st.vertices == stats[3]
    for how in MODES:
        rgb, st = sg.render(40, 32, 6, seed=4, chunk_spp=3, **how)
        _assert_images_equal(rgb, want)
        assert st.rays == stats[1] and st.vertices == stats[3]


def test_fused_and_wavefront_agree_at_many_samples(scenes):
    """The two loops share no scheduling code; 10^8 rays through both must give the same
    bits (the checker is too slow for this size)."""
    O, so, sg = scenes
    a, sa = sg.render(256, 256, 256, seed=21, mode=2)
    sg.render(256, 256, 256, seed=22, mode=2)  # overwrite every buffer the library keeps
    b, sb = sg.render(256, 256, 256, seed=21, mode=1)
    _assert_images_equal(a, b)
    assert sa.rays == sb.rays and sa.vertices == sb.vertices and sa.rays > 5e7


def test_benchmark_size_image_properties(rt, scenes):
    """The benchmark's own workload, 784 x 784 at 1024 spp (6.3e8 samples, 2e9 rays), which
    the CPU checker cannot reach: the two sampling loops, which share no scheduling code,
    must agree in every bit, and so must the image assembled from 8 column partitions."""
    O, so, sg = scenes
    W = H = 784
    spp = 1024
    full, st = sg.render(W, H, spp, seed=0)
    assert st.mode == rt.MODE_FUSED and st.samples == W * H * spp
    assert 3.0 < st.rays / st.samples < 3.5 and 1.9 < st.vertices / st.samples < 2.2
    sg.render(W, H, 8, seed=5)  # overwrite the library's buffers in between
    wf, sw = sg.render(W, H, spp, seed=0, mode=rt.MODE_WAVEFRONT)
    _assert_images_equal(wf, full)
    assert (sw.rays, sw.vertices) == (st.rays, st.vertices)
    out = np.zeros_like(full).reshape(W, H, 3)
    rays = 0
    for r in range(8):
        part, sp = sg.render(W, H, spp, seed=0, part_index=r, part_count=8, tile_cols=7)
        cols = rt.partition_columns(W, r, 8, 7)
        assert len(cols) == W // 8
        out[cols] = part.reshape(len(cols), H, 3)
        rays += sp.rays
    _assert_images_equal(out.reshape(-1, 3), full)
    assert rays == st.rays
    # light pixels are exact, and the picture is what the reference converges to
    lum = (full * np.array([0.2126, 0.7152, 0.0722], f32)).sum(1)
    assert 0.15 < lum.mean() < 0.18  # the reference's mean image is (0.224, 0.152, 0.119)


def test_host_api_render_and_ppm(rt, scenes, tmp_path):
    """Render::renderPathTracing of the host library: same buffer as the C ABI call,
    and the PPM bytes follow src/Render.cpp:3-21 (header 'height width', gamma 0.6)."""
    O, so, sg = scenes
    W, H, spp = 24, 20, 4
    rgb, _ = sg.render(W, H, spp, seed=5)
    rgb2, st = sg.render_via_host_api(W, H, spp, file=str(tmp_path / "img"), seed=5)
    _assert_images_equal(rgb2, rgb)
    data = (tmp_path / "img.ppm").read_bytes()
    header = b"P6\n%d %d\n255\n" % (H, W)
    assert data.startswith(header)
    body = np.frombuffer(data[len(header):], np.uint8).reshape(-1, 3)
    want = np.array([[O.lib.orc_tonemap(float(v)) for v in px] for px in rgb], np.uint8)
    assert np.array_equal(body, want)


def test_multi_device_render_reports_its_partitions_times(rt):
    sg = rt.default_scene()
    _, st = sg.render_multi(96, 64, 8, [0, 0, 0], seed=1)
    assert st.n_devices == 3 and 0 < st.ms_render_min <= st.ms_render_max <= st.ms_total
    _, st1 = sg.render(96, 64, 8, seed=1)
    assert st1.ms_render_min == st1.ms_render_max == st1.ms_total


def test_multi_device_render_after_the_scene_changed(rt):
    """render_multi caches one upload per device; a change of the scene (here: one node's box
    overwritten so that a wall triangle is no longer reached) must drop them, or the next render_multi
    shows the old geometry."""
    sg = rt.default_scene()
    W, H, spp = 64, 48, 4
    before, _ = sg.render_multi(W, H, spp, [0, 0], seed=4)
    lo, hi, child, prim = sg.flatten()
    leaf = int(np.nonzero(prim == 6)[0][0])  # the leaf of a wall triangle: a box far away hides it
    sg.set_node_box(leaf, [1e6, 1e6, 1e6], [1e6 + 1, 1e6 + 1, 1e6 + 1])
    single, _ = sg.render(W, H, spp, seed=4)
    after, _ = sg.render_multi(W, H, spp, [0, 0], seed=4)
    _assert_images_equal(after, single)
    assert not _rays.same_bits(after, before).all()


def test_host_api_fewer_than_sixteen_rows(rt, scenes):
    """src/Render.cpp:669-684: below 16 rows the reference renders row 0 only and leaves the other
    rows zero (tests/test_oracle_vs_ref.py::test_fewer_than_sixteen_rows).  The C ABI renders every
    row; Render::renderPathTracing of the host mirror applies the reference's rule."""
    O, so, sg = scenes
    W, H, spp = 40, 8, 3
    full, _ = sg.render(W, H, spp, seed=2)
    host, _ = sg.render_via_host_api(W, H, spp, seed=2)
    full, host = full.reshape(W, H, 3), host.reshape(W, H, 3)
    assert full[:, 1:].any()
    assert not host[:, 1:].any()
    _assert_images_equal(host[:, 0], full[:, 0])
    host16, _ = sg.render_via_host_api(W, 16, spp, seed=2)
    full16, _ = sg.render(W, 16, spp, seed=2)
    _assert_images_equal(host16, full16)


# ---------------- several devices behind one call (rtr_render_pt_multi) ----------------
def test_multi_device_render_equals_single_device(rt, scenes):
    """Virtual ranks: three uploads of the scene on device 0 render three partitions on three
    host threads and streams, the partitions travel to the first one's buffer by peer copy
    and one kernel restores column order.  The image must be the one-device image bit for
    bit, in host memory and in device memory, and through the C++ mirror's device list."""
    import torch
    O, so, sg = scenes
    W, H, spp = 100, 36, 6
    full, st = sg.render(W, H, spp, seed=9, chunk_spp=2)
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 5):
        got, sm = sg.render_multi(W, H, spp, devs, seed=9, chunk_spp=2)
        _assert_images_equal(got, full)
        assert sm.n_devices == len(devs) and sm.n_columns == W
        assert (sm.samples, sm.rays, sm.vertices) == (st.samples, st.rays, st.vertices)
    # explicit tiles, one of them narrow, more partitions than tiles
    got, sm = sg.render_multi(W, H, spp, [0] * 4, seed=9, chunk_spp=2, tile_cols=48)
    _assert_images_equal(got, full)
    # the wavefront loop under the same call
    got, sm = sg.render_multi(W, H, spp, [0, 0], seed=9, chunk_spp=2, mode=rt.MODE_WAVEFRONT)
    _assert_images_equal(got, full)
    assert sm.mode == rt.MODE_WAVEFRONT
    # device output
    out = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
    sg.render_multi(W, H, spp, [0, 0, 0], d_ptr=out.data_ptr(), seed=9, chunk_spp=2)
    torch.cuda.synchronize()
    _assert_images_equal(out.cpu().numpy().reshape(-1, 3), full)
    # Render::renderPathTracing with RenderOptions::devices
    got, sh = sg.render_via_host_api(W, H, spp, devices=[0, 0, 0], seed=9, chunk_spp=2)
    _assert_images_equal(got, full)
    assert sh.n_devices == 3
    got, sh = sg.render_via_host_api(W, H, spp, seed=9, chunk_spp=2)  # and back to one device
    _assert_images_equal(got, full)
    assert sh.n_devices == 1


def test_multi_device_argument_errors(rt, scenes):
    import ctypes as C
    O, so, sg = scenes
    with pytest.raises(rt.RtrError, match="part_count"):
        sg.render_multi(16, 16, 2, [0, 0], part_count=2)
    with pytest.raises(rt.RtrError):
        sg.render_multi(16, 16, 2, [rt.device_count() + 3])
    from raytracingrenderer_amd import api
    h = sg.device_scenes([0])[0]
    arr = (C.c_void_p * 2)(h.value, h.value)
    p = sg.make_params(16, 16, 2)
    p.part_count = 0
    cam = sg.camera_struct()
    out = np.zeros((256, 3), f32)
    rc = api.hip_lib().rtr_render_pt_multi(arr, 2, C.byref(cam), C.byref(p),
                                           out.ctypes.data_as(C.POINTER(C.c_float)), None)
    assert rc == -1 and b"twice" in api.hip_lib().rtr_last_error()


def test_render_2048_square_image(rt, scenes):
    """2048 x 2048 (4.2 M pixels, the multi-GPU configuration's image): the fused kernel at
    the full 4096 spp (1.7e10 samples), its 8-partition reassembly through the multi-device
    call at 512 spp, and the wavefront loop at 64 spp must agree bit for bit wherever they
    render the same thing; default chunking keeps the partial sums at 2 GiB."""
    O, so, sg = scenes
    W = H = 2048
    lumw = np.array([0.2126, 0.7152, 0.0722], f32)
    a, sa = sg.render(W, H, 64, seed=2)
    assert sa.mode == rt.MODE_FUSED and sa.samples == W * H * 64
    assert W * H * ((64 + sa.chunk_spp - 1) // sa.chunk_spp) <= 1 << 27
    b, sb = sg.render(W, H, 64, seed=2, mode=rt.MODE_WAVEFRONT, chunk_spp=sa.chunk_spp)
    _assert_images_equal(b, a)
    assert (sb.rays, sb.vertices) == (sa.rays, sa.vertices)
    del b
    c, sc = sg.render_multi(W, H, 64, [0] * 8, seed=2)  # default chunking: sized by the whole image
    assert sc.chunk_spp == sa.chunk_spp
    _assert_images_equal(c, a)
    assert sc.rays == sa.rays
    del c
    full, st = sg.render(W, H, 4096, seed=0)
    assert st.samples == W * H * 4096 and st.chunk_spp == 128
    assert 3.0 < st.rays / st.samples < 3.5
    lum = (full * lumw).sum(1)
    assert np.isfinite(full).all() and 0.15 < lum.mean() < 0.18
    # the light's pixels are exact: emission, summed and divided (src/Render.cpp:85-87)
    centre = full.reshape(W, H, 3)[W // 2, H // 2]
    assert np.allclose(centre, [47.834801, 38.566399, 31.080799], rtol=1e-6)
    # 4096 spp against 64 spp of another seed: same picture, less noise
    assert abs(lum.mean() - (a * lumw).sum(1).mean()) / lum.mean() < 0.01


def test_count_work_matches_the_checkers_counters(rt, scenes):
    """rtr_trace_count_work (what bench.py prices the traversal kernels against) counts the
    same box / triangle / sphere tests as the instrumented CPU traversal."""
    import torch
    O, so, sg = scenes
    spec = None
    for scene_o, scene_g, n in ((so, sg, 20000), (None, None, 3000)):
        if scene_o is None:
            spec = _random_scene_spec(6, n_tri=700, n_sph=20)
            scene_o, scene_g = O.build_scene(spec), rt.HostScene.from_spec(spec)
        org, d = _rays.interior_rays(n, seed=77)
        O.counters(reset=True)
        scene_o.intersect(org, d)
        want = O.counters(reset=True)
        o_t = torch.from_numpy(org).cuda()
        d_t = torch.from_numpy(d).cuda()
        got = scene_g.count_work(o_t.data_ptr(), d_t.data_ptr(), n)
        assert want[0] == n and tuple(int(x) for x in want[2:4]) == got[1:]
        # the reference tests a mesh's box twice (as the leaf of the top tree and as the root
        # of the mesh's own tree, include/Trangle.h:56,99-101); the flattened tree holds it once
        n_meshes = sum(1 for ob in (spec if scene_o is not so else default_scene_spec())["objects"] if ob[0] == "mesh")
        assert got[0] <= want[1] <= got[0] + n_meshes * n
        assert got[0] > 10 * n


def test_device_self_test_of_arithmetic_shortcuts(rt):
    """The short reciprocal sequence of the triangle test equals 1.0f / d for every float."""
    assert rt.self_test(0) == 0


# ---------------- scene upload: the tree is checked and laid out on the device ----------------
def _expected_layout(lo, hi, child, prim_of_node, kind_row):
    """Depth-first pre-order renumbering of a flattened tree, as plain Python."""
    n = len(child)
    is_child = np.zeros(n, bool)
    for l, r in child:
        if l >= 0:
            is_child[l] = is_child[r] = True
    root = int(np.nonzero(~is_child)[0][0])
    order, parent_old, stack = [], {root: -1}, [root]
    while stack:
        o = stack.pop()
        order.append(o)
        l, r = child[o]
        if l >= 0:
            parent_old[l] = parent_old[r] = o
            stack += [r, l]
    place = {o: i for i, o in enumerate(order)}
    size = {}
    for o in reversed(order):
        l, r = child[o]
        size[o] = 1 if l < 0 else 1 + size[l] + size[r]
    nodes = np.zeros((n, 8), f32)
    parent = np.zeros(n, np.int32)
    leaf_nodes = []
    for i, o in enumerate(order):
        nodes[i, 0:3], nodes[i, 4:7] = lo[o], hi[o]
        leaf = kind_row[prim_of_node[o]] if child[o][0] < 0 else -1
        nodes[i, 3] = np.array([i + size[o]], np.int32).view(f32)[0]
        nodes[i, 7] = np.array([leaf], np.int32).view(f32)[0]
        parent[i] = -1 if parent_old[o] < 0 else place[parent_old[o]]
        if child[o][0] < 0:
            leaf_nodes.append(i)
    return root, nodes, parent, leaf_nodes


@pytest.mark.parametrize("which", ["default", "random", "heightfield"])
def test_device_tree_layout(rt, which):
    """rtr_scene_create renumbers the tree, computes skip links, parents and leaf records in
    kernels; every array must equal a sequential depth-first walk of the same tree."""
    spec = {"default": default_scene_spec, "random": lambda: _random_scene_spec(11, n_tri=900, n_sph=40),
            "heightfield": lambda: _rays.heightfield_spec(96)}[which]()
    sg = rt.HostScene.from_spec(spec)
    lo, hi, child, prim = sg.flatten()
    # prim id -> (kind << 30) | row: spheres and triangles are numbered in prim order per kind
    import ctypes as C
    nodes, parent, leaves, (fast, nested) = sg.read_tree()
    assert fast == 1 and nested == 1
    payload = nodes[:, 7].view(np.int32)
    skip = nodes[:, 3].view(np.int32)
    # expected, from the flattened arrays alone
    n = len(child)
    is_leaf = child[:, 0] < 0
    # kind/row of every leaf node come from the device's own payloads via the prim ids:
    # check them against a second source, the closest-hit entry point, below
    leaf_payload_by_prim = {}
    want_root, want_nodes, want_parent, want_leaf_nodes = _expected_layout(
        lo, hi, child, prim, {p: -2 for p in set(prim[is_leaf].tolist())})
    assert np.array_equal(nodes[:, 0:3].view(np.uint32), want_nodes[:, 0:3].view(np.uint32))
    assert np.array_equal(nodes[:, 4:7].view(np.uint32), want_nodes[:, 4:7].view(np.uint32))
    assert np.array_equal(skip, want_nodes[:, 3].view(np.int32))
    assert np.array_equal(parent, want_parent)
    assert np.array_equal(np.nonzero(payload >= 0)[0], np.array(want_leaf_nodes))
    assert len(leaves) == len(want_leaf_nodes)
    # leaf records: node index and payload agree with the node array; triangle edges are
    # p1 - p0 and p2 - p0 of the triangle table entry the payload names
    rec_node = leaves[:, 14].view(np.int32)
    rec_leaf = leaves[:, 15].view(np.int32)
    assert np.array_equal(rec_node, np.array(want_leaf_nodes, np.int32))
    assert np.array_equal(rec_leaf, payload[rec_node])
    sph = rec_leaf >> 30 == 1
    assert np.array_equal(leaves[sph, 4], leaves[sph, 3] * leaves[sph, 3])
    T = leaves[sph, 5]
    r = leaves[sph, 3]
    assert (np.sqrt(T, dtype=f32) >= r).all() and (np.sqrt(np.nextafter(T, f32(0)), dtype=f32) < r).all()
    if len(leaves) <= 80:  # sphere numbers exist where the scene can be staged in LDS
        assert np.array_equal(leaves[sph, 6].view(np.int32), np.arange(sph.sum(), dtype=np.int32))
        # the fused kernel's record table: mirrored pairs are the neighbouring triangle leaves
        # whose edges are each other's exact negations (zero components +0 in both, the others
        # of magnitude in [2^-60, 2^60]) and whose normals agree; the second has no record
        records, seconds = sg.read_pairs()
        bits = leaves.view(np.uint32)

        def mirrored(a, b):
            if a == 0:
                return b == 0
            m = abs(float(np.array([a], np.uint32).view(f32)[0]))
            return b == a ^ 0x80000000 and np.isfinite(m) and 2.0 ** -60 <= m <= 2.0 ** 60

        want_records, want_seconds = [], []
        k = 0
        while k < len(leaves):
            rec = leaves[k].copy()
            if k + 1 < len(leaves) and not sph[k] and not sph[k + 1]:
                ok = all(mirrored(int(bits[k, c]), int(bits[k + 1, c])) for c in (4, 5, 6, 8, 9, 10))
                ok = ok and all(leaves[k, c] == leaves[k + 1, c] for c in (3, 7, 11))
                if ok:
                    want_seconds.append(leaves[k + 1, :12])
                    rec[12] = np.array([len(want_seconds)], np.int32).view(f32)[0]
                    k += 1
            want_records.append(rec)
            k += 1
        assert np.array_equal(records.view(np.uint32), np.array(want_records, f32).view(np.uint32))
        keep = [0, 1, 2, 4, 5, 6, 8, 9, 10]
        assert len(seconds) == len(want_seconds)
        if len(seconds):
            assert np.array_equal(seconds[:, keep].view(np.uint32),
                                  np.array(want_seconds, f32)[:, keep].view(np.uint32))
        # (a third condition -- no vertex coordinate of the second triangle is -0 -- is vacuous for
        # these scenes and has a test of its own: test_walls_with_negative_zero_coordinates)
        for ob in spec["objects"]:
            if ob[0] == "mesh":
                assert not (np.signbit(np.asarray(ob[2], f32)) & (np.asarray(ob[2], f32) == 0)).any()
        if which == "default":  # every wall of the reference's scene is a parallelogram split in two
            assert len(seconds) == (~sph).sum() // 2 > 0
    else:
        records, seconds = sg.read_pairs()
        assert len(records) == 0 and len(seconds) == 0


def test_walls_with_negative_zero_coordinates(rt):
    """The default scene with every zero vertex coordinate written as -0: a wall's two triangles are
    then tested each on its own (the shared-direction short cut gives a zero of the other sign where
    the reference's arithmetic has one, which a -0 coordinate could carry into a hit's position), and
    images stay equal to the checker's bit for bit."""
    spec = default_scene_spec()
    objs = []
    for ob in spec["objects"]:
        if ob[0] == "mesh":
            v = np.asarray(ob[2], f32).copy()
            v[v == 0] = -0.0
            ob = (ob[0], ob[1], v, ob[3])
        objs.append(ob)
    spec["objects"] = objs
    O = Oracle()
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    records, seconds = sg.read_pairs()
    walls_with_zero = sum(1 for ob in objs if ob[0] == "mesh" and (ob[2] == 0).any())
    assert walls_with_zero > 0 and len(seconds) < 5  # walls that touch a zero plane form no pair
    org, d = _rays.default_batch(O, so, n_interior=20000, prim_step=16)
    _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))
    img, stats = O.render(so, 64, 64, 16, seed=4, chunk_spp=4)
    for how in (dict(), dict(mode=1)):
        rgb, st = sg.render(64, 64, 16, seed=4, chunk_spp=4, **how)
        _assert_images_equal(rgb, img)
        assert st.rays == stats[1] and st.vertices == stats[3]


def test_scene_create_rejects_malformed_trees(rt):
    import ctypes as C
    import test_host_cpu as th
    lib = rt.hip_lib()
    Desc = th._desc_for(rt, None)
    lib.rtr_scene_create.argtypes = [C.POINTER(Desc), C.c_int, C.POINTER(C.c_void_p)]
    out = C.c_void_p()
    keep = []

    def arr(a, t):
        a = np.ascontiguousarray(a, t)
        keep.append(a)
        return a.ctypes.data

    def desc(left, right, prim, n_prims=0):
        d = Desc()
        d.n_nodes = len(left)
        d.root = 0
        z = [0.0] * len(left)
        for a in range(3):
            d.node_bmin[a] = arr(z, f32)
            d.node_bmax[a] = arr(z, f32)
            d.sph_c[a] = arr([0.0], f32)
            d.mat_emission[a] = arr([0.0], f32)
            d.mat_kd[a] = arr([0.0], f32)
            d.mat_ks[a] = arr([0.0], f32)
        d.node_left, d.node_right, d.node_prim = arr(left, np.int32), arr(right, np.int32), arr(prim, np.int32)
        d.n_prims = n_prims
        d.prim_kind, d.prim_index, d.prim_material = (arr([1] * max(n_prims, 1), np.int32),
                                                      arr([0] * max(n_prims, 1), np.int32),
                                                      arr([0] * max(n_prims, 1), np.int32))
        d.n_spheres = 1 if n_prims else 0
        d.sph_r, d.sph_area = arr([1.0], f32), arr([1.0], f32)
        d.n_materials = 1 if n_prims else 0
        d.mat_kind, d.mat_ior, d.mat_roughness = arr([0], np.int32), arr([1.0], f32), arr([1.0], f32)
        return d

    def rejected(d, word):
        assert lib.rtr_scene_create(C.byref(d), 0, C.byref(out)) == -1
        assert word in lib.rtr_last_error(), lib.rtr_last_error()
        assert not out.value

    rejected(desc([1, 0], [1, 0], [-1, -1]), b"twice")            # a child that points back at the root
    rejected(desc([-1, -1], [-1, -1], [-1, -1]), b"prim")          # leaves that name no primitive
    rejected(desc([1, -1, -1], [5, -1, -1], [-1, 0, 0], 1), b"out of range")
    rejected(desc([1, -1, -1], [-1, -1, -1], [-1, 0, 0], 1), b"one child")
    rejected(desc([1, -1, -1, -1, -1], [2, -1, -1, -1, -1], [-1, 0, 0, 0, 0], 1), b"unreachable")
    # a chain 70 levels deep: node 2i has children 2i+1 (a leaf) and 2i+2
    depth = 70
    left = [i + 1 if i % 2 == 0 and i < 2 * depth else -1 for i in range(2 * depth + 1)]
    right = [i + 2 if i % 2 == 0 and i < 2 * depth else -1 for i in range(2 * depth + 1)]
    prim = [-1 if l >= 0 else 0 for l in left]
    rejected(desc(left, right, prim, 1), b"deeper")
    # and the same chain 60 levels deep is accepted
    depth = 60
    right = [i + 2 if i % 2 == 0 and i < 2 * depth else -1 for i in range(2 * depth + 1)]
    left = [i + 1 if i % 2 == 0 and i < 2 * depth else -1 for i in range(2 * depth + 1)]
    prim = [-1 if l >= 0 else 0 for l in left]
    d = desc(left, right, prim, 1)
    assert lib.rtr_scene_create(C.byref(d), 0, C.byref(out)) == 0, lib.rtr_last_error()
    lib.rtr_scene_destroy(out)


def test_bench_multi_rank_code_path_on_one_gpu(rt, scenes, tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RCCL), rehearsed
    with one rank: process group, barrier, the gather of a device tensor, the column
    re-ordering and the reductions all run; the gathered image must equal a plain render."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + (os.getpid() % 200)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--width", "96", "--height", "64", "--spp", "8", "--no-cpu-baseline", "--big", "128x96x16",
                        "--rehearse-gather", "--dump-image", str(tmp_path / "img.npy")],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    # what a run on several GPUs needs to explain itself: per-rank render times, the gather's cost,
    # the tile width, and the large configuration rendered by the same ranks
    mg = line["multi_gpu"]
    assert len(mg["per_rank_render_ms"]) == 1 and mg["render_ms_max"] >= mg["render_ms_min"] > 0
    assert mg["imbalance_max_over_mean"] == 1.0 and mg["gather_ms"] >= 0 and mg["tile_cols"] == 8
    big = line["extra"][0]
    assert big["msamples_per_s"] > 0 and len(big["per_rank_render_ms"]) == 1 and "128x96 at 16 spp" in big["workload"]
    O, so, sg = scenes
    want, _ = sg.render(96, 64, 8, seed=0)
    got = np.load(tmp_path / "img.npy").reshape(-1, 3)
    _assert_images_equal(got, want)


def _tie_scene_spec(with_walls):
    """Spheres that coincide pairwise (same centre and radius, different materials), so that
    equal hit distances are the rule and the winner of a tie decides the picture; with walls
    (20 spheres, 59 nodes) or without (32 spheres: every bit of the per-ray sphere mask)."""
    spec = default_scene_spec()
    rng = np.random.default_rng(5)
    walls = [ob for ob in spec["objects"] if ob[0] == "mesh"] if with_walls else []
    objs = [("sphere", "light", (256, 256, -256), 20)]
    n_pairs = 8 if with_walls else 14
    mats = ["red", "spec", "micro", "green", "blue", "white"]
    for k in range(n_pairs):
        c = tuple(float(x) for x in (rng.uniform(60, 450, 3) * np.array([1, 1, -1])))
        r = float(rng.uniform(20, 60))
        objs.append(("sphere", mats[k % 6], c, r))
        objs.append(("sphere", mats[(k + 1) % 6], c, r))
    # a sphere that touches the back wall (tangent point hit at the wall's distance) and spares
    objs.append(("sphere", "white", (256, 256, -512 + 40), 40))
    objs.append(("sphere", "spec", (120, 400, -300), 25))
    objs.append(("sphere", "micro", (400, 120, -200), 25))
    spec["objects"] = objs[:1] + walls + objs[1:]
    return spec


@pytest.mark.parametrize("with_walls", [True, False])
def test_render_with_coinciding_spheres(rt, with_walls):
    """The fused kernel finishes its sphere tests out of leaf order (marked spheres, lane by
    lane); the reference's rule for equal distances -- the later leaf wins -- must still
    hold.  Images and counters in every loop variant against the checker, bit for bit."""
    O = Oracle()
    spec = _tie_scene_spec(with_walls)
    so = O.build_scene(spec)
    O.set_math(so, portable=True)
    sg = rt.HostScene.from_spec(spec)
    org, d = _rays.interior_rays(100000, seed=3)
    _assert_hits_equal(sg.intersect(org, d), so.intersect(org, d))
    want, stats = O.render(so, 64, 48, 12, seed=6, chunk_spp=4)
    for how in MODES:
        rgb, st = sg.render(64, 48, 12, seed=6, chunk_spp=4, **how)
        assert st.mode == how["mode"]
        _assert_images_equal(rgb, want)
        assert (st.rays, st.vertices) == (stats[1], stats[3])
    assert want.mean() > 0.001
