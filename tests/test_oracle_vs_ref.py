"""Pins the CPU restatement (oracle/pt_oracle.cpp) against the reference itself.

oracle/_ref/libref.so is the reference's own sources compiled by oracle/Makefile
(one translation unit, so its single mt19937 can be re-seeded and its draws
replayed).  Everything here is bit-exact unless a tolerance is written out.
Skipped where the reference build is absent; tests/test_golden.py then covers
the same ground through the committed vectors.
"""
import numpy as np
import pytest

import _rays
from _oracle import Oracle, Ref, default_scene_spec, have_ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libref.so not built")

f32 = np.float32


@pytest.fixture(scope="module")
def both():
    O, R = Oracle(), Ref()
    return O, R, O.default_scene(), R.default_scene()


def test_bvh_shape_and_bounds(both):
    O, R, so, sr = both
    fo, fr = so.flatten(), sr.flatten()
    assert len(fo[3]) == 37  # 27 top-level nodes + 5 meshes x 3 nodes - 5 shared roots
    for a, b in zip(fo, fr):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))
    assert np.array_equal(so.top_leaf_order(), sr.top_leaf_order())
    # leaf order recorded from the compiled reference: spec(64,64) Back spec(64,448) ...
    assert list(so.top_leaf_order()) == [10, 1, 11, 12, 13, 0, 2, 3, 4, 5, 9, 7, 8, 6]
    for o in range(14):
        assert so.object_area(o).view(np.uint32) == sr.object_area(o).view(np.uint32)


def test_generic_builder_equals_builtin_default(both):
    O, _, so, _ = both
    s2 = O.build_scene(default_scene_spec())
    for a, b in zip(s2.flatten(), so.flatten()):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))


def test_slab_line_test(both):
    O, R, _, _ = both
    rng = np.random.default_rng(5)
    n = 20000
    lo = rng.uniform(-10, 10, (n, 3)).astype(f32)
    hi = (lo + rng.uniform(0, 5, (n, 3))).astype(f32)
    hi[::7] = lo[::7]  # flat boxes
    org = rng.uniform(-12, 12, (n, 3)).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    d[::5, 0] = 0  # axis-parallel -> inf / NaN slabs
    d[::11, 1] = 0
    org[::13] = lo[::13]  # origin on a box plane -> 0/0
    from _oracle import _f
    for k in range(n):
        a = O.lib.orc_slab(_f(lo[k]), _f(hi[k]), _f(org[k]), _f(d[k]))
        b = R.lib.ref_slab(_f(lo[k]), _f(hi[k]), _f(org[k]), _f(d[k]))
        assert a == b, (k, lo[k], hi[k], org[k], d[k])


def test_closest_hit_bit_exact(both):
    O, R, so, sr = both
    org, d = _rays.default_batch(O, so)
    po, to, poso, no = so.intersect(org, d)
    pr, tr, posr, nr = sr.intersect(org, d)
    assert np.array_equal(po, pr)
    hit = po >= 0
    assert _rays.same_bits(to, tr)[hit].all()
    assert _rays.same_bits(poso, posr)[hit].all()
    assert _rays.same_bits(no, nr)[hit].all()
    assert hit.sum() > 0.5 * len(org) and (~hit).sum() > 100


def test_closest_hit_random_scene(both):
    """Random spheres + a random triangle soup: exercises deeper trees and ties."""
    O, R, _, _ = both
    rng = np.random.default_rng(21)
    spec = default_scene_spec()
    pts = rng.uniform(0, 512, (600, 3)).astype(f32) * np.array([1, 1, -1], f32)
    idx = rng.integers(0, 600, 900).astype(np.uint32)
    objs = [("mesh", "white", pts, idx)]
    for k in range(40):
        c = rng.uniform(0, 512, 3) * np.array([1, 1, -1])
        objs.append(("sphere", ["red", "spec", "micro", "light"][k % 4], tuple(c), float(rng.uniform(2, 40))))
    spec["objects"] = objs
    so, sr = O.build_scene(spec), R.build_scene(spec)
    for a, b in zip(so.flatten(), sr.flatten()):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))
    org, d = _rays.interior_rays(30000, seed=3)
    po, to, poso, no = so.intersect(org, d)
    pr, tr, posr, nr = sr.intersect(org, d)
    assert np.array_equal(po, pr)
    hit = po >= 0
    assert _rays.same_bits(to, tr)[hit].all()
    assert _rays.same_bits(poso, posr)[hit].all()
    assert _rays.same_bits(no, nr)[hit].all()


def test_grid_mesh_with_tied_sort_keys(both):
    """A regular height field: every column of triangles shares one box-centre key, so the
    tree the reference builds depends on how its std::sort orders equal keys."""
    O, R, _, _ = both
    spec = _rays.heightfield_spec(60)
    so, sr = O.build_scene(spec), R.build_scene(spec)
    fo, fr = so.flatten(), sr.flatten()
    assert len(fo[3]) == 2 * (19 + 2 * 60 * 60) - 1
    for a, b in zip(fo, fr):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))


def test_boxes_that_are_not_nested(both):
    """Node boxes overwritten so that they no longer contain their children's (the same in
    the reference's own BVHNode::bound): closest hits and whole paths still agree."""
    O, R, _, _ = both
    rng = np.random.default_rng(12)
    spec = default_scene_spec()
    pts = rng.uniform(0, 512, (300, 3)).astype(f32) * np.array([1, 1, -1], f32)
    spec["objects"] = spec["objects"] + [("mesh", "white", pts, rng.integers(0, 300, 600).astype(np.uint32))]
    so, sr = O.build_scene(spec), R.build_scene(spec)
    lo, hi, child, prim = so.flatten()
    for k in rng.choice(len(prim), len(prim) // 4, replace=False):
        ext = hi[k] - lo[k]
        nlo = (lo[k] + rng.uniform(-0.2, 0.6, 3) * ext).astype(f32)
        nhi = np.maximum(nlo, (hi[k] - rng.uniform(-0.2, 0.6, 3) * ext).astype(f32))
        O.set_node_box(so, k, nlo, nhi)
        R.set_node_box(sr, k, nlo, nhi)
    for a, b in zip(so.flatten(), sr.flatten()):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))
    org, d = _rays.interior_rays(40000, seed=8)
    po, to, poso, no = so.intersect(org, d)
    pr, tr, posr, nr = sr.intersect(org, d)
    assert np.array_equal(po, pr)
    hit = po >= 0
    assert _rays.same_bits(to, tr)[hit].all() and _rays.same_bits(poso, posr)[hit].all()
    nested = O.build_scene(spec).intersect(org, d)
    assert (nested[0] != po).mean() > 0.01
    for k in range(300):
        u = R.draws(1000 + k, 600)
        want = R.path_trace(sr, org[k], d[k], 1000 + k)
        got, used = O.path_trace_u(so, org[k], d[k], u)
        assert used < 600 and _rays.same_bits(got, want).all(), k


def test_check_connectable(both):
    """Scene::checkConnectable, include/Scene.h:49-56: same booleans on surface-point pairs
    (about 40 % connected), degenerate pairs included; also on a random scene."""
    O, R, so, sr = both
    a, b = _rays.connection_pairs(sr, 20000, 3)
    co, cr = so.check_connectable(a, b), sr.check_connectable(a, b)
    assert np.array_equal(co, cr)
    assert 0.2 < cr.mean() < 0.8
    rng = np.random.default_rng(5)
    spec = default_scene_spec()
    pts = rng.uniform(0, 512, (600, 3)).astype(f32) * np.array([1, 1, -1], f32)
    idx = rng.integers(0, 600, 900).astype(np.uint32)
    spec["objects"] = spec["objects"] + [("mesh", "white", pts, idx)]
    so2, sr2 = O.build_scene(spec), R.build_scene(spec)
    a, b = _rays.connection_pairs(sr2, 10000, 4)
    assert np.array_equal(so2.check_connectable(a, b), sr2.check_connectable(a, b))


def _dirs(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3)).astype(f32)
    return v


def test_material_eval_pdf(both):
    O, R, so, sr = both
    n = 20000
    wi, wo, nn = _dirs(n, 1), _dirs(n, 2), _dirs(n, 3)
    # include exact mirror pairs so the SPECULAR branch fires, and grazing cases
    nu = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    wiu = wi / np.linalg.norm(wi, axis=1, keepdims=True)
    wo[: n // 4] = (2 * (nu * wiu).sum(1, keepdims=True) * nu - wiu)[: n // 4].astype(f32)
    for m in ["white", "red", "spec", "micro", "light"]:
        a = so.material_eval(m, wi, wo, nn)
        b = sr.material_eval(m, wi, wo, nn)
        assert _rays.same_bits(a, b).all(), m
        assert _rays.same_bits(so.material_pdf(m, wi, wo, nn), sr.material_pdf(m, wi, wo, nn)).all()
    assert (so.material_eval("spec", wi, wo, nn)[: n // 4].sum(1) > 0).mean() > 0.4


def test_material_sample_replays_reference_draws(both):
    O, R, so, sr = both
    n = 3000
    wi, nn = _dirs(n, 4), _dirs(n, 5)
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(f32)
    for m in ["white", "spec", "micro"]:
        ref = R.material_sample(sr, m, wi, nn, seed=1000)
        u = np.stack([R.draws(1000 + k, 2) for k in range(n)])
        got = O.material_sample(so, m, wi, nn, u)
        assert _rays.same_bits(got, ref).all(), m


def test_sample_light_replays_reference_draws(both):
    O, R, so, sr = both
    for seed in range(300):
        pr, nr, pdr, primr = R.sample_light(sr, seed)
        u = R.draws(seed, 16)
        po, no, pdo, primo, used = O.sample_light(so, u)
        assert used == 3
        assert primo == primr == 0
        assert _rays.same_bits(po, pr).all() and _rays.same_bits(no, nr).all()
        assert f32(pdo).view(np.uint32) == f32(pdr).view(np.uint32)


def test_sample_light_multi_light_and_mesh_lights(both):
    """Two sphere lights + an emissive mesh: the pick loop's off-by-one and the
    mesh walk that lets the last leaf win (src/Scene.cpp:22-33, src/BVH.cpp:82-95)."""
    O, R, _, _ = both
    spec = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    spec["objects"] = [("sphere", "light", (100, 300, -200), 15)] + spec["objects"] + \
                      [("mesh", "light", plate, [0, 1, 2, 3, 2, 1])]
    so, sr = O.build_scene(spec), R.build_scene(spec)
    prims = set()
    for seed in range(400):
        pr, nr, pdr, primr = R.sample_light(sr, seed)
        u = R.draws(seed, 32)
        po, no, pdo, primo, used = O.sample_light(so, u)
        assert primo == primr
        assert _rays.same_bits(po, pr).all() and _rays.same_bits(no, nr).all()
        assert f32(pdo).view(np.uint32) == f32(pdr).view(np.uint32)
        prims.add(primo)
    assert len(prims) >= 1


def test_path_trace_replays_reference_draws(both):
    """Render::pathTracing, whole recursion, bit for bit on 4000 (ray, seed) pairs."""
    O, R, so, sr = both
    org, d = _rays.primary_rays(O, so, 784, 784, 13)
    lengths = []
    for k in range(len(org)):
        seed = 77 + k
        ref = R.path_trace(sr, org[k], d[k], seed)
        u = R.draws(seed, 4096)
        got, used = O.path_trace_u(so, org[k], d[k], u)
        assert used <= 4096
        assert _rays.same_bits(got, ref).all(), (k, got, ref, used)
        lengths.append(used)
    assert max(lengths) > 12  # some paths recurse


def test_path_trace_variant_scene(both):
    """Same replay on the specular/microfacet-heavy variant with two lights."""
    O, R, _, _ = both
    spec = default_scene_spec()
    objs = []
    for ob in spec["objects"]:
        if ob[0] == "sphere" and ob[1] in ("red", "blue"):
            ob = ("sphere", "spec", ob[2], ob[3])
        if ob[0] == "sphere" and ob[1] == "green":
            ob = ("sphere", "micro", ob[2], ob[3])
        objs.append(ob)
    objs.append(("sphere", "light", (100, 400, -100), 10))
    spec["objects"] = objs
    so, sr = O.build_scene(spec), R.build_scene(spec)
    org, d = _rays.primary_rays(O, so, 784, 784, 29)
    for k in range(len(org)):
        ref = R.path_trace(sr, org[k], d[k], 5 + k)
        got, used = O.path_trace_u(so, org[k], d[k], R.draws(5 + k, 4096))
        assert _rays.same_bits(got, ref).all(), (k, got, ref)


def test_photon_map_replays_reference_draws(both):
    """Render::createPhotonMap with 15 photons (below 16 the reference runs one thread) on
    the default scene, on its specular / microfacet variant and with three lights: the
    restatement fed with the generator's draws gives the same photons bit for bit."""
    O, R, so, sr = both
    variant = default_scene_spec()
    objs = []
    for ob in variant["objects"]:
        if ob[0] == "sphere" and ob[1] in ("red", "blue"):
            ob = ("sphere", "spec", ob[2], ob[3])
        if ob[0] == "sphere" and ob[1] == "green":
            ob = ("sphere", "micro", ob[2], ob[3])
        objs.append(ob)
    variant["objects"] = objs
    lights = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    lights["objects"] = [("sphere", "light", (100, 300, -200), 15)] + lights["objects"] + \
                        [("mesh", "light", plate, [0, 1, 2, 3, 2, 1])]
    cases = [(so, sr, 250)] + [(O.build_scene(sp), R.build_scene(sp), 120) for sp in (variant, lights)]
    for a, b, n_seeds in cases:
        total = 0
        for seed in range(n_seeds):
            want = R.create_photon_map(b, 15, seed)
            got = O.create_photon_map_u(a, R.draws(seed, 2000), 15)
            assert got[3] < 2000
            for x, y in zip(got[:3], want):
                assert x.shape == y.shape and _rays.same_bits(x, y).all(), seed
            total += len(want[0])
        assert total == 15 * n_seeds
    # fewer photons than one path yields, and none at all
    for n in (1, 2):
        want = R.create_photon_map(sr, n, 7)
        got = O.create_photon_map_u(so, R.draws(7, 500), n)
        assert len(want[0]) == n and _rays.same_bits(got[0], want[0]).all()


def test_render_deterministic_pixels_match_reference(both, tmp_path):
    """Pixels whose primary ray hits the light (or nothing) do not depend on the
    random stream: they pin the camera set-up and the pixel->buffer indexing."""
    O, R, so, sr = both
    W, H, spp = 96, 80, 3
    ref = R.render(sr, W, H, spp, str(tmp_path / "ref"))
    got, stats = O.render(so, W, H, spp, seed=1, threads=2, recursive=True)
    org, d = O.primary_rays(so, W, H)
    prim = so.intersect(org, d)[0]
    fixed = (prim == 0) | (prim < 0)
    assert fixed.sum() > 20
    assert _rays.same_bits(got[fixed], ref[fixed]).all()
    assert stats[0] == W * H * spp
    # and the rest agrees statistically (different random streams): mean within 15 %
    m_ref, m_got = ref[~fixed].mean(0), got[~fixed].mean(0)
    assert np.allclose(m_ref, m_got, rtol=0.15), (m_ref, m_got)
    # the PPM the reference wrote on the way (src/Render.cpp:686 -> :3-21): header with the height
    # first, then the buffer in its own order through the tone map
    data = (tmp_path / "ref.ppm").read_bytes()
    header = b"P6\n%d %d\n255\n" % (H, W)
    assert data.startswith(header)
    body = np.frombuffer(data[len(header):], np.uint8).reshape(-1, 3)
    want = np.array([[O.lib.orc_tonemap(float(v)) for v in px] for px in ref], np.uint8)
    assert np.array_equal(body, want)


def test_fewer_than_sixteen_rows(both, tmp_path):
    """src/Render.cpp:669-684: with height < 16 the reference's one thread renders row 0 only
    (step = height / 16 = 0); rows 1... stay zero.  The checker renders every row -- the host
    mirror applies the reference's rule on top (tests/test_gpu_parity.py) -- so row 0 is compared
    on the pixels that do not depend on the random stream, and the rest must be zero."""
    O, R, so, sr = both
    W, H, spp = 64, 8, 2
    ref = R.render(sr, W, H, spp, str(tmp_path / "ref8")).reshape(W, H, 3)
    assert not ref[:, 1:].any()
    assert ref[:, 0].any()
    got = O.render(so, W, H, spp, seed=3, threads=1, recursive=True)[0].reshape(W, H, 3)
    org, d = O.primary_rays(so, W, H)
    prim = so.intersect(org, d)[0].reshape(W, H)
    fixed = (prim[:, 0] == 0) | (prim[:, 0] < 0)
    assert _rays.same_bits(got[:, 0][fixed], ref[:, 0][fixed]).all()
