"""The CPU checker against the committed golden vectors (tests/golden/*.npz), which
were produced by the reference itself (tests/golden/make_golden.py drives the
reference's own sources compiled by oracle/Makefile).  Runs without the reference
checkout and without a GPU.  Bit-exact; NaN equals NaN.

The ``gpu``-marked tests at the bottom put the HIP path against the same vectors.
"""
import os

import numpy as np
import pytest

import _rays
from _oracle import Oracle, default_scene_spec

G = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


def _load(name):
    path = os.path.join(G, name)
    if not os.path.exists(path):
        pytest.fail("golden vector %s is missing (run tests/golden/make_golden.py where the "
                    "reference checkout exists)" % name)
    return np.load(path)


def _multi_light_spec():
    spec = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    spec["objects"] = [("sphere", "light", (100, 300, -200), 15)] + spec["objects"] + \
                      [("mesh", "light", plate, [0, 1, 2, 3, 2, 1])]
    return spec


@pytest.fixture(scope="module")
def osc():
    O = Oracle()
    return O, O.default_scene()


def test_bvh_golden(osc):
    O, so = osc
    g = _load("bvh_default.npz")
    lo, hi, child, prim = so.flatten()
    assert np.array_equal(_rays.bits(lo), _rays.bits(g["bmin"]))
    assert np.array_equal(_rays.bits(hi), _rays.bits(g["bmax"]))
    assert np.array_equal(child, g["child"]) and np.array_equal(prim, g["prim"])
    assert np.array_equal(so.top_leaf_order(), g["top_leaf_order"])
    areas = np.array([so.object_area(o) for o in range(14)], f32)
    assert np.array_equal(_rays.bits(areas), _rays.bits(g["areas"]))


def _check_tree_against_reference(flat):
    """flat = (bmin, bmax, child, prim) of the mesh-only 72 x 72 grid scene against what the
    reference's BVH::selfBuildBVH built (tests/golden/tree_tied_keys.npz)."""
    g = _load("tree_tied_keys.npz")
    lo, hi, child, prim = flat
    assert np.array_equal(prim, g["prim"]) and np.array_equal(child, g["child"])
    assert np.array_equal(_rays.bits(lo[::16]), _rays.bits(g["bmin16"]))
    assert np.array_equal(_rays.bits(hi[::16]), _rays.bits(g["bmax16"]))


def _tied_grid_spec():
    spec = _rays.heightfield_spec(72)
    spec["objects"] = spec["objects"][-1:]
    return spec


def test_tree_with_tied_keys_golden_checker():
    """Whole rows of box centres tie: the tree depends on the order std::sort leaves equal keys in."""
    _check_tree_against_reference(Oracle().build_scene(_tied_grid_spec()).flatten())


def test_tree_with_tied_keys_golden_host_builder(monkeypatch):
    import raytracingrenderer_amd as rt
    monkeypatch.setenv("RTR_TREE_BUILDER", "host")
    _check_tree_against_reference(rt.HostScene.from_spec(_tied_grid_spec()).flatten())


@pytest.mark.gpu
def test_tree_with_tied_keys_golden_device_builder(monkeypatch):
    """rtr_build_bvh directly against the reference's tree: 10 368 triangles take the kernels over
    the whole array (ranges above 2048 objects) and the subtrees in LDS."""
    import raytracingrenderer_amd as rt
    monkeypatch.setenv("RTR_TREE_BUILDER", "device")
    _check_tree_against_reference(rt.HostScene.from_spec(_tied_grid_spec()).flatten())


def _check_hits(got, g):
    p, t, pos, nrm = got
    assert np.array_equal(p, g["prim"])
    hit = g["prim"] >= 0
    assert _rays.same_bits(t, g["t"])[hit].all()
    assert _rays.same_bits(pos, g["pos"])[hit].all()
    assert _rays.same_bits(nrm, g["nrm"])[hit].all()
    assert hit.sum() > 5000 and (~hit).sum() > 100


def test_closest_hit_golden(osc):
    O, so = osc
    g = _load("rays_default.npz")
    _check_hits(so.intersect(g["org"], g["dir"]), g)


def test_materials_golden(osc):
    O, so = osc
    g = _load("materials_default.npz")
    ns = len(g["sample_u"])
    for m in ["white", "red", "spec", "micro", "light"]:
        assert _rays.same_bits(so.material_eval(m, g["wi"], g["wo"], g["n"]), g["eval_" + m]).all(), m
        assert _rays.same_bits(so.material_pdf(m, g["wi"], g["wo"], g["n"]), g["pdf_" + m]).all(), m
        got = O.material_sample(so, m, g["wi"][:ns], g["n_unit"][:ns], g["sample_u"])
        assert _rays.same_bits(got, g["sample_" + m]).all(), m


def test_light_sampling_golden(osc):
    O, so = osc
    g = _load("lights_default.npz")
    multi = O.build_scene(_multi_light_spec())
    for tag, scene in (("single_", so), ("multi_", multi)):
        for k in range(len(g[tag + "u"])):
            pos, nrm, pdf, prim, used = O.sample_light(scene, g[tag + "u"][k])
            assert prim == g[tag + "prim"][k]
            assert _rays.same_bits(pos, g[tag + "pos"][k]).all()
            assert _rays.same_bits(nrm, g[tag + "nrm"][k]).all()
            assert f32(pdf).view(np.uint32) == g[tag + "pdf"][k].view(np.uint32)
    assert len(set(g["multi_prim"].tolist())) >= 2


def test_path_tracing_golden(osc):
    """Render::pathTracing replayed with the reference's own mt19937 draws."""
    O, so = osc
    g = _load("paths_default.npz")
    assert len(g["org"]) > 500
    for k in range(len(g["org"])):
        got, used = O.path_trace_u(so, g["org"][k], g["dir"][k], g["u"][k])
        assert used <= g["u"].shape[1]
        assert _rays.same_bits(got, g["rgb"][k]).all(), (k, got, g["rgb"][k])


def test_check_connectable_golden(osc):
    O, so = osc
    g = _load("connect_default.npz")
    assert np.array_equal(so.check_connectable(g["a"], g["b"]), g["connected"])
    assert 2000 < g["connected"].sum() < len(g["a"]) - 2000


def test_photon_map_golden(osc):
    """Render::createPhotonMap of the reference (15 photons per call, one thread) against
    the restatement fed with the same generator draws: positions, directions, powers."""
    O, so = osc
    g = _load("photons.npz")
    scenes = {"default": so, "lights": O.build_scene(_multi_light_spec())}
    for name, sc in scenes.items():
        n_draws = []
        for k in range(len(g[name + "_seeds"])):
            pos, wi, power, used = O.create_photon_map_u(sc, g[name + "_draws"][k], 15)
            assert used <= g[name + "_draws"].shape[1]
            n_draws.append(used)
            assert _rays.same_bits(pos, g[name + "_pos"][k]).all(), (name, k)
            assert _rays.same_bits(wi, g[name + "_wi"][k]).all(), (name, k)
            assert _rays.same_bits(power, g[name + "_power"][k]).all(), (name, k)
        assert max(n_draws) > min(n_draws)


def test_photon_map_is_the_concatenation_of_its_paths(osc):
    """The counter-stream form: the map is the first n photons of paths 0, 1, 2, ... (the
    last path cut where the map is full), and a path's draws do not depend on the others."""
    O, so = osc
    n = 700
    pos, wi, power, used = O.create_photon_map(so, n, seed=5)
    assert len(pos) == n
    cat = [[], [], []]
    for path in range(used):
        ph = O.photon_path(so, 5, path)
        for a, b in zip(cat, ph):
            a.append(b)
    cat = [np.concatenate(a)[:n] for a in cat]
    for a, b in zip((pos, wi, power), cat):
        assert _rays.same_bits(a, b).all()
    assert sum(len(O.photon_path(so, 5, p)[0]) for p in range(used - 1)) < n
    # photons lie on surfaces of the box, arrive from outside the surface, carry positive power
    assert (power >= 0).all() and power.max() > 1
    assert np.allclose(np.linalg.norm(wi, axis=1), 1, atol=1e-5)
    other = O.create_photon_map(so, n, seed=6)[0]
    assert not np.array_equal(other, pos)


def test_tone_map_golden(osc, tmp_path):
    """Render::saveFile, src/Render.cpp:3-21: the checker's tone map and the host library's PPM
    writer against a PPM the reference itself wrote from a buffer of hard values (both sides of
    every 8-bit step, values outside [0, 1], denormals, NaNs, infinities)."""
    O, _ = osc
    g = _load("tonemap.npz")
    rgb, height, width = g["rgb"], int(g["height"]), int(g["width"])
    ppm = g["ppm"].tobytes()
    header = b"P6\n%d %d\n255\n" % (height, width)  # the reference prints the height first
    assert ppm.startswith(header) and len(ppm) == len(header) + 3 * height * width
    body = np.frombuffer(ppm[len(header):], np.uint8).reshape(-1, 3)
    want = np.array([[O.lib.orc_tonemap(float(v)) for v in px] for px in rgb], np.uint8)
    assert np.array_equal(want, body)
    import raytracingrenderer_amd as rt
    rt.save_ppm(rgb, height, width, tmp_path / "host")
    assert (tmp_path / "host.ppm").read_bytes() == ppm


def test_philox_known_answers(osc):
    """Philox4x32-10 known-answer vectors of the Random123 distribution."""
    O, _ = osc
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in O.philox(ctr, key)) == want


def test_forward_and_recursive_estimators_agree(osc):
    """The forward-accumulation form the kernels use equals the reference's recursion
    up to fp32 rounding (same draws, same rays)."""
    O, so = osc
    a, sa = O.render(so, 48, 40, 6, seed=3, threads=2, recursive=True)
    b, sb = O.render(so, 48, 40, 6, seed=3, threads=2, recursive=False)
    assert sa == sb  # identical sample / ray / vertex counts
    assert np.allclose(a, b, rtol=2e-5, atol=1e-6)


def test_portable_trig_close_to_libm(osc):
    """Fixed-sequence sin/cos vs libm: directions differ in the last ulps.  Individual
    samples may change (the reference's visibility test compares positions against a
    1e-3 threshold at coordinates of a few hundred, so it is sensitive to the last
    bits), the image statistics may not."""
    O, so = osc
    a, _ = O.render(so, 48, 48, 64, seed=3)
    s2 = O.default_scene()
    O.set_math(s2, portable=True)
    b, _ = O.render(s2, 48, 48, 64, seed=3)
    assert abs(a.mean() - b.mean()) / a.mean() < 0.02
    assert np.isclose(a, b, rtol=1e-4, atol=1e-6).all(1).mean() > 0.2


def test_converged_checker_image_matches_reference(osc):
    """Statistical tier for the checker's own renderer (counter-based stream instead of
    mt19937): 128x128 at 256 spp against the reference's 4096-spp render."""
    O, so = osc
    ref = _load("ref_render_128x128_4096spp.npy")
    got, _ = O.render(so, 128, 128, 256, seed=1)
    lum = lambda a: (a * np.array([0.2126, 0.7152, 0.0722], f32)).sum(1)
    rel = abs(lum(got).mean() - lum(ref).mean()) / lum(ref).mean()
    assert rel < 0.01, rel


# ---------------- the HIP path against the same vectors ----------------
@pytest.fixture(scope="module")
def gsc():
    import raytracingrenderer_amd as rt
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    return rt, rt.default_scene()


@pytest.mark.gpu
def test_gpu_closest_hit_golden(gsc):
    rt, sg = gsc
    g = _load("rays_default.npz")
    _check_hits(sg.intersect(g["org"], g["dir"]), g)


@pytest.mark.gpu
def test_gpu_material_eval_golden(gsc):
    rt, sg = gsc
    g = _load("materials_default.npz")
    for m in ["white", "red", "spec", "micro", "light"]:
        f, pdf = sg.material_eval(m, g["wi"], g["wo"], g["n"])
        assert _rays.same_bits(f, g["eval_" + m]).all(), m
        assert _rays.same_bits(pdf, g["pdf_" + m]).all(), m
    # mirror direction is deterministic: also bit-exact against the reference
    ns = len(g["sample_u"])
    got = sg.material_sample("spec", g["wi"][:ns], g["n_unit"][:ns], g["sample_u"])
    assert _rays.same_bits(got, g["sample_spec"]).all()
    # diffuse / microfacet directions go through sin/cos: the kernels use a fixed fp32
    # sequence, the reference its libm.  Tolerance: 4 ulp of 1.0 per component.
    for m in ["white", "micro"]:
        got = sg.material_sample(m, g["wi"][:ns], g["n_unit"][:ns], g["sample_u"])
        err = np.abs(got - g["sample_" + m])
        flipped = (np.sign(got) != np.sign(g["sample_" + m])).all(1)  # dot(wo,n) ~ 0
        assert (err[~flipped] <= 4 * 1.2e-7).all(), (m, err.max())
        assert flipped.mean() < 0.002


@pytest.mark.gpu
def test_gpu_check_connectable_golden(gsc):
    rt, sg = gsc
    g = _load("connect_default.npz")
    assert np.array_equal(sg.check_connectable(g["a"], g["b"]), g["connected"])
    assert np.array_equal(sg.check_connectable(g["a"][:500], g["b"][:500], via_host_api=True),
                          g["connected"][:500])


@pytest.mark.gpu
def test_gpu_photon_map(gsc):
    """rtr_create_photon_map against the restatement with the same counter streams, bit for
    bit: default scene at several sizes (below and above one batch of paths), through the
    C++ mirror, and the scene with three lights, one of them an emissive mesh."""
    rt, sg = gsc
    O = Oracle()
    so = O.default_scene()
    O.set_math(so, portable=True)
    for n, seed in ((1, 0), (15, 1), (1000, 2), (60000, 3)):
        want = O.create_photon_map(so, n, seed)
        got = sg.create_photon_map(n, seed)
        assert got[3] == want[3], (n, got[3], want[3])
        for a, b in zip(got[:3], want[:3]):
            assert a.shape == b.shape and _rays.same_bits(a, b).all(), n
    got = sg.create_photon_map(500, 2, via_host_api=True)
    want = O.create_photon_map(so, 500, 2)
    for a, b in zip(got[:3], want[:3]):
        assert a.shape == b.shape and _rays.same_bits(a, b).all()
    spec = _multi_light_spec()
    so2 = O.build_scene(spec)
    O.set_math(so2, portable=True)
    sg2 = rt.HostScene.from_spec(spec)
    want = O.create_photon_map(so2, 20000, 4)
    got = sg2.create_photon_map(20000, 4)
    assert got[3] == want[3]
    for a, b in zip(got[:3], want[:3]):
        assert a.shape == b.shape and _rays.same_bits(a, b).all()
    assert len(sg.create_photon_map(0)[0]) == 0


@pytest.mark.gpu
def test_gpu_photon_map_when_the_staging_room_runs_out(gsc, tmp_path):
    """The emission pass stages every wave's photons in room of a fixed size and redoes a batch
    the slow way (count, scan, run again) when a wave runs out of it.  With room for one photon
    per path (RTR_PHOTON_WAVE_ROOM=64, read when the library first needs it: hence a process of
    its own) every batch takes that way; the map must be the same."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import raytracingrenderer_amd as rt; "
            "sg = rt.default_scene(); p = sg.create_photon_map(60000, 3); "
            "np.savez(%r, pos=p[0], wi=p[1], power=p[2], used=p[3])"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "ph.npz")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RTR_PHOTON_WAVE_ROOM="64"),
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(tmp_path / "ph.npz")
    O = Oracle()
    so = O.default_scene()
    O.set_math(so, portable=True)
    want = O.create_photon_map(so, 60000, 3)
    assert int(got["used"]) == want[3]
    for k, b in zip(("pos", "wi", "power"), want[:3]):
        assert got[k].shape == b.shape and _rays.same_bits(got[k], b).all(), k


@pytest.mark.gpu
@pytest.mark.parametrize("size", [128, 256])
def test_gpu_converged_image_matches_reference(gsc, size):
    """>=4096 spp image tier: linear float buffers, RMSE <= 1.5 x the RMSE between two
    independent reference renders at the same spp, mean luminance within 0.5 %."""
    rt, sg = gsc
    ref = _load("ref_render_%dx%d_4096spp.npy" % (size, size))
    noise = float(_load("ref_render_%dx%d_4096spp_noise.npy" % (size, size)))
    assert np.isfinite(ref).all() and noise > 0
    rgb, st = sg.render(size, size, 4096, seed=1)
    lum = lambda a: (a * np.array([0.2126, 0.7152, 0.0722], f32)).sum(1)
    rel = abs(lum(rgb).mean() - lum(ref).mean()) / lum(ref).mean()
    rmse = float(np.sqrt(((rgb - ref) ** 2).mean()))
    assert rel < 0.005, rel
    assert rmse < 1.5 * noise, (rmse, noise)


def test_division_free_quotient_is_exact(tmp_path):
    """The traversal kernels rebuild a/d from RN(1/d) with one fma correction instead of
    dividing (rtr_trace.hpp: div_corrected).  oracle/check_division.c compares that sequence
    with the IEEE quotient over the kernels' operand ranges; any mismatch fails."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_division")
    flags = ["-O2", "-ffp-contract=off"]
    try:
        if " fma " in open("/proc/cpuinfo").read():
            flags.append("-mfma")  # hardware fma: same results as glibc's fmaf, much faster
    except OSError:
        pass
    subprocess.check_call(["gcc", *flags, "-o", exe, os.path.join(root, "oracle", "check_division.c"), "-lm"])
    n = "20000000" if "-mfma" in flags else "1000000"
    r = subprocess.run([exe, n, "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 mismatches" in r.stdout
