"""CPU-side checks of the product's host logic (no compute on a GPU):

* the host library's BVH build + flattening equals the checker's (and therefore
  the reference's) tree bit for bit;
* librtr_hip.so loads and exports every symbol include/rtr.h declares;
* the C ABI's argument validation and error reporting.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _rays
from _oracle import Oracle, default_scene_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


@pytest.fixture(scope="module")
def rt():
    import raytracingrenderer_amd as rt
    return rt


def test_abi_exports_every_declared_symbol(rt):
    hdr = open(os.path.join(ROOT, "include", "rtr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rtr_[a-z_]+)\s*\(", hdr)))
    assert len(declared) >= 13
    lib = rt.hip_lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(rt.api.ABI_SYMBOLS) == declared


def test_host_flatten_matches_checker_default_scene(rt):
    O = Oracle()
    so = O.default_scene()
    sh = rt.default_scene()
    for a, b in zip(sh.flatten(), so.flatten()):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))
    assert sh.num_prims() == 19
    for o in range(14):
        assert sh.object_area(o).view(np.uint32) == so.object_area(o).view(np.uint32)
    kind, prim, area, begin, tris = sh.lights()
    assert list(kind) == [1] and list(prim) == [0] and area[0] == f32(5026.54833984375)


def test_host_flatten_matches_checker_random_mesh(rt):
    rng = np.random.default_rng(4)
    spec = default_scene_spec()
    n = 2000
    verts = (rng.uniform(0, 512, (3 * n, 3)) * np.array([1, 1, -1])).astype(f32)
    idx = rng.permutation(3 * n).astype(np.uint32)
    spec["objects"] = spec["objects"] + [("mesh", "light", verts, idx)]
    O = Oracle()
    so = O.build_scene(spec)
    sh = rt.HostScene.from_spec(spec)
    for a, b in zip(sh.flatten(), so.flatten()):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))
    kind, prim, area, begin, tris = sh.lights()
    assert list(kind) == [1, 0] and begin[-1] == n == len(tris)
    assert area[1].view(np.uint32) == so.object_area(14).view(np.uint32)
    # mesh-light triangles are listed in the leaf order of the mesh's own tree
    lo, hi, child, nprim = so.flatten()
    mesh_prims = set(range(19, 19 + n))
    leaf_order = [p for p in nprim if p in mesh_prims]
    assert list(tris) == leaf_order


def test_host_flatten_matches_checker_grid_mesh_with_tied_keys(rt):
    """A regular grid: whole columns of triangles share one box centre, so the tree depends
    on how the reference's std::sort orders equal keys.  45 000 triangles also take the
    host build through its concurrent subtree tasks, which must not change the result."""
    spec = _rays.heightfield_spec(150)
    so = Oracle().build_scene(spec)
    sh = rt.HostScene.from_spec(spec)
    fh, fo = sh.flatten(), so.flatten()
    assert len(fh[3]) == 2 * (19 + 2 * 150 * 150) - 1    # one leaf per primitive
    for a, b in zip(fh, fo):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))


def test_node_box_override_matches_checker(rt):
    """Scene::setNodeBound and the checker's orc_set_node_box address the same nodes."""
    O = Oracle()
    so = O.default_scene()
    sh = rt.default_scene()
    lo, hi, child, prim = sh.flatten()
    for k in (0, 3, 11, len(prim) - 1):
        nlo, nhi = lo[k] + f32(1.5), hi[k] - f32(0.25)
        sh.set_node_box(k, nlo, nhi)
        O.set_node_box(so, k, nlo, nhi)
    fh, fo = sh.flatten(), so.flatten()
    for a, b in zip(fh, fo):
        assert np.array_equal(_rays.bits(a), _rays.bits(b))
    assert not np.array_equal(fh[0], lo)
    with pytest.raises(rt.RtrError):
        sh.set_node_box(len(prim), lo[0], hi[0])


def _desc_for(rt, spec):
    """Build an rtr_scene_desc by hand (ctypes) from the host library's arrays."""
    class Desc(C.Structure):
        _fields_ = [("n_nodes", C.c_int32), ("root", C.c_int32),
                    ("node_bmin", C.c_void_p * 3), ("node_bmax", C.c_void_p * 3),
                    ("node_left", C.c_void_p), ("node_right", C.c_void_p), ("node_prim", C.c_void_p),
                    ("n_prims", C.c_int32), ("prim_kind", C.c_void_p), ("prim_index", C.c_void_p),
                    ("prim_material", C.c_void_p),
                    ("n_tris", C.c_int32), ("tri_p0", C.c_void_p * 3), ("tri_p1", C.c_void_p * 3),
                    ("tri_p2", C.c_void_p * 3), ("tri_n", C.c_void_p * 3), ("tri_area", C.c_void_p),
                    ("n_spheres", C.c_int32), ("sph_c", C.c_void_p * 3), ("sph_r", C.c_void_p),
                    ("sph_area", C.c_void_p),
                    ("n_materials", C.c_int32), ("mat_kind", C.c_void_p),
                    ("mat_emission", C.c_void_p * 3), ("mat_kd", C.c_void_p * 3),
                    ("mat_ks", C.c_void_p * 3), ("mat_ior", C.c_void_p), ("mat_roughness", C.c_void_p),
                    ("n_lights", C.c_int32), ("light_kind", C.c_void_p), ("light_prim", C.c_void_p),
                    ("light_area", C.c_void_p), ("light_tri_begin", C.c_void_p),
                    ("light_tris", C.c_void_p)]
    return Desc


def test_scene_create_rejects_malformed_descriptions(rt):
    lib = rt.hip_lib()
    Desc = _desc_for(rt, None)
    lib.rtr_scene_create.argtypes = [C.POINTER(Desc), C.c_int, C.POINTER(C.c_void_p)]
    out = C.c_void_p()
    assert lib.rtr_scene_create(None, 0, C.byref(out)) == -1
    assert b"null" in lib.rtr_last_error()
    d = Desc()
    d.n_nodes = -1
    assert lib.rtr_scene_create(C.byref(d), 0, C.byref(out)) == -1
    assert b"negative" in lib.rtr_last_error()
    d = Desc()
    d.n_nodes = (1 << 26) + 1
    assert lib.rtr_scene_create(C.byref(d), 0, C.byref(out)) == -1
    assert b"too many nodes" in lib.rtr_last_error()


def test_partition_columns_cover_image_once(rt):
    for W, R, tile in [(784, 8, 8), (2048, 8, 16), (100, 3, 8), (7, 4, 2), (5, 8, 8)]:
        seen = np.zeros(W, int)
        sizes = []
        for r in range(R):
            cols = rt.partition_columns(W, r, R, tile)
            assert (np.diff(cols) > 0).all()
            seen[cols] += 1
            sizes.append(len(cols))
        assert (seen == 1).all()
        if W >= R * tile * 4:
            assert max(sizes) - min(sizes) <= tile
    assert list(rt.partition_columns(10, 0, 1, 8)) == list(range(10))


def test_no_device_means_loud_failure(rt):
    """Without a GPU every compute entry point fails with an error, never a CPU result."""
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    sh = rt.default_scene()
    with pytest.raises(rt.RtrError):
        sh.intersect(np.zeros((4, 3), f32), np.ones((4, 3), f32))
    with pytest.raises(rt.RtrError):
        sh.render(8, 8, 1)


REFERENCE_MAIN = "/root/reference/src/main.cpp"


@pytest.mark.skipif(not os.path.exists(REFERENCE_MAIN), reason="reference checkout not present")
def test_reference_main_compiles_against_the_host_mirror(tmp_path):
    """The drop-in claim, checked: the reference's own scene set-up (the body of its main up
    to the render call, src/main.cpp:7-86, read from the checkout at test time, never kept
    here) compiles unchanged against rtr_host.hpp and links against the two libraries,
    followed by the path-tracing call its main has commented out (src/main.cpp:90)."""
    import subprocess
    lines = open(REFERENCE_MAIN, encoding="utf-8", errors="replace").read().splitlines()
    body = "\n".join(lines[6:86])  # lines 7..86: materials, meshes, spheres, camera, scene, renderer
    assert "scene.rebuildBVH();" in body and "Render renderer;" in body
    src = tmp_path / "main_from_reference.cpp"
    src.write_text('#include "rtr_host.hpp"\nint main() {\n' + body +
                   '\n  renderer.options.devices = {0};\n'
                   '  renderer.renderPathTracing(scene, 512, 512, 1, "test0");\n'
                   '  renderer.render(scene, 512, 512, 1, "test0", RenderType::PathTracing);\n'
                   '  return 0;\n}\n')
    pkg = os.path.join(ROOT, "raytracingrenderer_amd")
    r = subprocess.run(["g++", "-std=c++17", "-o", str(tmp_path / "main_from_reference"), str(src),
                        "-I" + os.path.join(pkg, "host"), "-L" + pkg, "-lrtr_host", "-lrtr_hip",
                        "-Wl,-rpath," + pkg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_fused_kernel_spill_gate():
    """The fused kernel's scalar-register spills stay under their ceiling (`make resources`
    recompiles the device code with the compiler's resource remarks; tools/check_resources.py
    holds the ceilings and why they exist)."""
    import shutil
    import subprocess
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "raytracingrenderer_amd", "csrc"), "resources"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "k_render_fusedILb1E" in r.stdout


def test_parallel_form_of_std_sort_matches_std_sort(tmp_path):
    """The tree builder's kernels rest on a reformulation of libstdc++'s std::sort (Hoare partitions by
    ranks, all ranges of a round at once, heap sort and median of three move for move as in
    raytracingrenderer_amd/csrc/rtr_heapsort.hpp -- the header the kernels include --, the final
    insertion sort as a stable sort).  tools/introsort_model.cpp runs that formulation on the CPU
    against std::sort itself on key sets with ties."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "introsort_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "introsort_model.cpp")])
    out = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok: 600 cases"), out.stdout + out.stderr
    heaps = int(out.stdout.split(",")[1].split()[0])
    assert heaps > 0  # some of the cases exhaust the depth budget and end in heap sorts
