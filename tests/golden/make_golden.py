#!/usr/bin/env python3
"""Generates the golden vectors in this directory from the REFERENCE itself.

Needs oracle/_ref/libref.so (the reference's own sources compiled by
oracle/Makefile), so it only runs where the reference checkout exists.  The
outputs are data -- inputs and the reference's outputs -- never source text:

  bvh_default.npz        spliced tree of the default scene (bounds, children, prim ids),
                         top-level leaf order, object areas
  rays_default.npz       ray batch (primary, secondary, interior, adversarial) and the
                         reference's (prim id, distance, position, normal)
  materials_default.npz  Material::eval / pdf tables, and Material::sample outputs together
                         with the mt19937 draws that produced them
  lights_default.npz     Scene::samplingLight outputs + draws (default scene and a
                         multi-light scene with an emissive mesh)
  paths_default.npz      Render::pathTracing radiance for (ray, seed) pairs + the draws
  tree_tied_keys.npz     BVH::selfBuildBVH of a 72 x 72 height field over a regular grid (10 368 triangles whose
                         box centres tie along whole rows): the leaf of every node in depth-first order (which
                         fixes the topology, tie order of the reference's std::sort included), the children,
                         and the boxes of every 16th node
  tonemap.npz            Render::saveFile: a buffer of hard values (zero, one, beyond both, denormals,
                         NaN, infinities, both sides of every 8-bit step) and the PPM the reference wrote
  ref_render_NxN_4096spp.npy        Render::renderPathTracing float buffer, N = 128 and 256
  ref_render_NxN_4096spp_noise.npy  RMSE between two independent such renders

Run:  python tests/golden/make_golden.py [--skip-render | --only-render-256]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _rays  # noqa: E402
from _oracle import Oracle, Ref, default_scene_spec  # noqa: E402

f32 = np.float32


def multi_light_spec():
    spec = default_scene_spec()
    plate = np.array([[224, 20, -288], [288, 20, -288], [224, 20, -224], [288, 20, -224]], f32)
    spec["objects"] = [("sphere", "light", (100, 300, -200), 15)] + spec["objects"] + \
                      [("mesh", "light", plate, [0, 1, 2, 3, 2, 1])]
    return spec


def connectable(sr):
    """Scene::checkConnectable (include/Scene.h:49-56) of the reference on point pairs found
    with the reference's own closest-hit query."""
    a, b = _rays.connection_pairs(sr, 8000, 7)
    got = sr.check_connectable(a, b)
    np.savez_compressed(os.path.join(HERE, "connect_default.npz"), a=a, b=b, connected=got)
    print("connectable pairs:", len(a), "connected:", int(got.sum()))


def photon_maps(R):
    """Render::createPhotonMap (src/Render.cpp:405-481) of the reference, 15 photons per call
    (below 16 the reference runs one thread, so its draws can be replayed), with the
    generator's draws after the same seeding, on the default scene and on the scene with a
    second sphere light and an emissive mesh."""
    out = {}
    for name, spec in (("default", default_scene_spec()), ("lights", multi_light_spec())):
        sr = R.build_scene(spec)
        seeds = np.arange(100, 160, dtype=np.uint32)
        draws, pos, wi, power = [], [], [], []
        for sd in seeds:
            p = R.create_photon_map(sr, 15, int(sd))
            assert len(p[0]) == 15
            draws.append(R.draws(int(sd), 320))
            pos.append(p[0]); wi.append(p[1]); power.append(p[2])
        out[name + "_seeds"] = seeds
        out[name + "_draws"] = np.array(draws, f32)
        out[name + "_pos"] = np.array(pos, f32)
        out[name + "_wi"] = np.array(wi, f32)
        out[name + "_power"] = np.array(power, f32)
    np.savez_compressed(os.path.join(HERE, "photons.npz"), **out)
    print("photon maps:", {k: v.shape for k, v in out.items()})


def tree_tied_keys(R):
    spec = _rays.heightfield_spec(72)
    spec["objects"] = spec["objects"][-1:]  # the mesh alone: the flattened tree is the mesh's own
    lo, hi, child, prim = R.build_scene(spec).flatten()
    np.savez_compressed(os.path.join(HERE, "tree_tied_keys.npz"), cells=72, prim=prim, child=child,
                        bmin16=lo[::16], bmax16=hi[::16])


def tonemap(R):
    """Render::saveFile (src/Render.cpp:3-21) of the reference on values chosen to be hard for
    `(unsigned char)(255 * pow(clamp(0, 1, v), 0.6f))`: around every step of the 8-bit output."""
    vals = [0.0, -0.0, 1.0, 1.5, -1.0, 1e-45, -1e-45, 1e-40, 1.1754944e-38, np.nan, -np.nan, np.inf, -np.inf,
            0.5, 0.25, 0.999999, 1.0000001, 3.4e38, -3.4e38]
    v = np.array(vals, f32)
    steps = ((np.arange(1, 256, dtype=np.float64) / 255.0) ** (1.0 / 0.6)).astype(f32)
    near = [np.nextafter(steps, f32(2)), np.nextafter(steps, f32(-1))]
    near += [np.nextafter(near[0], f32(2)), np.nextafter(near[1], f32(-1))]
    rng = np.random.default_rng(77)
    v = np.concatenate([v, steps] + near + [rng.random(3000).astype(f32) ** 3])
    v = v[: len(v) // 6 * 6]
    height, width = 6, len(v) // 18
    px = v[: height * width * 3].reshape(-1, 3)
    data = R.save_file(px, height, width, "/tmp/golden_tonemap")
    np.savez_compressed(os.path.join(HERE, "tonemap.npz"), rgb=px, height=height, width=width,
                        ppm=np.frombuffer(data, np.uint8))
    print("tonemap:", px.shape, "ppm bytes", len(data), "header", data[:16])


def converged_render(R, sr, size, keep_first=False):
    """Two independent 4096-spp renders of the reference: the first is the fixture, the RMSE
    between the two calibrates the tolerance of the test that uses it."""
    stem = "ref_render_%dx%d_4096spp" % (size, size)
    path = os.path.join(HERE, stem + ".npy")
    # The reference's renders are not reproducible (random_device seed, one generator raced
    # by 16 threads) and a render now and then holds a NaN pixel: the fixture must be all
    # finite, and the noise figure is taken over the pixels that are finite in both runs.
    if keep_first and os.path.exists(path):
        a = np.load(path)
    else:
        a = R.render(sr, size, size, 4096, "/tmp/golden_a")
        while not np.isfinite(a).all():
            a = R.render(sr, size, size, 4096, "/tmp/golden_a")
        np.save(path, a)
    b = R.render(sr, size, size, 4096, "/tmp/golden_b")
    ok = np.isfinite(b).all(1)
    rmse = float(np.sqrt(((a[ok] - b[ok]) ** 2).mean()))
    np.save(os.path.join(HERE, stem + "_noise.npy"), np.array(rmse, f32))
    lum = lambda x: float((x * np.array([0.2126, 0.7152, 0.0722], f32)).sum(1).mean())
    print("%s: rmse between runs %g (%d non-finite pixels left out), lum %g %g"
          % (stem, rmse, int((~ok).sum()), lum(a), lum(b[ok])))


def main():
    R = Ref()
    O = Oracle()  # only used to generate primary-ray INPUTS (camera set-up)
    sr = R.default_scene()
    so = O.default_scene()
    if "--only-render-256" in sys.argv:
        converged_render(R, sr, 256, keep_first="--keep-first" in sys.argv)
        return
    if "--only-tonemap" in sys.argv:
        tonemap(R)
        return
    if "--only-tree" in sys.argv:
        tree_tied_keys(R)
        return
    tree_tied_keys(R)
    if "--only-photons" in sys.argv:
        photon_maps(R)
        return
    tonemap(R)
    connectable(sr)
    if "--only-connectable" in sys.argv:
        return
    photon_maps(R)

    lo, hi, child, prim = sr.flatten()
    np.savez_compressed(os.path.join(HERE, "bvh_default.npz"), bmin=lo, bmax=hi, child=child,
                        prim=prim, top_leaf_order=sr.top_leaf_order(),
                        areas=np.array([sr.object_area(o) for o in range(14)], f32))

    org, d = _rays.default_batch(O, so, n_interior=6000, prim_step=16)
    p, t, pos, nrm = sr.intersect(org, d)
    np.savez_compressed(os.path.join(HERE, "rays_default.npz"), org=org, dir=d, prim=p, t=t,
                        pos=pos, nrm=nrm)

    rng = np.random.default_rng(100)
    n = 3000
    wi, wo, nn = [rng.normal(size=(n, 3)).astype(f32) for _ in range(3)]
    nu = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    wiu = wi / np.linalg.norm(wi, axis=1, keepdims=True)
    wo[: n // 4] = (2 * (nu * wiu).sum(1, keepdims=True) * nu - wiu)[: n // 4].astype(f32)
    out = dict(wi=wi, wo=wo, n=nn, n_unit=nu.astype(f32))
    ns = 1000
    out["sample_u"] = np.stack([R.draws(5000 + k, 2) for k in range(ns)])
    for m in ["white", "red", "spec", "micro", "light"]:
        out["eval_" + m] = sr.material_eval(m, wi, wo, nn)
        out["pdf_" + m] = sr.material_pdf(m, wi, wo, nn)
        out["sample_" + m] = R.material_sample(sr, m, wi[:ns], nu.astype(f32)[:ns], seed=5000)
    np.savez_compressed(os.path.join(HERE, "materials_default.npz"), **out)

    def light_table(scene, seeds, ndraw):
        rows = dict(u=[], pos=[], nrm=[], pdf=[], prim=[])
        for s in seeds:
            po, no, pd, pr = R.sample_light(scene, s)
            rows["u"].append(R.draws(s, ndraw))
            rows["pos"].append(po)
            rows["nrm"].append(no)
            rows["pdf"].append(pd)
            rows["prim"].append(pr)
        return {k: np.array(v) for k, v in rows.items()}
    a = light_table(sr, range(500), 8)
    b = light_table(R.build_scene(multi_light_spec()), range(500), 16)
    np.savez_compressed(os.path.join(HERE, "lights_default.npz"),
                        **{"single_" + k: v for k, v in a.items()},
                        **{"multi_" + k: v for k, v in b.items()})

    po, pd = _rays.primary_rays(O, so, 784, 784, 23)
    rgb, draws, keep_o, keep_d = [], [], [], []
    ND = 384
    for k in range(len(po)):
        seed = 9000 + k
        c = R.path_trace(sr, po[k], pd[k], seed)
        u = R.draws(seed, ND)
        # keep the pair only if the path provably consumed fewer than ND draws
        got, used = O.path_trace_u(so, po[k], pd[k], u)
        if used <= ND:
            rgb.append(c)
            draws.append(u)
            keep_o.append(po[k])
            keep_d.append(pd[k])
    np.savez_compressed(os.path.join(HERE, "paths_default.npz"), org=np.array(keep_o, f32),
                        dir=np.array(keep_d, f32), u=np.array(draws, f32), rgb=np.array(rgb, f32))
    print("paths kept:", len(rgb), "of", len(po))

    if "--skip-render" not in sys.argv:
        converged_render(R, sr, 128)
        converged_render(R, sr, 256)


if __name__ == "__main__":
    main()
