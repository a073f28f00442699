"""Deterministic ray batches shared by the parity tests and the golden generator."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

f32 = np.float32


def _unit(v):
    v = np.asarray(v, f32)
    n = np.sqrt((v * v).sum(-1, keepdims=True, dtype=f32), dtype=f32)
    return (v / n).astype(f32)


def primary_rays(oracle, scene, W, H, step):
    org, d = [], []
    for i in range(0, W, step):
        for j in range(0, H, step):
            o1, d1 = oracle.primary_ray(scene, W, H, i, j)
            org.append(o1)
            d.append(d1)
    return np.array(org, f32), np.array(d, f32)


def secondary_rays(scene, org, d, seed, light_center=(256, 256, -256), light_radius=20):
    """Shadow-like and bounce-like rays leaving the first hits of (org, d)."""
    rng = np.random.default_rng(seed)
    prim, t, pos, nrm = scene.intersect(org, d)
    ok = (prim >= 0) & np.isfinite(t)
    pos = pos[ok]
    n = len(pos)
    # towards points on the light sphere (src/Sphere.cpp:41-48 style parametrisation)
    th = (2 * np.pi * rng.random(n)).astype(f32)
    ph = (np.pi * rng.random(n)).astype(f32)
    lp = np.stack([np.cos(ph), np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th)], -1).astype(f32)
    lp = (np.array(light_center, f32) + f32(light_radius) * lp).astype(f32)
    shadow = _unit(lp - pos)
    # angle-uniform bounce directions (src/Material.cpp:124-126 style)
    a = (2 * np.pi * rng.random(n)).astype(f32)
    b = (2 * np.pi * rng.random(n)).astype(f32)
    wo = np.stack([np.cos(a) * np.cos(b), np.sin(b), np.sin(a) * np.cos(b)], -1).astype(f32)
    return (np.concatenate([pos, pos]).astype(f32),
            np.concatenate([shadow, _unit(wo)]).astype(f32))


from raytracingrenderer_amd.scenes import interior_rays  # noqa: E402,F401


def adversarial_rays():
    """Axis-parallel, in-plane, corner, zero-component and degenerate rays."""
    org, d = [], []
    axes = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    starts = [(256, 256, 512), (256, 256, -256), (0, 0, 0), (512, 512, -512), (0, 256, -256),
              (256, 0, -256), (256, 512, -256), (256, 256, -512), (64, 64, -448),
              (64, 64, -384), (128, 64, -448), (256, 256, -236), (256, 256, -276),
              (256, 448, -256), (256, 480, -256), (100, 0, -100), (0, 100, -100),
              (512, 100, -100), (100, 512, -100), (100, 100, -512), (100, 100, 0)]
    for s in starts:
        for a in axes:
            org.append(s)
            d.append(a)
    # diagonals with one zero component, through edges and corners
    diag = [(1, 1, 0), (1, 0, 1), (0, 1, 1), (1, -1, 0), (-1, 0, 1), (0, -1, -1), (1, 1, 1),
            (-1, -1, -1), (1, 1, -1), (1, -1, -1), (-1, 1, -1)]
    for s in starts:
        for a in diag:
            org.append(s)
            d.append(a)
    # rays lying exactly in wall planes and along box edges
    for s, a in [((0, 0, -512), (1, 0, 0)), ((0, 0, -512), (0, 1, 0)), ((0, 0, -512), (0, 0, 1)),
                 ((0, 0, -512), (1, 1, 0)), ((0, 0, 0), (1, 1, -1)), ((512, 512, 0), (-1, -1, -1)),
                 ((10, 0, -10), (1, 0, -1)), ((0, 10, -10), (0, 1, -1)), ((10, 10, -512), (1, 2, 0)),
                 ((256, 256, 600), (0, 0, -1)), ((256, 256, 600), (0, 0, 1)),
                 ((-100, 256, -256), (1, 0, 0)), ((700, 256, -256), (-1, 0, 0)),
                 ((256, -100, -256), (0, 1, 0)), ((256, 700, -256), (0, -1, 0))]:
        org.append(s)
        d.append(a)
    # tangent / grazing rays on spheres, rays from inside spheres
    for s, a in [((0, 96, -256), (1, 0, 0)), ((0, 288, -256), (1, 0, 0)), ((64, 0, -256), (0, 1, 0)),
                 ((256, 256, -256), (1, 0, 0)), ((256, 256, -250), (0, 1, 0)),
                 ((64, 64, -448), (1, 1, 1)), ((64.5, 64, -448), (0, 0, 1)),
                 ((256, 236, -256), (0, -1, 0)), ((256, 236, -256), (0, 1, 0)),
                 ((256, 276, -256), (0, 1, 0)), ((256, 276, -256), (0, -1, 0))]:
        org.append(s)
        d.append(a)
    # degenerate directions
    for a in [(0, 0, 0), (1e-30, 0, 0), (0, 1e-30, 1e-30), (1e20, 1e20, 1e20), (1e-20, 1, 0),
              (np.inf, 0, 0), (np.nan, 1, 0)]:
        org.append((256, 256, 100))
        d.append(a)
    return np.array(org, f32), np.array(d, f32)


def default_batch(oracle, scene, n_interior=20000, prim_step=8):
    """The standard traversal batch: primary + secondary + interior + adversarial."""
    po, pd = primary_rays(oracle, scene, 784, 784, prim_step)
    so, sd = secondary_rays(scene, po, pd, seed=11)
    io, idr = interior_rays(n_interior, seed=12)
    ao, ad = adversarial_rays()
    org = np.concatenate([po, so, io, ao]).astype(f32)
    d = np.concatenate([pd, sd, idr, ad]).astype(f32)
    return np.ascontiguousarray(org), np.ascontiguousarray(d)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same_bits(a, b):
    """Bit equality, except that any NaN equals any NaN."""
    a = np.ascontiguousarray(a, f32)
    b = np.ascontiguousarray(b, f32)
    eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return eq


from raytracingrenderer_amd.scenes import heightfield_spec  # noqa: E402,F401


def connection_pairs(scene, n, seed):
    """Point pairs for Scene::checkConnectable: surface points found by tracing random
    interior rays with `scene` (anything with .intersect), paired up at random -- roughly
    half of them see each other -- plus pairs that must fail or are degenerate: a point
    and itself, a point behind a wall, points in mid-air."""
    rng = np.random.default_rng(seed)
    org, d = interior_rays(2 * n, seed + 1)
    prim, t, pos, nrm = scene.intersect(org, d)
    pts = pos[prim >= 0]
    # lift the points a little off their surface so that the connecting ray does not start
    # inside the 1e-5 slack of the triangle it lies on
    lift = nrm[prim >= 0] * f32(0.05)
    pts = (pts + lift).astype(f32)
    k = min(n, len(pts) // 2)
    a, b = pts[:k].copy(), pts[k:2 * k].copy()
    # the exact surface points as targets (what a light or camera vertex would be)
    b[::2] = pos[prim >= 0][k:2 * k][::2]
    extra_a = np.array([[256, 256, -100], [256, 256, -100], [100, 100, -100], [256, 400, -256],
                        [10, 10, -10]], f32)
    extra_b = np.array([[256, 256, -100], [256, 256, -700], [300, 310, -120], [256, 100, -256],
                        [500, 500, -500]], f32)
    return np.concatenate([a, extra_a]), np.concatenate([b, extra_b])
