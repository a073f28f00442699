"""rtr_build_bvh on 2^23 random boxes: structural invariants of the result (every object in one leaf,
depth-first numbering, every inner box the union of its children's, leaves carrying their object's box,
the root's order sorted) -- a check of the index arithmetic at a size the comparison with the host
builder does not reach.  8.4 M objects build in 0.45 s including the copies."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import raytracingrenderer_amd as rt
f32 = np.float32
n = 1 << 23
rng = np.random.default_rng(1)
lo = rng.uniform(-1000, 1000, (n, 3)).astype(f32)
hi = lo + rng.uniform(0, 2, (n, 3)).astype(f32)
t0 = time.time()
nlo, nhi, left, right, obj, order = rt.build_bvh(lo, hi)
print("built %d objects in %.2f s" % (n, time.time() - t0), flush=True)
N = 2 * n - 1
leaves = left < 0
assert leaves.sum() == n
cnt = np.bincount(obj[leaves], minlength=n)
assert (cnt == 1).all()
inner = np.flatnonzero(~leaves)
assert (left[inner] == inner + 1).all()
# children boxes inside parents, parent box = union
l, r = left[inner], right[inner]
assert np.array_equal(nlo[inner], np.minimum(nlo[l], nlo[r])) and np.array_equal(nhi[inner], np.maximum(nhi[l], nhi[r]))
assert np.array_equal(nlo[np.flatnonzero(leaves)], lo[obj[leaves]])
# subtree sizes: right child index = i + 2 * floor(s / 2) with s the leaves below
print("structure ok", flush=True)
ext = nhi[0] - nlo[0]
axis = int(np.argmax(ext))
key = lo[:, axis] * f32(0.5) + hi[:, axis] * f32(0.5)
assert (np.diff(key[order]) >= 0).all()
print("root order sorted: ok")
