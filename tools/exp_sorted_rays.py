"""Development experiment: does ordering the rays (by origin cell and direction octant) speed
up the large-scene traversal?  Same rays, three orders."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import raytracingrenderer_amd as rt
import _rays
sg = rt.HostScene.from_spec(_rays.heightfield_spec(708)); sg.device_scene()
nr = 1 << 24
org, d = _rays.interior_rays(nr, seed=5)
dev = torch.device("cuda", 0)
lib = rt.hip_lib()
def bench(o_np, d_np, label):
    o_t = torch.from_numpy(o_np).to(dev); d_t = torch.from_numpy(d_np).to(dev)
    prim = torch.empty(nr, dtype=torch.int32, device=dev); t = torch.empty(nr, dtype=torch.float32, device=dev)
    def run():
        rc = lib.rtr_trace_closest_device(sg.device_scene(), C.c_void_p(o_t.data_ptr()), C.c_void_p(d_t.data_ptr()),
                                          C.c_size_t(nr), C.c_void_p(prim.data_ptr()), C.c_void_p(t.data_ptr()), None, None, None)
        assert rc == 0; torch.cuda.synchronize()
    run(); t0 = time.time(); run(); run(); dt = (time.time() - t0) / 2
    print("%-40s %.1f ms  %.1f Mrays/s" % (label, dt * 1e3, nr / dt / 1e6), flush=True)
bench(org, d, "as generated (incoherent)")
def key(bits):
    cell = np.clip(((org - org.min(0)) / (np.ptp(org, 0) + 1e-6) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    k = np.zeros(nr, np.int64)
    for b in range(bits):
        for a in range(3):
            k |= ((cell[:, a] >> b) & 1) << (3 * b + a)
    octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
    return (k << 3) | octant
for bits in (3, 5, 7):
    p = np.argsort(key(bits), kind="stable")
    bench(np.ascontiguousarray(org[p]), np.ascontiguousarray(d[p]), "sorted by %d-bit Morton cell + octant" % bits)
p = np.argsort((((d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)) << 40) | key(5), kind="stable")
bench(np.ascontiguousarray(org[p]), np.ascontiguousarray(d[p]), "sorted by octant, then 5-bit cell")
