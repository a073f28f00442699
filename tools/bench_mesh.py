"""Development helper: closest-hit throughput of the general traversal on the
1 M-triangle height-field scene (rays resident in HBM, rtr_trace_closest_device)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import raytracingrenderer_amd as rt
import _rays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
t0 = time.time()
sg = rt.HostScene.from_spec(_rays.heightfield_spec(n))
sg.device_scene()
print("scene build + upload %.1f s, %d triangles" % (time.time() - t0, 2 * n * n), flush=True)
org, d = _rays.interior_rays(nr, seed=5)
dev = torch.device("cuda", 0)
o_t = torch.from_numpy(org).to(dev); d_t = torch.from_numpy(d).to(dev)
prim = torch.empty(nr, dtype=torch.int32, device=dev); t = torch.empty(nr, dtype=torch.float32, device=dev)
lib = rt.hip_lib()
def run():
    rc = lib.rtr_trace_closest_device(sg.device_scene(), C.c_void_p(o_t.data_ptr()), C.c_void_p(d_t.data_ptr()),
                                      C.c_size_t(nr), C.c_void_p(prim.data_ptr()), C.c_void_p(t.data_ptr()), None, None, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
run()
t0 = time.time(); run(); run(); dt = (time.time() - t0) / 2
print("%d rays in %.1f ms: %.1f Mrays/s; hit fraction %.3f" % (nr, dt * 1e3, nr / dt / 1e6, float((prim >= 0).float().mean())))
if "--json" in sys.argv:
    # one line in the shape of bench.py's, for the large-scene traversal.  Algorithmic bytes:
    # 32 B per node visited + 48 B per triangle / 16 B per sphere tested; the averages per ray
    # (78.7 nodes, 7.5 primitives, nearly all triangles) were counted on the device with a
    # -DRTR_STREAM_STATS build (tools/stream_stats.py) for exactly these rays.
    import json
    bytes_per_ray = 78.7 * 32 + 7.5 * 48
    print(json.dumps({"metric": "closest-hit queries per second, 1 M-triangle scene", "value": nr / dt / 1e6,
                      "unit": "Mrays/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "default scene + %dx%d height field (%d triangles), %d incoherent rays"
                                 % (n, n, 2 * n * n, nr)},
                      "roofline": {"kernel": "k_trace_api_stream", "bound": "hbm",
                                   "achieved": bytes_per_ray * nr / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": bytes_per_ray * nr / dt / 1e9 / 8000.0, "bytes_per_ray": bytes_per_ray}}))
