"""Development helper: throughput of the photon emission pass (rtr_create_photon_map_device)
on the default scene, with the reference's own createPhotonMap beside it when
oracle/_ref/libref.so is present (its 16 threads, 15-photon calls excluded)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import raytracingrenderer_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sg = rt.default_scene()
lib = rt.hip_lib()
dev = torch.device("cuda", 0)
bufs = [torch.empty(3 * n, dtype=torch.float32, device=dev) for _ in range(3)]
stored = C.c_uint64(0); used = np.zeros(1, np.uint32)
def run(seed):
    rc = lib.rtr_create_photon_map_device(sg.device_scene(), n, seed, C.c_void_p(bufs[0].data_ptr()),
                                          C.c_void_p(bufs[1].data_ptr()), C.c_void_p(bufs[2].data_ptr()),
                                          C.byref(stored), used.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0, lib.rtr_last_error()
run(0)
t0 = time.time(); run(1); run(2); dt = (time.time() - t0) / 2
print("%d photons from %d emission paths in %.1f ms: %.1f Mphotons/s" % (stored.value, used[0], dt * 1e3, n / dt / 1e6))
