"""Development helper: counters of the fused kernel's two-phase sphere test (library built
with -DRTR_STREAM_STATS as raytracingrenderer_amd/librtr_hip_stats.so, swapped in by the caller)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
lib = rt.hip_lib()
sg = rt.default_scene()
sg.render(784, 784, 64)
out = (C.c_ulonglong * 8)()
assert lib.rtr_debug_stream_stats(out) == 0
calls, iters, active, va, vb, live = [int(x) for x in out[:6]]
print("wave rounds %d  phase-2 iterations per round %.2f  lanes active per iteration %.1f  "
      "shadow rays per round %.1f  extension rays per round %.1f  lanes with a ray %.1f"
      % (calls, iters / calls, active / max(iters, 1), va / calls, vb / calls, live / calls))
