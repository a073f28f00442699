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
print("rays re-traced literally per wave round %.4f" % (out[6] / calls))
if "--causes" in sys.argv:  # library built with -DRTR_STREAM_STATS -DRTR_CAUSE_STATS
    print("lanes flagged per wave round, by cause: se range %.5f, delta range %.5f, numerator high %.5f, numerator tiny %.5f, "
          "distance sum %.5f, a range %.5f" % tuple(out[i] / calls for i in (1, 2, 3, 4, 5, 7)))
