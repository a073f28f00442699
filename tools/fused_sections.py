"""Development helper: shader cycles the waves of the fused kernel spend in each section of a
round (library built with -DRTR_STREAM_STATS -DRTR_SECTION_TIMES, named by RTR_HIP_SO).  With four
waves per SIMD interleaved a section's figure includes the time its wave waited for the others,
so shares are rough."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
lib = rt.hip_lib()
sg = rt.default_scene()
sg.render(784, 784, 64)
out = (C.c_ulonglong * 8)()
assert lib.rtr_debug_stream_stats(out) == 0
v = [int(x) for x in out[:7]]
names = ["new sample", "shade vertex", "leaf pass", "marked spheres", "boxes, re-trace", "consume + deal units"]
total = sum(v[:6])
for n, c in zip(names, v[:6]):
    print("%-22s %5.1f %%  %8.0f cycles per round" % (n, 100.0 * c / total, c / max(v[6], 1)))
