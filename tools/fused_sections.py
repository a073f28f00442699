"""Development helper: where a wave of the fused kernel spends its cycles, section by section
(library built with -DRTR_STREAM_STATS -DRTR_SECTION_TIMES as librtr_hip_sect.so, swapped in by the caller)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import raytracingrenderer_amd as rt
lib = rt.hip_lib()
sg = rt.default_scene()
sg.render(784, 784, 16)
out = (C.c_ulonglong * 8)()
lib.rtr_debug_stream_stats(out)
sg.render(784, 784, 128)
assert lib.rtr_debug_stream_stats(out) == 0
names = ["deal units / new sample", "shade vertex", "leaf pass (phase 1)", "marked spheres (phase 2)", "winners' boxes, re-traces", "consume results"]
rounds = int(out[6])
tot = sum(int(out[i]) for i in range(6))
for i, n in enumerate(names):
    print("%-28s %8.0f cycles per wave round  %5.1f %%" % (n, out[i] / rounds, 100.0 * out[i] / tot))
print("%-28s %8.0f cycles per wave round (wave time; four waves share a SIMD)" % ("total", tot / rounds))
