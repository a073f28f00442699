"""Development helper: time of one rank's share of the benchmark image for N = 1, 2, 4, 8."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import raytracingrenderer_amd as rt
from raytracingrenderer_amd.distributed import local_buffer
sg = rt.default_scene(); sg.set_device(0); sg.device_scene()
dev = torch.device("cuda", 0)
for world in (1, 2, 4, 8):
    local = local_buffer(torch, 784, 784, world, 8, dev)
    ts = []
    for rank in range(min(world, 3)):
        sg.render_device(784, 784, 1024, local.data_ptr(), part_index=rank, part_count=world, tile_cols=8)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            st = sg.render_device(784, 784, 1024, local.data_ptr(), part_index=rank, part_count=world, tile_cols=8, profile=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
    print("world %d: per-rank ms %s  last call: total %.2f fused %.2f trace %.3f other %.3f" % (world, ["%.2f" % t for t in ts], st.ms_total, st.ms_fused, st.ms_trace, st.ms_other), flush=True)
