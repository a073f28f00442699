"""Development helper: one rank's share at N = 8 for several chunk sizes and tile widths."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import raytracingrenderer_amd as rt
from raytracingrenderer_amd.distributed import local_buffer
sg = rt.default_scene(); sg.set_device(0); sg.device_scene()
dev = torch.device("cuda", 0)
world = 8
for tile in (8, 7):
    local = local_buffer(torch, 784, 784, world, tile, dev)
    for chunk in (4, 8, 16):
        ts = []
        for rank in (0, 5):
            kw = dict(part_index=rank, part_count=world, tile_cols=tile, chunk_spp=chunk)
            sg.render_device(784, 784, 1024, local.data_ptr(), **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                sg.render_device(784, 784, 1024, local.data_ptr(), **kw)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
        print("tile %d chunk %d: ms %s (ideal %.2f)" % (tile, chunk, ["%.2f" % t for t in ts], 114.9 / 8), flush=True)
