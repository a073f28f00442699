"""The N>1 path under gloo on CPU: column partition, one gather, column re-ordering.

Each rank's "render" is stood in for by the CPU checker's image of the same
scene (test infrastructure), cut down to the columns that rank owns and packed
the way rtr_render_pt packs them; the code under test is the partition / gather /
re-assembly that bench.py and render_distributed use on the GPUs.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import raytracingrenderer_amd as rt
    from raytracingrenderer_amd.distributed import ColumnGather, local_buffer
    from _oracle import Oracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H, spp, tile = {W}, {H}, 2, {tile}
    O = Oracle(); so = O.default_scene()
    full, _ = O.render(so, W, H, spp, seed=4, threads=1)
    full = full.reshape(W, H, 3)
    cols = rt.partition_columns(W, rank, world, tile)
    local = local_buffer(torch, W, H, world, tile, torch.device("cpu"))
    local[: len(cols) * H] = torch.from_numpy(full[cols].reshape(-1, 3))
    g = ColumnGather(torch, dist, W, H, rank, world, tile, torch.device("cpu"), dst=0)
    out = g(local)
    if rank == 0:
        assert out is not None and tuple(out.shape) == (W, H, 3)
        assert np.array_equal(out.numpy().view(np.uint32), full.view(np.uint32))
        print("GATHER_OK", world)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("world,W,H,tile", [(2, 40, 12, 8), (3, 37, 9, 4)])
def test_gather_reassembles_image_under_gloo(tmp_path, world, W, H, tile):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, W=W, H=H, tile=tile))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 500) + world
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "GATHER_OK %d" % world in r.stdout
