"""Multi-GPU rendering: one process per GPU, columns partitioned, one gather.

Pixels are independent, so the image is cut into interleaved tiles of
``tile_cols`` columns (tile t belongs to rank t % world); each rank renders its
tiles with no traffic to the others, and a single ``gather`` (RCCL over xGMI when
the backend is "nccl") brings the packed column blocks to one rank, which puts
them back in column order.  Nothing else is exchanged.

The functions take any torch.distributed backend and device, so the same code
runs under ``gloo`` on CPU tensors in the tests.
"""
import numpy as np

from .api import partition_columns


def max_columns(W, world, tile_cols):
    return max(len(partition_columns(W, r, world, tile_cols)) for r in range(world))


def local_buffer(torch, W, H, world, tile_cols, device):
    """Packed framebuffer of one rank, padded to the largest rank's column count so
    that every rank contributes the same number of bytes to the gather."""
    return torch.zeros((max_columns(W, world, tile_cols) * H, 3), dtype=torch.float32, device=device)


class ColumnGather:
    """Pre-allocated buffers and the index tensor for gathering a W x H framebuffer: the ranks'
    packed blocks land side by side in one buffer, and one indexed copy (a single kernel on a
    GPU) puts every image column in its place."""

    def __init__(self, torch, dist, W, H, rank, world, tile_cols, device, dst=0, always_collective=False):
        self.torch, self.dist = torch, dist
        self.always_collective = always_collective  # run the gather even with one rank (rehearsal)
        self.W, self.H, self.rank, self.world, self.dst = W, H, rank, world, dst
        self.n_local = len(partition_columns(W, rank, world, tile_cols))
        self.is_dst = rank == dst
        if self.is_dst:
            cmax = max_columns(W, world, tile_cols)
            self.flat = torch.zeros((world * cmax * H, 3), dtype=torch.float32, device=device)
            self.parts = [self.flat[r * cmax * H: (r + 1) * cmax * H] for r in range(world)]
            # image column i is local column c of rank r: row r * cmax + c of the gathered blocks
            src = np.zeros(W, np.int64)
            for r in range(world):
                cols = partition_columns(W, r, world, tile_cols)
                src[cols] = r * cmax + np.arange(len(cols))
            self.src = torch.from_numpy(src).to(device)
            self.full = torch.zeros((W, H, 3), dtype=torch.float32, device=device)
        else:
            self.flat = self.parts = self.src = self.full = None

    def __call__(self, local):
        """local: this rank's packed columns (padded).  Returns the (W, H, 3) image on
        ``dst`` (buffer index i*H + j like the reference's), None elsewhere."""
        if self.world == 1 and not self.always_collective:
            n = self.n_local
            self.full.copy_(local[: n * self.H].view(n, self.H, 3))
            return self.full
        self.dist.gather(local, self.parts, dst=self.dst)
        if not self.is_dst:
            return None
        self.torch.index_select(self.flat.view(-1, self.H, 3), 0, self.src, out=self.full)
        return self.full


def render_distributed(scene, W, H, spp, torch, dist, tile_cols=8, dst=0, **render_kw):
    """Render on every rank's GPU and gather on ``dst``.  ``scene`` is a HostScene
    already placed on this rank's device.  Returns ((W,H,3) tensor or None, Stats)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    device = torch.device("cuda", torch.cuda.current_device())
    local = local_buffer(torch, W, H, world, tile_cols, device)
    st = scene.render_device(W, H, spp, local.data_ptr(), part_index=rank, part_count=world,
                             tile_cols=tile_cols, **render_kw)
    g = ColumnGather(torch, dist, W, H, rank, world, tile_cols, device, dst)
    return g(local), st
