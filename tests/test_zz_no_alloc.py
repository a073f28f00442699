"""The render entry point that allocates nothing (rtr_render_pt_device_noalloc) against the
allocating one (rtr_render_pt_device), through the C ABI: same image bit for bit, on two
calls in a row that share the library's fixed scratch array, the second smaller than the
first; its limits are refused with a message; and the device's free memory is the same
before and after its calls, also when it is the first render a scene ever does."""
import numpy as np
import pytest

import _rays

pytestmark = pytest.mark.gpu

f32 = np.float32


def _pair(sg, torch, W, H, spp, **kw):
    import raytracingrenderer_amd as rt
    ncol = len(rt.partition_columns(W, kw.get("part_index", 0), kw.get("part_count", 1),
                                    kw.get("tile_cols", 8)))
    a = torch.full((ncol * H * 3,), -1.0, dtype=torch.float32, device="cuda:0")
    b = torch.full((ncol * H * 3,), -2.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    st = sg.render_device_noalloc(W, H, spp, a.data_ptr(), **kw)
    kw = dict(kw)
    kw["chunk_spp"] = st.chunk_spp
    st2 = sg.render_device(W, H, spp, b.data_ptr(), **kw)
    torch.cuda.synchronize()
    assert st.samples == st2.samples == ncol * H * spp and st.n_columns == st2.n_columns == ncol
    ga, gb = a.cpu().numpy().reshape(-1, 3), b.cpu().numpy().reshape(-1, 3)
    eq = _rays.same_bits(ga, gb)
    assert eq.all(), (W, H, spp, np.nonzero(~eq.all(1))[0][:10])
    assert gb.mean() > 0.005
    return st


def test_no_alloc_render_equals_allocating_render():
    import torch
    import raytracingrenderer_amd as rt
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    sg = rt.default_scene()
    # the largest image it takes, many partial sums per pixel ...
    st = _pair(sg, torch, 512, 512, 16, seed=5)
    assert 512 * 512 * ((16 + st.chunk_spp - 1) // st.chunk_spp) <= 1 << 20
    # ... then a smaller one over the same scratch array, then one partition of a wider image
    _pair(sg, torch, 96, 72, 24, seed=6)
    _pair(sg, torch, 640, 200, 8, seed=7, part_index=1, part_count=4, tile_cols=8, max_depth=3)
    # a scene whose tree is walked in memory (the other flavour of the kernel)
    import test_gpu_parity as tg
    sw = rt.HostScene.from_spec(tg._random_scene_spec(5))
    _pair(sw, torch, 64, 48, 6, seed=8)
    _pair(sw, torch, 32, 16, 3, seed=8)
    # "allocates nothing": the device's free memory does not move across its calls -- on a scene
    # that has rendered before, and on a fresh one whose first render this is (nothing of the
    # allocating call's workspace exists for it then)
    out = torch.zeros(3 * 600 * 600, dtype=torch.float32, device="cuda:0")
    fresh = rt.default_scene()
    fresh.device_scene()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for scene, (W, H, spp) in ((sg, (512, 512, 8)), (sg, (100, 60, 5)), (fresh, (512, 512, 4)), (fresh, (64, 64, 2)),
                               (sw, (128, 128, 4))):
        scene.render_device_noalloc(W, H, spp, out.data_ptr(), seed=9)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free0, (W, H, spp)
    # (the allocating call does move it: the check above can fail)
    fresh.render_device(512, 512, 4, out.data_ptr(), seed=9)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] < free0
    with pytest.raises(rt.RtrError, match="pixels"):
        sg.render_device_noalloc(600, 600, 4, out.data_ptr())
    with pytest.raises(rt.RtrError, match="chunk_spp"):
        sg.render_device_noalloc(512, 512, 64, out.data_ptr(), chunk_spp=1)
    with pytest.raises(rt.RtrError, match="wavefront"):
        sg.render_device_noalloc(64, 64, 4, out.data_ptr(), mode=rt.MODE_WAVEFRONT)
    with pytest.raises(rt.RtrError, match="profile"):
        sg.render_device_noalloc(64, 64, 4, out.data_ptr(), profile=True)
