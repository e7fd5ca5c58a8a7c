"""Tiled inference driver (vmambair_amd/infer.py): the tiling rule of the reference's RealESRGANer.tile_process
(RealSR/VmambaIR/utils.py:97-160) restated, and one hipGraph per padded-tile shape."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vmambair_amd.infer import TiledSR, tile_plan


def reference_rule(height, width, tile, pad):
    """the reference's arithmetic, written out independently of tile_plan (per-tile scalars as in utils.py:112-133)"""
    import math
    res = []
    for y in range(math.ceil(height / tile)):
        for x in range(math.ceil(width / tile)):
            ofs_x, ofs_y = x * tile, y * tile
            sx, ex = ofs_x, min(ofs_x + tile, width)
            sy, ey = ofs_y, min(ofs_y + tile, height)
            res.append((sy, ey, sx, ex, max(sy - pad, 0), min(ey + pad, height), max(sx - pad, 0), min(ex + pad, width)))
    return res


@pytest.mark.parametrize("hw", [(512, 512), (100, 70), (64, 64), (33, 129), (7, 5)])
@pytest.mark.parametrize("tile,pad", [(128, 16), (64, 8), (32, 0), (50, 10)])
def test_tile_plan_matches_the_reference_rule(hw, tile, pad):
    assert list(tile_plan(hw[0], hw[1], tile, pad)) == reference_rule(hw[0], hw[1], tile, pad)


@pytest.mark.parametrize("hw", [(40, 56), (33, 47)])
def test_tiling_is_exact_for_a_pointwise_upsampler(hw):
    """a net without spatial context (nearest x4 + per-pixel affine) must be reproduced exactly by the tiled driver:
    every output pixel is written exactly once, from the right place"""
    class Up(torch.nn.Module):
        def forward(self, x):
            return F.interpolate(x, scale_factor=4, mode="nearest") * 2.0 + 1.0

    torch.manual_seed(0)
    img = torch.randn(2, 3, *hw)
    drv = TiledSR(Up(), scale=4, tile=16, tile_pad=4, autocast_dtype=None, use_graph=False)
    assert torch.equal(drv(img), Up()(img))
    n = int(np.ceil(hw[0] / 16) * np.ceil(hw[1] / 16))
    assert drv.tiles_run == n


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [None, torch.float16], ids=["fp32", "fp16"])
def test_graph_replay_equals_eager_tiles(dt):
    from vmambair_amd.archs import MambaRealSR11
    torch.manual_seed(1)
    net = MambaRealSR11(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to("cuda:0")
    img = torch.rand(1, 3, 64, 48, device="cuda:0")
    eager = TiledSR(net, 4, tile=16, tile_pad=8, autocast_dtype=dt, use_graph=False)(img)
    drv = TiledSR(net, 4, tile=16, tile_pad=8, autocast_dtype=dt, use_graph=True)
    out = drv(img)
    assert out.shape == (1, 3, 256, 192)
    assert torch.equal(out, eager), "same kernels on the same data: the replay must be bit-identical"
    assert drv.n_graphs <= 9 and drv.tiles_run == 12
    out2 = drv(img * 0.5)   # second image: replays only
    assert torch.isfinite(out2).all() and drv.n_graphs <= 9
