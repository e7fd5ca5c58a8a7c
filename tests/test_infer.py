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


# ------------------------------------------------------------------------------------------------
# round 2: the reference's own drivers, run on position-coded images with a recording stand-in network
# (tests/golden/g6_tiles.npz, written by tests/golden/make_golden.py::make_g6 from RealESRGANer.pre_process /
# tile_process / post_process, RealSR/VmambaIR/utils.py:68-171, and MambaSISRModel2.test, MambaSISR2_model.py:99-193)
# ------------------------------------------------------------------------------------------------
class Recorder(torch.nn.Module):
    """the stand-in make_g6 used: nearest upsampling + terms that depend on the whole window it is given, so that a
    wrong window, a wrong paste offset or a different tile order all change the result"""

    def __init__(self, scale):
        super().__init__()
        self.scale, self.shapes = scale, []
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, t):
        self.shapes.append(tuple(t.shape[-2:]))
        return F.interpolate(t.float(), scale_factor=self.scale, mode="nearest") + 1000.0 * t.float().mean() \
            + 7.0 * float(t.shape[-1]) + 3.0 * float(t.shape[-2])


def _g6():
    from conftest import GOLDEN
    import os
    return np.load(os.path.join(GOLDEN, "g6_tiles.npz"))


@pytest.mark.parametrize("case", range(5))
def test_realsr_enhancer_reproduces_the_reference_run(case):
    from vmambair_amd.infer import RealSREnhancer
    z = _g6()
    h, w, scale, tile, pad, pre = (int(v) for v in z[f"realsr_{case}.cfg"])
    net = Recorder(scale)
    drv = RealSREnhancer(net, scale, tile=tile, tile_pad=pad, pre_pad=pre, half=False, use_graph=False)
    padded = drv.pre_process(z[f"realsr_{case}.img"])
    assert torch.equal(padded, torch.from_numpy(z[f"realsr_{case}.padded"])), "pre-pad / mod-pad"
    drv.tile_process()
    out = drv.post_process()
    assert [list(s) for s in net.shapes] == z[f"realsr_{case}.shapes"].tolist(), "windows handed to the net, in order"
    # the stand-in's window mean is summed in a different order on a contiguous copy than on the reference's strided
    # view: values ~500 agree to fp32 rounding, while a wrong window or offset moves them by >= 1
    assert torch.allclose(out, torch.from_numpy(z[f"realsr_{case}.out"]), rtol=0, atol=5e-3)
    # the one-call form
    net2 = Recorder(scale)
    out2 = RealSREnhancer(net2, scale, tile=tile, tile_pad=pad, pre_pad=pre, use_graph=False).enhance_tensor(z[f"realsr_{case}.img"])
    assert torch.equal(out2, out)


@pytest.mark.parametrize("case", range(4))
def test_split64_reproduces_the_reference_run(case):
    from vmambair_amd.infer import Split64SR, split64_plan
    z = _g6()
    h, w, scale = (int(v) for v in z[f"srgan_{case}.cfg"])
    net = Recorder(scale)
    out = Split64SR(net, scale, use_graph=False)(torch.from_numpy(z[f"srgan_{case}.lq"]))
    assert [list(s) for s in net.shapes] == z[f"srgan_{case}.shapes"].tolist()
    assert torch.allclose(out, torch.from_numpy(z[f"srgan_{case}.out"]), rtol=0, atol=5e-3)
    mph, mpw, rows, cols = split64_plan(h, w)
    assert rows * cols == len(net.shapes) and (h + mph) % 64 == 0 and (w + mpw) % 64 == 0


def test_split64_rejects_what_the_reference_rejects():
    """reflect padding needs pad < size: a 30-row image cannot be padded to 64 (the reference raises the same error)"""
    from vmambair_amd.infer import Split64SR
    with pytest.raises(RuntimeError):
        Split64SR(Recorder(4), 4, use_graph=False)(torch.rand(1, 3, 30, 200))


def test_split64_batched_cells_equal_single_cells():
    """cells stacked on the batch axis give the same image for a net without cross-batch terms"""
    from vmambair_amd.infer import Split64SR

    class Local(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(3, 3, 3, padding=1)

        def forward(self, x):
            return F.interpolate(self.c(x), scale_factor=2, mode="nearest")

    torch.manual_seed(0)
    net, lq = Local(), torch.rand(1, 3, 150, 130)
    a = Split64SR(net, 2, use_graph=False, batch_tiles=1)(lq)
    b = Split64SR(net, 2, use_graph=False, batch_tiles=4)(lq)
    assert a.shape == (1, 3, 300, 260) and torch.allclose(a, b, atol=1e-6)


@pytest.mark.gpu
def test_split64_graph_on_gpu_matches_cpu_twin():
    """64-px cells of a small MambaSISR6 on the HIP kernels (ONE graph, replayed per group of cells) against the same
    cells through the CPU oracle twins.  The OSS net sees the whole cell (scan + pooled channel gate), so tiled and
    untiled outputs differ by design, in the reference as here: parity is per cell."""
    from conftest import assert_close, install_oracle_cpu_kernel
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.infer import Split64SR
    install_oracle_cpu_kernel()
    torch.manual_seed(2)
    net = MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    lq = torch.rand(1, 3, 100, 130)
    want = Split64SR(net, 4, use_graph=False)(lq)
    net_g = MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net_g.load_state_dict(net.state_dict())
    net_g.to("cuda:0")
    drv = Split64SR(net_g, 4, use_graph=True, batch_tiles=3)
    got = drv(lq.to("cuda:0"))
    assert drv.fwd.n_graphs == 1 and drv.fwd.calls == 2     # 6 cells in two replays of one graph
    assert_close(got, want, 2e-3, 2e-3, "split-64 output")
    half = Split64SR(net_g, 4, autocast_dtype=torch.float16, use_graph=True, batch_tiles=6)(lq.to("cuda:0"))
    assert_close(half, want, 3e-2, 3e-2, "split-64 output, fp16 autocast")


@pytest.mark.gpu
def test_realsr_enhancer_fp16_on_gpu_matches_cpu_twin():
    """config 5's driver (pre-pad 10, tile + halo, fp16) on a small MambaRealSR11 against the CPU twins in fp32"""
    from conftest import assert_close, install_oracle_cpu_kernel
    from vmambair_amd.archs import MambaRealSR11
    from vmambair_amd.infer import RealSREnhancer
    install_oracle_cpu_kernel()
    torch.manual_seed(3)
    net = MambaRealSR11(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    # 54x38 + pre-pad 10 = 64x48: every padded window (40 / 24 wide) is a multiple of 8, which the 3-level UNet needs
    img = np.random.RandomState(0).rand(54, 38, 3).astype(np.float32)
    want = RealSREnhancer(net, 4, tile=32, tile_pad=8, pre_pad=10, half=False, use_graph=False).enhance_tensor(img)
    net_g = MambaRealSR11(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net_g.load_state_dict(net.state_dict())
    net_g.to("cuda:0")
    drv = RealSREnhancer(net_g, 4, tile=32, tile_pad=8, pre_pad=10, half=True, use_graph=True)
    got = drv.enhance_tensor(img)
    assert got.shape == (1, 3, 216, 152) and drv.tiled.n_graphs <= 9
    assert_close(got, want, 3e-2, 3e-2, "tiled fp16 output")
    # the tiles of one padded shape stacked on the batch axis: the same image (and fewer forwards)
    drv4 = RealSREnhancer(net_g, 4, tile=32, tile_pad=8, pre_pad=10, half=True, use_graph=True, batch_tiles=4)
    got4 = drv4.enhance_tensor(img)
    assert_close(got4, want, 3e-2, 3e-2, "tiled fp16 output, tiles stacked")
    assert_close(got4, got, 2e-2, 2e-2, "stacked against tile by tile")
    assert drv4.tiled.tiles_run == drv.tiled.tiles_run
    # (round 4) the groups of different padded shapes replayed side by side, one HIP stream per shape: the same graphs and
    # kernels in an overlapping order -- bit-identical image, on the capturing call and on the replaying ones
    drv4c = RealSREnhancer(net_g, 4, tile=32, tile_pad=8, pre_pad=10, half=True, use_graph=True, batch_tiles=4, concurrent_shapes=True)
    first = drv4c.enhance_tensor(img).clone()
    for _ in range(3):
        again = drv4c.enhance_tensor(img)
        torch.cuda.synchronize()
        assert torch.equal(again, first)
    # (round 6) side by side the scans are never cut into time segments (infer.TiledSR: scan_tuning): where the sequential form's
    # heuristic segments a call, the two agree to fp32 round-off of the scan's carries instead of bit for bit
    assert_close(first, got4, 2e-3, 2e-3, "shape groups side by side against one after the other")
    assert len(drv4c.tiled._streams) == drv4c.tiled.n_graphs > 1


def test_tiles_stacked_on_the_batch_axis_give_the_same_image():
    """TiledSR(batch_tiles=4): tiles of one padded shape go through the net together -- same output as one by one (CPU, a
    net with per-sample statistics like the real one: instance-normalised 3x3 conv)"""
    from vmambair_amd.infer import TiledSR
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(3, 12, 3, padding=1)
        def forward(self, x):
            y = self.c(x)
            y = y - y.mean(dim=(2, 3), keepdim=True)       # depends on the whole tile, never on other batch entries
            return torch.nn.functional.pixel_shuffle(y, 2)
    net = Net()
    img = torch.rand(1, 3, 72, 88)
    one = TiledSR(net, 2, tile=32, tile_pad=4, autocast_dtype=None, use_graph=False)(img)
    for bt in (2, 4, 7):
        many = TiledSR(net, 2, tile=32, tile_pad=4, autocast_dtype=None, use_graph=False, batch_tiles=bt)(img)
        assert torch.allclose(many, one, atol=1e-6), bt
