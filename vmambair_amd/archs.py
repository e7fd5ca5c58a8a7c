"""UNets that stack the OSS block -- host-side mirror of the reference's registered archs, needed
here only because the headline metric (images/s of a ×4 SR training step) is quoted on the whole
net.  Module / parameter names equal the reference's so its checkpoints load unchanged.

Reference: ``MambaSISR6`` SRGAN/VmambaIR/archs/MambaSISR6_arch.py:520-643 (+ ``Upsampler`` /
``default_conv`` of archs/common.py:8-9,45-60), ``Mamber32`` Deraining/basicsr/models/archs/
mamber32_arch.py:519-649, ``MambaRealSR11`` RealSR/VmambaIR/archs/MambaRealSR11_arch.py:878-975.
All three are the same Restormer-shaped 4-level encoder/decoder; they differ in the OSS block
variant and in the tail.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .oss_block import MamberBlock, conv1x1
from .ops.conv3x3 import conv3x3


def _conv3(cin: int, cout: int, bias: bool) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, in_c: int = 3, embed_dim: int = 48, bias: bool = False):
        super().__init__()
        self.proj = _conv3(in_c, embed_dim, bias)

    def forward(self, x):
        return conv3x3(x, self.proj)   # 3 -> dim: the in-tree thin-convolution kernels for 16-bit activations


class Downsample(nn.Module):
    def __init__(self, n_feat: int):
        super().__init__()
        self.body = nn.Sequential(_conv3(n_feat, n_feat // 2, False), nn.PixelUnshuffle(2))

    def forward(self, x):
        return self.body(x)


class Upsample(nn.Module):
    def __init__(self, n_feat: int):
        super().__init__()
        self.body = nn.Sequential(_conv3(n_feat, n_feat * 2, False), nn.PixelShuffle(2))

    def forward(self, x):
        return self.body(x)


def _x4_tail(n_feat: int, out_channels: int) -> nn.Sequential:
    """conv(n -> 4n) + PixelShuffle(2), twice, then conv(n -> out), all 3x3 with bias
    (archs/common.py:45-60; MambaSISR6_arch.py:598-602)."""
    up = nn.Sequential(_conv3(n_feat, 4 * n_feat, True), nn.PixelShuffle(2),
                       _conv3(n_feat, 4 * n_feat, True), nn.PixelShuffle(2))
    return nn.Sequential(up, _conv3(n_feat, out_channels, True))


class _OSSUNet(nn.Module):
    variant = "srgan"

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=(6, 2, 2, 1), num_refinement_blocks=6,
                 heads=(1, 2, 4, 8), ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias"):
        super().__init__()

        def stage(width, count, head):
            return nn.Sequential(*[MamberBlock(dim=width, num_heads=head, ffn_expansion_factor=ffn_expansion_factor,
                                               bias=bias, LayerNorm_type=LayerNorm_type, variant=self.variant)
                                   for _ in range(count)])

        d1, d2, d3, d4 = dim, dim * 2, dim * 4, dim * 8
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level1 = stage(d1, num_blocks[0], heads[0])
        self.down1_2 = Downsample(d1)
        self.encoder_level2 = stage(d2, num_blocks[1], heads[1])
        self.down2_3 = Downsample(d2)
        self.encoder_level3 = stage(d3, num_blocks[2], heads[2])
        self.down3_4 = Downsample(d3)
        self.latent = stage(d4, num_blocks[3], heads[3])
        self.up4_3 = Upsample(d4)
        self.reduce_chan_level3 = nn.Conv2d(d4, d3, kernel_size=1, bias=bias)
        self.decoder_level3 = stage(d3, num_blocks[2], heads[2])
        self.up3_2 = Upsample(d3)
        self.reduce_chan_level2 = nn.Conv2d(d3, d2, kernel_size=1, bias=bias)
        self.decoder_level2 = stage(d2, num_blocks[1], heads[1])
        self.up2_1 = Upsample(d2)  # level 1 of the decoder keeps 2*dim channels (no 1x1 reduce)
        self.decoder_level1 = stage(d2, num_blocks[0], heads[0])
        self.refinement = stage(d2, num_refinement_blocks, heads[0])

    def flops(self, shape=(3, 64, 64)) -> str:
        """Counterpart of the reference's ``flops()`` (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:101-138,646-664: fvcore's
        ``flop_count`` with a handler that prices every ``SelectiveScan`` call at ``9 B L D N + B D L``) without fvcore and
        without running the net: the same counting rules applied ANALYTICALLY to the shapes this net's layers see for one
        ``(1, *shape)`` input.  A multiply-accumulate is one flop (fvcore's convention for convolutions / einsums); elementwise
        ops, the hand-written norms, activations, flips and shuffles are ignored (fvcore has no handler for them either); the six
        scans of an OSS module (four spatial directions as ONE call with D = 4 d_inner, two channel directions with
        D = 2 dc_inner) by the reference's formula.

        The count does not depend on which kernels the layers dispatch to (module forward hooks miss every Conv2d that
        runs through ``conv1x1`` / ``dwconv3x3`` / ``conv3x3``): each ``nn.Conv2d`` is priced from its own weight shape at the
        resolution of its UNet level, every Conv2d of the net exactly once (asserted).
        -> ``"params(M) <p> GFLOPs <g>"`` (the reference's format); ``flops_table`` holds the per-kind breakdown.
        Anchor: the reference publishes 10.50 M / 20.5 G for the RealSR net (README.md:82, figure) -- tests/test_host_logic.py."""
        from .oss_block import SS2D_1
        _, H, W = shape
        assert H % 8 == 0 and W % 8 == 0, "three 2x down-samplings"
        tally = {"conv": 0, "proj": 0, "scan": 0}
        seen = set()

        def conv(m: nn.Conv2d, h: int, w: int):
            assert id(m) not in seen, "a convolution counted twice"
            seen.add(id(m))
            assert m.stride == (1, 1), "every convolution of these nets keeps its resolution (re-sampling is Pixel(Un)Shuffle)"
            tally["conv"] += h * w * m.out_channels * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]

        def ss2d(m: SS2D_1, h: int, w: int):
            L, D, R, N = h * w, m.d_inner, m.dt_rank, m.d_state
            tally["proj"] += 4 * L * D * (R + 2 * N) + 4 * L * D * R                      # x_proj, dt_proj einsums
            tally["scan"] += 9 * L * (4 * D) * N + (4 * D) * L                            # flops_selective_scan_fn
            dc = m.dc_inner if m.dc_inner is not None else 1
            Rc, Nc, Lc = m.dtc_rank, m.dc_state, D                                        # channel branch: L = d_inner
            tally["proj"] += 2 * Lc * dc * (Rc + 2 * Nc) + 2 * Lc * dc * Rc
            tally["scan"] += 9 * Lc * (2 * dc) * Nc + (2 * dc) * Lc
            if m.dc_inner is not None:                                                    # conv_cin / conv_cout on the (1, 1, d, 1) map
                conv(m.conv_cin, Lc, 1)
                conv(m.conv_cout, Lc, 1)

        def tree(mod: nn.Module, h: int, w: int):
            """every Conv2d / OSS module below ``mod`` sees an (h, w) map; the OSS modules first, so that their conv_cin /
            conv_cout are priced on the channel map and not on (h, w)"""
            for sub in mod.modules():
                if isinstance(sub, SS2D_1):
                    ss2d(sub, h, w)
            for sub in mod.modules():
                if isinstance(sub, nn.Conv2d) and id(sub) not in seen:
                    conv(sub, h, w)

        def stages(level):
            return [self.encoder_level1, self.encoder_level2, self.encoder_level3, self.latent][level]

        res = [(H >> k, W >> k) for k in range(4)]
        tree(self.patch_embed, *res[0])
        for k in range(4):
            tree(stages(k), *res[k])
        for down, k in ((self.down1_2, 0), (self.down2_3, 1), (self.down3_4, 2)):
            tree(down, *res[k])                          # conv, then PixelUnshuffle
        for up, k in ((self.up4_3, 3), (self.up3_2, 2), (self.up2_1, 1)):
            tree(up, *res[k])                            # conv, then PixelShuffle
        for dec, red, k in ((self.decoder_level3, self.reduce_chan_level3, 2), (self.decoder_level2, self.reduce_chan_level2, 1),
                            (self.decoder_level1, None, 0), (self.refinement, None, 0)):
            if red is not None:
                conv(red, *res[k])
            tree(dec, *res[k])
        self._tail_flops(conv, H, W)
        missing = [n for n, m in self.named_modules() if isinstance(m, nn.Conv2d) and id(m) not in seen]
        assert not missing, f"convolutions without a price: {missing}"
        params = sum(p.numel() for p in self.parameters())
        self.flops_table = {k: v / 1e9 for k, v in tally.items()}
        return f"params(M) {params / 1e6} GFLOPs {sum(tally.values()) / 1e9}"

    def _tail_flops(self, conv, H: int, W: int):
        """price the convolutions behind ``body`` (the x4 tail / the output convolution): the nets below override it; the bare
        body has none, and ``flops()`` asserts that no Conv2d of the net is left without a price"""

    def body(self, inp_img: torch.Tensor) -> torch.Tensor:
        e1 = self.encoder_level1(self.patch_embed(inp_img))
        e2 = self.encoder_level2(self.down1_2(e1))
        e3 = self.encoder_level3(self.down2_3(e2))
        lat = self.latent(self.down3_4(e3))
        d3 = self.decoder_level3(conv1x1(torch.cat([self.up4_3(lat), e3], 1), self.reduce_chan_level3))
        d2 = self.decoder_level2(conv1x1(torch.cat([self.up3_2(d3), e2], 1), self.reduce_chan_level2))
        d1 = self.decoder_level1(torch.cat([self.up2_1(d2), e1], 1))
        return self.refinement(d1)


class MambaSISR6(_OSSUNet):
    """×scale SR net of the SRGAN tree (YAML ``network_g.type: MambaSISR6``,
    SRGAN/options/MambaSISR15_x4.yml:55-65)."""
    variant = "srgan"

    def __init__(self, inp_channels=3, out_channels=3, scale=4, dim=48, num_blocks=(6, 2, 2, 1),
                 num_refinement_blocks=6, heads=(1, 2, 4, 8), ffn_expansion_factor=2.66, bias=False,
                 LayerNorm_type="WithBias"):
        super().__init__(inp_channels, out_channels, dim, num_blocks, num_refinement_blocks, heads,
                         ffn_expansion_factor, bias, LayerNorm_type)
        self.scale = scale
        self.tail = _x4_tail(dim * 2, out_channels)

    def _tail_flops(self, conv, H, W):
        up, last = self.tail          # conv @ (H, W), shuffle, conv @ (2H, 2W), shuffle, conv_last @ (4H, 4W)
        conv(up[0], H, W)
        conv(up[2], 2 * H, 2 * W)
        conv(last, 4 * H, 4 * W)

    def forward(self, inp_img):
        # tail = Sequential(upsampler, conv_last): the last layer (2 dim -> out_channels at the output resolution) on the in-tree
        # thin-convolution kernels; parameter names (tail.0.*, tail.1.*) are the reference's
        return conv3x3(self.tail[0](self.body(inp_img)), self.tail[1]) + F.interpolate(inp_img, scale_factor=self.scale, mode="nearest")


class MambaRealSR11(MambaSISR6):
    """Real-world SR net (RealSR tree): same wiring, RealSR channel-scan variant."""
    variant = "realsr"


class Mamber32(_OSSUNet):
    """Deraining net: 3x3 output conv + global residual (mamber32_arch.py:608,647)."""
    variant = "mamber32"

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=(4, 6, 6, 8), num_refinement_blocks=2,
                 heads=(1, 2, 4, 8), ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias",
                 dual_pixel_task=False):
        super().__init__(inp_channels, out_channels, dim, num_blocks, num_refinement_blocks, heads,
                         ffn_expansion_factor, bias, LayerNorm_type)
        assert not dual_pixel_task, "dual-pixel defocus deblurring is not a config of the reference's options"
        self.output = _conv3(dim * 2, out_channels, bias)

    def _tail_flops(self, conv, H, W):
        conv(self.output, H, W)

    def forward(self, inp_img):
        return conv3x3(self.body(inp_img), self.output) + inp_img


class Mamber33(Mamber32):
    variant = "mamber33"


ARCHS = {"MambaSISR6": MambaSISR6, "MambaRealSR11": MambaRealSR11, "Mamber32": Mamber32, "Mamber33": Mamber33}


def build_network(opt: dict) -> nn.Module:
    """``network_g`` dict of a reference YAML (``type`` + kwargs) -> module, the way the
    reference's ARCH_REGISTRY resolves it (Deraining/basicsr/models/archs/__init__.py:6-46)."""
    opt = dict(opt)
    return ARCHS[opt.pop("type")](**opt)
