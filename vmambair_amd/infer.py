"""Tiled super-resolution inference with one hipGraph per distinct tile shape.

Restates the tiling rule of the reference's ``RealESRGANer.tile_process`` (RealSR/VmambaIR/utils.py:97-160; the SRGAN
tree's ``MambaSISRModel2.test`` uses the same scheme with 64-pixel tiles, SRGAN/VmambaIR/models/MambaSISR2_model.py:99-193):
the image is cut into ``tile`` x ``tile`` cells; every cell is fed to the net together with a halo of ``tile_pad`` pixels
clipped at the image border, and only the cell's own ``scale``-times enlarged area is written to the output.  With fixed
``tile`` / ``tile_pad`` an image produces at most nine padded-tile shapes (corner / edge / interior variants), so the forward
pass of each shape is captured once and replayed for every tile of that shape -- the net is launch-bound at tile sizes
(>= 1000 kernels per forward), which is exactly what a graph removes.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import torch


def tile_plan(height: int, width: int, tile: int, pad: int) -> Iterator[Tuple[int, int, int, int, int, int, int, int]]:
    """Yield, per cell in row-major order, ``(y0, y1, x0, x1, py0, py1, px0, px1)``: the cell ``[y0:y1, x0:x1]`` and the
    padded window ``[py0:py1, px0:px1]`` the net sees for it (utils.py:112-133)."""
    for y0 in range(0, height, tile):
        y1 = min(y0 + tile, height)
        py0, py1 = max(y0 - pad, 0), min(y1 + pad, height)
        for x0 in range(0, width, tile):
            x1 = min(x0 + tile, width)
            px0, px1 = max(x0 - pad, 0), min(x1 + pad, width)
            yield y0, y1, x0, x1, py0, py1, px0, px1


class TiledSR:
    """``out = TiledSR(net, scale)(img)``: ``img`` (B, C, H, W) -> (B, C, H*scale, W*scale), tile by tile.

    ``autocast_dtype``: run the net under autocast (fp16 = the reference's ``half=True`` inference, with fp32 master weights
    narrowed inside the kernels); ``use_graph``: capture one hipGraph per padded-tile shape (GPU only)."""

    def __init__(self, net: torch.nn.Module, scale: int = 4, tile: int = 128, tile_pad: int = 16,
                 autocast_dtype: Optional[torch.dtype] = torch.float16, use_graph: bool = True):
        assert tile > 0 and tile_pad >= 0
        self.net, self.scale, self.tile, self.pad = net.eval(), scale, tile, tile_pad
        self.autocast_dtype, self.use_graph = autocast_dtype, use_graph
        self._graphs: Dict[Tuple[int, int, int, int], Tuple[torch.cuda.CUDAGraph, torch.Tensor, torch.Tensor]] = {}
        self.tiles_run = 0

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        dev = x.device.type
        with torch.no_grad(), torch.autocast(dev, dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
            return self.net(x)

    def _run_tile(self, x: torch.Tensor) -> torch.Tensor:
        self.tiles_run += 1
        if not (self.use_graph and x.is_cuda):
            return self._forward(x)
        key = tuple(x.shape)
        if key not in self._graphs:
            static_in = x.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # warm-up outside the capture: lazy inits, vendor conv find, LDS attributes
                for _ in range(2):
                    self._forward(static_in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self._forward(static_in)
            self._graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = self._graphs[key]
        static_in.copy_(x, non_blocking=True)
        g.replay()
        return static_out

    @property
    def n_graphs(self) -> int:
        return len(self._graphs)

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        B, C, H, W = img.shape
        s = self.scale
        out = None
        for y0, y1, x0, x1, py0, py1, px0, px1 in tile_plan(H, W, self.tile, self.pad):
            o = self._run_tile(img[:, :, py0:py1, px0:px1].contiguous())
            if out is None:  # start from a black image of the net's output dtype (utils.py:109)
                out = o.new_zeros((B, o.shape[1], H * s, W * s))
            oy, ox = (y0 - py0) * s, (x0 - px0) * s          # the cell's own area inside the enlarged padded tile
            out[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = o[:, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
        return out
