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

import contextlib
import math
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


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
                 autocast_dtype: Optional[torch.dtype] = torch.float16, use_graph: bool = True, batch_tiles: int = 1,
                 concurrent_shapes: bool = False):
        assert tile > 0 and tile_pad >= 0
        self.net, self.scale, self.tile, self.pad = net.eval(), scale, tile, tile_pad
        self.autocast_dtype, self.use_graph = autocast_dtype, use_graph
        # > 1: tiles of the same padded shape go through the net together, stacked on the batch axis (single-image input only).
        # The reference runs them one by one (utils.py:118-160); nothing in the nets mixes batch entries, so the result is the same.
        self.batch_tiles = max(1, int(batch_tiles))
        # (round 4) with ``batch_tiles``: the groups of DIFFERENT padded shapes (corner / edge / interior: four at 512 x 512) are
        # independent forwards, each too small to fill 256 CUs at batch 4 -- replay their graphs side by side, one HIP stream per
        # shape, and join before the result is read.  Same graphs, same kernels, same results; only the launch order overlaps.
        # The FIRST call captures every shape on the main stream, one after the other: the overlap starts with the second image.
        self.concurrent_shapes = bool(concurrent_shapes)
        self._streams: Dict[Tuple[int, ...], torch.cuda.Stream] = {}
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
        if self.concurrent_shapes and self.batch_tiles > 1 and img.shape[0] == 1 and img.is_cuda:
            # (round 6) the shape groups run side by side on their own streams and fill the GPU TOGETHER: a scan that cuts itself into
            # time segments because ITS OWN call leaves CUs idle only adds its local pass (512 x 512 in tiles of 128 + 16: 20.2 ->
            # 21.5 images/s with the segments off, profiles/r06_ab_realsr_segments.txt).  Per-call tune fields, baked into the graphs.
            from .ops.scan import scan_tuning
            with scan_tuning(fwd=(None, 1, None)):
                return self._call(img)
        return self._call(img)

    def _call(self, img: torch.Tensor) -> torch.Tensor:
        B, C, H, W = img.shape
        s = self.scale
        out = None
        if self.batch_tiles > 1 and B == 1:
            groups: Dict[Tuple[int, int], list] = {}
            for t in tile_plan(H, W, self.tile, self.pad):
                groups.setdefault((t[5] - t[4], t[7] - t[6]), []).append(t)
            graphed = self.use_graph and img.is_cuda
            side = graphed and self.concurrent_shapes
            main = torch.cuda.current_stream() if side else None
            used = []
            for tiles in groups.values():
                st = None
                if side:
                    shape_key = (tiles[0][5] - tiles[0][4], tiles[0][7] - tiles[0][6])
                    if (self.batch_tiles, C, *shape_key) in self._graphs:   # (first call: captured on the main stream, sequentially)
                        st = self._streams.get(shape_key)
                        if st is None:
                            st = self._streams[shape_key] = torch.cuda.Stream(device=img.device)
                        if out is None:   # the first group's output fixes dtype / channels: allocate before any side stream writes
                            o0 = self._graphs[(self.batch_tiles, C, *shape_key)][2]
                            out = o0.new_zeros((1, o0.shape[1], H * s, W * s))
                        st.wait_stream(main)
                        used.append(st)
                ctx = torch.cuda.stream(st) if st is not None else contextlib.nullcontext()
                with ctx:
                    for k in range(0, len(tiles), self.batch_tiles):
                        grp = tiles[k:k + self.batch_tiles]
                        chops = [img[:, :, py0:py1, px0:px1] for (_, _, _, _, py0, py1, px0, px1) in grp]
                        if graphed and len(grp) < self.batch_tiles:       # one graph per shape: pad the group with repeats
                            chops = chops + [chops[-1]] * (self.batch_tiles - len(grp))
                        o = self._run_tile(torch.cat(chops, 0).contiguous())
                        self.tiles_run += len(grp) - 1
                        if out is None:
                            out = o.new_zeros((1, o.shape[1], H * s, W * s))
                        for n, (y0, y1, x0, x1, py0, py1, px0, px1) in enumerate(grp):
                            oy, ox = (y0 - py0) * s, (x0 - px0) * s
                            out[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = o[n:n + 1, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
            for st in used:   # join: whoever reads `out` next runs after every shape's stream
                main.wait_stream(st)
            return out
        for y0, y1, x0, x1, py0, py1, px0, px1 in tile_plan(H, W, self.tile, self.pad):
            o = self._run_tile(img[:, :, py0:py1, px0:px1].contiguous())
            if out is None:  # start from a black image of the net's output dtype (utils.py:109)
                out = o.new_zeros((B, o.shape[1], H * s, W * s))
            oy, ox = (y0 - py0) * s, (x0 - px0) * s          # the cell's own area inside the enlarged padded tile
            out[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = o[:, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
        return out


class GraphedForward:
    """``net(x)`` under ``no_grad`` (+ autocast), one hipGraph per input shape -- shared by the drivers below."""

    def __init__(self, net: torch.nn.Module, autocast_dtype: Optional[torch.dtype], use_graph: bool = True):
        self._drv = TiledSR(net, 1, tile=1 << 30, tile_pad=0, autocast_dtype=autocast_dtype, use_graph=use_graph)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self._drv._run_tile(x)

    @property
    def n_graphs(self) -> int:
        return self._drv.n_graphs

    @property
    def calls(self) -> int:
        return self._drv.tiles_run


class RealSREnhancer:
    """The tensor part of the reference's ``RealESRGANer`` (RealSR/VmambaIR/utils.py:68-173): ``pre_process`` (reflect
    pre-pad on the right / bottom, then reflect mod-pad to a multiple of 2 for scale 2 and of 4 for scale 1, :74-91),
    ``tile_process`` / ``process`` (:93-160) and ``post_process`` (crop both pads off the enlarged image, :162-171).
    ``half=True`` is the reference's ``model.half()`` inference: here fp16 autocast over fp32 master weights.  Colour
    conversion / alpha handling of ``enhance`` (cv2) stay with the caller.  Pinned by tests/golden/g6_tiles.npz."""

    def __init__(self, net: torch.nn.Module, scale: int, tile: int = 0, tile_pad: int = 10, pre_pad: int = 10,
                 half: bool = False, use_graph: bool = True, device: Optional[torch.device] = None, batch_tiles: int = 1,
                 concurrent_shapes: bool = False):
        self.scale, self.tile_size, self.tile_pad, self.pre_pad, self.half = scale, tile, tile_pad, pre_pad, half
        self.device = device if device is not None else next(net.parameters(), torch.zeros(())).device
        acdt = torch.float16 if half else None
        graph = use_graph and torch.device(self.device).type == "cuda"
        self.tiled = TiledSR(net, scale, tile=max(tile, 1), tile_pad=tile_pad, autocast_dtype=acdt, use_graph=graph, batch_tiles=batch_tiles,
                             concurrent_shapes=concurrent_shapes)
        self.whole = GraphedForward(net, acdt, use_graph=graph)
        self.mod_scale = None
        self.mod_pad_h = self.mod_pad_w = 0

    def pre_process(self, img) -> torch.Tensor:
        """``img``: (H, W, C) array in [0, 1] (as ``enhance`` hands it over) or a (1, C, H, W) tensor"""
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).float().unsqueeze(0)
        t = img.to(self.device)
        if self.half:
            t = t.half()
        if self.pre_pad != 0:
            t = F.pad(t, (0, self.pre_pad, 0, self.pre_pad), "reflect")
        if self.scale == 2:
            self.mod_scale = 2
        elif self.scale == 1:
            self.mod_scale = 4
        self.mod_pad_h = self.mod_pad_w = 0
        if self.mod_scale is not None:
            _, _, h, w = t.shape
            if h % self.mod_scale != 0:
                self.mod_pad_h = self.mod_scale - h % self.mod_scale
            if w % self.mod_scale != 0:
                self.mod_pad_w = self.mod_scale - w % self.mod_scale
            t = F.pad(t, (0, self.mod_pad_w, 0, self.mod_pad_h), "reflect")
        self.img = t
        return t

    def process(self) -> torch.Tensor:
        self.output = self.whole(self.img)
        return self.output

    def tile_process(self) -> torch.Tensor:
        self.output = self.tiled(self.img)
        return self.output

    def post_process(self) -> torch.Tensor:
        out = self.output
        if self.mod_scale is not None:
            _, _, h, w = out.shape
            out = out[:, :, 0:h - self.mod_pad_h * self.scale, 0:w - self.mod_pad_w * self.scale]
        if self.pre_pad != 0:
            _, _, h, w = out.shape
            out = out[:, :, 0:h - self.pre_pad * self.scale, 0:w - self.pre_pad * self.scale]
        self.output = out
        return out

    @torch.no_grad()
    def enhance_tensor(self, img) -> torch.Tensor:
        self.pre_process(img)
        if self.tile_size > 0:
            self.tile_process()
        else:
            self.process()
        return self.post_process()


def split64_plan(h: int, w: int, split: int = 64) -> Tuple[int, int, int, int]:
    """-> (mod_pad_h, mod_pad_w, rows, cols) of ``MambaSISRModel2.test`` (SRGAN/VmambaIR/models/MambaSISR2_model.py:99-117)"""
    mod_pad_h = (h // split + 1) * split - h if h % split != 0 else 0
    mod_pad_w = (w // split + 1) * split - w if w % split != 0 else 0
    return mod_pad_h, mod_pad_w, (h + mod_pad_h) // split, (w + mod_pad_w) // split


class Split64SR:
    """Validation-time inference of the SRGAN tree (``MambaSISRModel2.test``, MambaSISR2_model.py:99-193): reflect-pad
    the LQ image on the right / bottom to multiples of 64, run the net on every non-overlapping 64x64 cell, paste the
    enlarged cells, crop the padding.  Every cell has ONE shape, so the forward is ONE hipGraph replayed per cell --
    or per group of ``batch_tiles`` cells stacked on the batch axis (cells are independent: nothing in the nets mixes
    batch entries), which is how a launch-bound 64x64 forward fills a 256-CU GPU.  The reference assembles the output
    on the CPU in fp32 (``torch.zeros(1, C, H*scale, W*scale)``, :164); here it stays on the device."""

    def __init__(self, net: torch.nn.Module, scale: int = 4, split: int = 64, autocast_dtype: Optional[torch.dtype] = None,
                 use_graph: bool = True, batch_tiles: int = 1):
        self.scale, self.split, self.batch_tiles = scale, split, max(1, int(batch_tiles))
        self.fwd = GraphedForward(net, autocast_dtype, use_graph)
        self.net = net

    @torch.no_grad()
    def __call__(self, lq: torch.Tensor) -> torch.Tensor:
        _, C, h, w = lq.shape
        s, sp = self.scale, self.split
        mph, mpw, rows, cols = split64_plan(h, w, sp)
        img = F.pad(lq, (0, mpw, 0, mph), "reflect") if (mph or mpw) else lq
        cells = [(i, j) for i in range(rows) for j in range(cols)]
        out = None
        use_graph = self.fwd._drv.use_graph and img.is_cuda
        for k in range(0, len(cells), self.batch_tiles):
            grp = cells[k:k + self.batch_tiles]
            chops = [img[..., i * sp:(i + 1) * sp, j * sp:(j + 1) * sp] for i, j in grp]
            if use_graph and len(grp) < self.batch_tiles:      # keep ONE graph shape: pad the last group with repeats
                chops = chops + [chops[-1]] * (self.batch_tiles - len(grp))
            o = self.fwd(torch.cat(chops, 0).contiguous())
            if out is None:
                out = torch.zeros((1, o.shape[1], rows * sp * s, cols * sp * s), dtype=torch.float32, device=o.device)
            for n, (i, j) in enumerate(grp):
                out[..., i * sp * s:(i + 1) * sp * s, j * sp * s:(j + 1) * sp * s] = o[n:n + 1]
        H, W = out.shape[-2:]
        return out[:, :, 0:H - mph * s, 0:W - mpw * s]
