"""Config 5 of BASELINE.json / SURVEY.md 8d: MambaRealSR11 [6,2,2,1]/6, fp16 forward under no_grad, a 512x512 image cut by the
tile rule (tile 128, halo 16) -> one hipGraph per padded-tile shape.  Prints one JSON line: tiles/s and images/s, graph replay
next to eager tiles.  python tools/infer_bench.py [--size 512] [--reps 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd.archs import MambaRealSR11  # noqa: E402
from vmambair_amd.infer import TiledSR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--tile", type=int, default=128)
    ap.add_argument("--pad", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    net = MambaRealSR11(dim=48, num_blocks=[6, 2, 2, 1], num_refinement_blocks=6).to(dev)
    img = torch.rand(1, 3, args.size, args.size, device=dev)
    res = {}
    for name, graph in (("eager", False), ("graph", True)):
        drv = TiledSR(net, 4, tile=args.tile, tile_pad=args.pad, autocast_dtype=torch.float16, use_graph=graph)
        out = drv(img)                      # warm-up / capture
        torch.cuda.synchronize()
        n_tiles = drv.tiles_run
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = drv(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res[name] = {"s_per_image": round(dt, 4), "images_per_s": round(1.0 / dt, 3), "tiles_per_s": round(n_tiles / dt, 2),
                     "tiles_per_image": n_tiles, "graphs": drv.n_graphs}
        assert out.shape[-1] == args.size * 4 and torch.isfinite(out.float()).all()
    print(json.dumps({"metric": "tiled x4 real-world SR inference, MambaRealSR11 [6,2,2,1]/6, fp16 autocast, no_grad",
                      "image": [args.size, args.size], "tile": args.tile, "tile_pad": args.pad, "data": "synthetic", **res}))


if __name__ == "__main__":
    main()
