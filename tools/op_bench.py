"""Per-op timings at the shapes of the headline workload (batch 8, 64x64 LQ): our kernels next to the vendor
path they replace.  python tools/op_bench.py [--dtype bf16]   (GPU box; prints one line per op, us per call)"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=20, replays=5, warm=3):
    """GPU time per call: `reps` calls captured into one hipGraph (no host launch cost in the number), replayed"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / (reps * replays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    torch.manual_seed(0)
    B = 8
    rows = []
    shapes = [(48, 64), (96, 64)] if args.quick else [(48, 64), (96, 64), (96, 32), (192, 16), (384, 8)]
    from vmambair_amd import _capi
    tiny_a, tiny_b = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    lib = _capi.load()
    rows.append(("-", "tiny kernel (graph node floor)",
                 timeit(lambda: lib.oss_hbm_copy(tiny_b.data_ptr(), tiny_a.data_ptr(), 256, torch.cuda.current_stream().cuda_stream))))
    big_a, big_b = torch.zeros(1 << 26, device=DEV), torch.zeros(1 << 26, device=DEV)
    us = timeit(lambda: lib.oss_hbm_copy(big_b.data_ptr(), big_a.data_ptr(), 1 << 28, torch.cuda.current_stream().cuda_stream))
    rows.append(("-", f"copy 256 MiB ({2 * (1 << 28) / us / 1e6:.2f} TB/s)", us))
    for (dm, H) in shapes:
        D, L = 2 * dm, H * H
        R, N = max(1, -(-dm // 16)), 16
        Cc = R + 2 * N
        tag = f"d_model={dm} HxW={H}x{H}"
        x = torch.randn(B, D, H, H, device=DEV).to(dt)
        x2 = ops.cross_scan2(x)
        wx = torch.randn(4, Cc, D, device=DEV) / D ** 0.5
        wdt = torch.randn(4, D, R, device=DEV)
        rows.append((tag, "cross_scan2", timeit(lambda: ops.cross_scan2(x))))
        rows.append((tag, "proj_fwd", timeit(lambda: ops.proj_fwd(x2, wx, wdt))))
        xdbl, dts = ops.proj_fwd(x2, wx, wdt)
        ddts = torch.randn_like(dts)
        dxdbl = torch.randn_like(xdbl)
        du = torch.randn_like(dts)
        rows.append((tag, "proj_dgrad", timeit(lambda: ops.proj_dgrad(ddts, dxdbl, du, wx, wdt))))
        if dt != torch.float32:
            rows.append((tag, "proj_wgrad", timeit(lambda: ops.proj_wgrad(x2, xdbl, dxdbl, ddts, R))))
        # 1x1 convs of the block: in_conv (dm -> 2D), out_conv (D -> dm)
        xin = torch.randn(B, dm, H, H, device=DEV).to(dt)
        for (name, ci, co, inp) in [("in_conv", dm, 2 * D, xin), ("out_conv", D, dm, x)]:
            w = torch.randn(co, ci, 1, 1, device=DEV) / ci ** 0.5
            bias = torch.randn(co, device=DEV)
            w16, b16 = w.to(dt), bias.to(dt)
            rows.append((tag, f"{name} vendor fwd", timeit(lambda: F.conv2d(inp, w16, b16))))
            if dt != torch.float32:
                rows.append((tag, f"{name} mfma fwd", timeit(lambda: ops.conv1x1_fwd(inp, w, bias))))
                dy = torch.randn(B, co, H, H, device=DEV).to(dt)
                rows.append((tag, f"{name} mfma bwd", timeit(lambda: ops.conv1x1_bwd(inp, w, dy, True))))
                xi = inp.clone().requires_grad_()
                wi, bi = w16.clone().requires_grad_(), b16.clone().requires_grad_()

                def vendor_bwd():
                    y = F.conv2d(xi, wi, bi)
                    torch.autograd.grad(y, (xi, wi, bi), dy)
                rows.append((tag, f"{name} vendor fwd+bwd", timeit(vendor_bwd)))
        # LayerNorm
        lw, lb = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
        gate = torch.randn_like(x)
        rows.append((tag, "ln fwd (gated)", timeit(lambda: ops.ln_nchw_fwd(x, lw, lb, gate, ops._DT_CODE[dt]))))
        y, mean, rstd = ops.ln_nchw_fwd(x, lw, lb, gate, ops._DT_CODE[dt])
        rows.append((tag, "ln bwd (gated)", timeit(lambda: ops.ln_nchw_bwd(x, lw, lb, gate, y, mean, rstd))))
    for tag, name, us in rows:
        print(f"{tag:28s} {name:24s} {us:9.1f} us")


if __name__ == "__main__":
    main()
