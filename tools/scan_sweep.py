#!/usr/bin/env python3
"""Time every kernel variant of the HIP scan on the BASELINE shapes (GPU box only).

For each (shape, dtype, variant): N timed launches bracketed by the library's own HIP events
(oss_prof_*), reported as ms per launch and algorithmic GB/s (SURVEY.md 8d bytes / kernel time).
Also runs the library's copy kernel to put the achievable HBM bandwidth next to the numbers.
Usage: python tools/scan_sweep.py [--quick] > gpurun_out/sweep.txt
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmambair_amd  # noqa: E402
from vmambair_amd import _capi  # noqa: E402

DT = {"f32": (torch.float32, 0), "f16": (torch.float16, 1), "bf16": (torch.bfloat16, 2)}


def collect(lib, which, variant, io):
    ms, n, by = C.c_double(), C.c_longlong(), C.c_double()
    lib.oss_prof_collect(which, variant, io, C.byref(ms), C.byref(n), C.byref(by))
    return ms.value, n.value, by.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--fwd-variants", default="0,6")
    ap.add_argument("--bwd-variants", default="3,7")
    ap.add_argument("--shapes", default="", help="B,KD,L,G;... (default: the headline shapes)")
    ap.add_argument("--segs", default="-1", help="time segments per row to sweep (oss_scan_set_segments): -1 heuristic, 1 off, n")
    ap.add_argument("--dtypes", default="", help="f32,bf16,f16 (default f32,bf16; --quick: bf16)")
    args = ap.parse_args()
    lib = _capi.load()
    dev = "cuda:0"
    # copy-kernel peak
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.oss_hbm_copy(src.data_ptr(), dst.data_ptr(), n, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.oss_hbm_copy(src.data_ptr(), dst.data_ptr(), n, st)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    print(json.dumps({"copy_kernel_GBps": round(copy_gbs, 1)}), flush=True)
    del src, dst

    shapes = [(8, 192, 4096, 4)]
    if not args.quick:
        shapes += [(8, 768, 256, 4), (8, 8, 96, 2), (4, 384, 16384, 4)]
    if args.shapes:
        shapes = [tuple(int(v) for v in sh.split(",")) for sh in args.shapes.split(";")]
    fvs = tuple(int(v) for v in args.fwd_variants.split(",") if v != "")
    bvs = tuple(int(v) for v in args.bwd_variants.split(",") if v != "")
    for (B, KD, L, G) in shapes:
        for dname in (args.dtypes.split(",") if args.dtypes else (["f32", "bf16"] if not args.quick else ["bf16"])):
            dt, io = DT[dname]
            torch.manual_seed(0)
            u = torch.randn(B, KD, L, device=dev).to(dt)
            delta = (0.5 * torch.rand(B, KD, L, device=dev)).to(dt)
            A = -0.5 * torch.rand(KD, 16, device=dev)
            Bm = torch.randn(B, G, 16, L, device=dev).to(dt)
            Cm = torch.randn(B, G, 16, L, device=dev).to(dt)
            D = torch.randn(KD, device=dev)
            bias = 0.5 * torch.rand(KD, device=dev)
            dout = torch.randn(B, KD, L, device=dev).to(dt)
            for which, variants in ((0, fvs), (1, bvs)):
              for v in variants:
                for sg in [int(t) for t in args.segs.split(",")]:
                    lib.oss_scan_set_variant(v if which == 0 else -1, v if which == 1 else -1)
                    lib.oss_scan_set_segments(sg if which == 0 else -1, sg if which == 1 else -1)
                    try:
                        out, x = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
                        if which == 1:
                            vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
                        torch.cuda.synchronize()
                        lib.oss_prof_reset()
                        lib.oss_prof_enable(1)
                        for _ in range(args.reps):
                            if which == 0:
                                vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
                            else:
                                vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
                        torch.cuda.synchronize()
                        lib.oss_prof_enable(0)
                        vv = lib.oss_scan_last_variant(which)   # -1 = heuristic: the bucket of what it picked
                        bucket = vv + (16 if lib.oss_scan_last_segments(which) > 1 else 0)   # segmented launches: own bucket
                        ms, cnt, by = collect(lib, which, bucket, io)
                        fms = collect(lib, 2, bucket, io)[0] if which == 1 else 0.0
                        rec = {"kernel": "fwd" if which == 0 else "bwd", "shape": [B, KD, L, G], "dtype": dname,
                               "variant": vv, "segments": lib.oss_scan_last_segments(which), "launches": cnt,
                               "ms": round(ms / max(cnt, 1), 4), "finish_ms": round(fms / max(cnt, 1), 4),
                               "alg_GBps": round(by / max(ms, 1e-9) / 1e6, 1),
                               "Melem_per_s": round(B * KD * L * cnt / max(ms, 1e-9) / 1e3, 1)}
                    except Exception as e:  # keep sweeping
                        rec = {"kernel": "fwd" if which == 0 else "bwd", "shape": [B, KD, L, G], "dtype": dname,
                               "variant": v, "error": str(e)[:200]}
                    print(json.dumps(rec), flush=True)
            lib.oss_scan_set_variant(-1, -1)
            lib.oss_scan_set_segments(-1, -1)
            del u, delta, Bm, Cm, dout
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
