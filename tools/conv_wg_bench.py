"""Workgroup-level 1x1-conv kernel (oss_conv1x1_wg.hip) against the wave-level kernels of oss_conv1x1.hip through the SAME entry
points (oss_conv1x1_fwd / _dgrad; the dispatch switched with oss_conv1x1_set_wg): same values, time per call.
python tools/conv_wg_bench.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import _capi, ops  # noqa: E402
from op_bench import timeit  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
lib = _capi.load()
B, H = 8, 64
dt = torch.bfloat16
big_a, big_b = torch.zeros(1 << 26, device=dev), torch.zeros(1 << 26, device=dev)
us = timeit(lambda: lib.oss_hbm_copy(big_b.data_ptr(), big_a.data_ptr(), 1 << 28, torch.cuda.current_stream().cuda_stream))
print(f"copy 256 MiB: {2 * (1 << 28) / us / 1e6:.2f} TB/s (the box's health check)")
for ci, co, wt, with_res in ((96, 192, 0, 0), (96, 510, 0, 0), (96, 96, 0, 1), (48, 96, 0, 0), (48, 254, 0, 0), (48, 48, 0, 1),
                             (192, 96, 1, 0), (96, 96, 1, 0), (96, 255, 1, 0), (96, 48, 1, 0), (48, 127, 1, 0)):
    x = torch.randn(B, ci, H, H, device=dev).to(dt)
    res = torch.randn(B, co, H, H, device=dev).to(dt) if with_res else None
    if wt:   # input gradient: dy = x (B, Cout = ci, H, W), weight (Cout = ci, Cin = co) -> dx (B, Cin = co, H, W)
        w = torch.randn(ci, co, 1, 1, device=dev) / ci ** 0.5
        fn = lambda: ops.conv1x1_bwd_input(x, w) if hasattr(ops, "conv1x1_bwd_input") else None  # noqa: E731
        dxr = torch.empty(B, co, H, H, device=dev, dtype=dt)

        def fn(w=w, dxr=dxr):
            _capi.check(lib.oss_conv1x1_dgrad(ops._DT[dt], x.data_ptr(), w.data_ptr(), dxr.data_ptr(), B, ci, co, H * H,
                                              x.stride(0), x.stride(1), torch.cuda.current_stream().cuda_stream), "dgrad")
            return dxr
    else:
        w = torch.randn(co, ci, 1, 1, device=dev) / ci ** 0.5
        bias = torch.randn(co, device=dev)
        fn = lambda w=w, bias=bias, res=res: ops.conv1x1_fwd(x, w, bias, res)  # noqa: E731
    out, t = {}, {}
    for name, on, pix in (("wave", 0, 0), ("wg128", 1, 128), ("wg64", 1, 64)):
        lib.oss_conv1x1_set_wg(on, pix)
        out[name] = fn().clone()
        torch.cuda.synchronize()
        t[name] = timeit(fn)
    lib.oss_conv1x1_set_wg(1, 0)
    mb = (x.numel() + out["wave"].numel() * (2 if with_res else 1)) * 2 / 1e6
    d1 = float((out["wg128"].float() - out["wave"].float()).abs().max())
    d2 = float((out["wg64"].float() - out["wave"].float()).abs().max())
    print(f"K {ci:4d} M {co:4d} {'dgrad' if wt else 'fwd  '}{' +res' if with_res else '     '}: max|diff| {d1:.4f} {d2:.4f}   wave-level {t['wave']:6.1f} us   "
          f"wg128 {t['wg128']:6.1f} us ({mb / t['wg128']:.2f} TB/s)   wg64 {t['wg64']:6.1f} us", flush=True)
