#!/usr/bin/env python3
"""Host cost of one eager scan call through the two boundaries (GPU box): compiled TORCH_LIBRARY layer
(csrc_host/oss_torch_host.cpp) vs the ctypes marshalling of ops/scan.py.  The kernels of a tiny shape take a few
microseconds, so the wall time per call of a long eager loop is the host's.  Usage: python tools/host_overhead.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmambair_amd  # noqa: E402
from vmambair_amd import _host  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
B, KD, N, G, L = 1, 8, 16, 2, 64
u = torch.randn(B, KD, L, device=dev)
delta = 0.5 * torch.rand(B, KD, L, device=dev)
A = -0.5 * torch.rand(KD, N, device=dev)
Bm, Cm = torch.randn(B, G, N, L, device=dev), torch.randn(B, G, N, L, device=dev)
D, bias = torch.randn(KD, device=dev), 0.5 * torch.rand(KD, device=dev)
dout = torch.randn(B, KD, L, device=dev)
res = {}
for mode in ("c++", "ctypes", "c++", "ctypes"):
    _host.use(mode)
    out, x = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    torch.cuda.synchronize()
    n = 3000
    t0 = time.perf_counter()
    for _ in range(n):
        out, x = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / n
    res.setdefault(mode, []).append({"fwd_us_per_call": round(tf * 1e6, 1), "bwd_us_per_call": round(tb * 1e6, 1)})
_host.use(None)
print(json.dumps({"shape": [B, KD, L, G], "calls": 3000, "what": "wall time per eager call (host-bound at this shape)", **res}))
