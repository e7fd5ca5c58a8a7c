"""Which (K, M) shapes of the wave-level 16-bit 1x1 convolution kernels disagree with a float64 convolution of the same rounded
operands (round 6: found by the wide-K kernel's cross-check at K = 193, M = 33).  GPU box only."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import _capi, ops  # noqa: E402

lib = _capi.load()
dev = "cuda:0"
for dt in (torch.bfloat16,):
    for K in (127, 129, 191, 193, 200, 201, 255, 257, 300, 510):
        for M in (31, 32, 33, 48, 64, 65, 96, 97):
            for bias in (False, True):
                torch.manual_seed(K * 1000 + M)
                conv = torch.nn.Conv2d(K, M, 1, bias=bias).to(dev)
                x = torch.randn(1, K, 16, 16, device=dev).to(dt)
                y = ops.conv1x1(x, conv, None)
                ref = F.conv2d(x.double(), conv.weight.detach().to(dt).double(), conv.bias.detach().double() if bias else None)
                err = (y.double() - ref).abs()
                bad_rows = sorted(set((err > 0.02 * float(ref.abs().max())).nonzero()[:, 1].tolist()))
                if bad_rows:
                    print(f"K {K} M {M} bias {bias}: max err {float(err.max()):.3f} (max|ref| {float(ref.abs().max()):.2f}) bad rows {bad_rows[:8]}")
print("done")
