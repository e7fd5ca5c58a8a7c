"""A few launches of the 1x1-conv forward kernels at the dominant shapes (batch 8, 64x64): in_conv 96 -> 384 (whole-K kernel,
16-byte weight loads), project_out 255 -> 96 (K-chunked kernel, element-wise weight loads), and their input gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
B, H = 8, 64
dt = torch.bfloat16
for ci, co in ((96, 384), (255, 96), (96, 510), (192, 96)):
    x = torch.randn(B, ci, H, H, device=dev).to(dt)
    w = torch.randn(co, ci, 1, 1, device=dev) / ci ** 0.5
    bias = torch.randn(co, device=dev)
    for _ in range(int(os.environ.get("REPS", "4"))):
        y = ops.conv1x1_fwd(x, w, bias)
    torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
