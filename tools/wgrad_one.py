"""A few launches of the 1x1-conv backward at the dominant shape (d_model 96, 64x64, batch 8: in_conv 96 -> 384), target of
tools/pmc_kernel.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
B, ci, co, H = 8, 96, 384, 64
dt = torch.bfloat16
x = torch.randn(B, ci, H, H, device=dev).to(dt)
w = torch.randn(co, ci, 1, 1, device=dev) / ci ** 0.5
dy = torch.randn(B, co, H, H, device=dev).to(dt)
for _ in range(int(os.environ.get("REPS", "5"))):
    r = ops.conv1x1_bwd(x, w, dy, True)
torch.cuda.synchronize()
print("done", float(r[0].float().abs().mean()))
