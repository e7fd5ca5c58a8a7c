"""A few launches of the one-launch EFFN forward at one shape (counter runs: tools/pmc_kernel.sh).  SHAPE=B,D,H,W DTYPE=f16|bf16 REPS=n"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import oss_block  # noqa: E402

B, D, H, W = (int(v) for v in os.environ.get("SHAPE", "1,96,512,512").split(","))
dt = torch.bfloat16 if os.environ.get("DTYPE", "f16") == "bf16" else torch.float16
torch.manual_seed(0)
norm = oss_block.LayerNorm(D, "WithBias").to("cuda:0")
ff = oss_block.FeedForward(D, 2.66, False).to("cuda:0")
x = torch.randn(B, D, H, W, device="cuda:0").to(dt)
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", "3"))):
        y = ff(x, pre_norm=norm)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
