"""Time the one-launch EFFN forward (csrc/oss_effn.hip) against the launch-per-layer chain at the inference shapes (no_grad, 16-bit).
   python tools/effn_bench.py [f16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import oss_block
from vmambair_amd.ops import ffn as ffn_ops

dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float16
dev = "cuda:0"
shapes = [(1, 96, 512, 512), (1, 48, 512, 512), (1, 96, 256, 256), (4, 96, 272, 272), (4, 48, 272, 272), (4, 96, 160, 160), (4, 48, 160, 160),
          (1, 96, 160, 160), (8, 96, 64, 64), (8, 48, 64, 64)]
for B, D, H, W in shapes:
    torch.manual_seed(0)
    norm = oss_block.LayerNorm(D, "WithBias").to(dev)
    ff = oss_block.FeedForward(D, 2.66, False).to(dev)
    x = torch.randn(B, D, H, W, device=dev).to(dt)
    res = {}
    for name, flag in (("chain", False), ("fused", True)):
        ffn_ops.EFFN_FUSED = flag
        with torch.no_grad():
            for _ in range(3):
                y = ff(x, pre_norm=norm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = ff(x, pre_norm=norm)
            e1.record()
            torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    ffn_ops.EFFN_FUSED = True
    mb = 2 * x.numel() * 2 / 1e6
    print(f"{dt} x {tuple(x.shape)}: chain {res['chain']:8.1f} us  fused {res['fused']:8.1f} us  ({res['chain'] / res['fused']:.2f} x; "
          f"{mb / res['fused'] * 1e3:.0f} GB/s of block input + output)", flush=True)
