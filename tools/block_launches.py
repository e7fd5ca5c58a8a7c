"""Which aten ops launch the small fill / copy / cat kernels of one MamberBlock training pass (bf16 autocast)?
python tools/block_launches.py  (GPU box) -> per aten op: calls, input shapes; and the python stack of each fill / copy / cat"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd.oss_block import MamberBlock  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
blk = MamberBlock(96, variant="srgan").to(dev)
x = torch.randn(8, 96, 64, 64, device=dev).to(torch.bfloat16).requires_grad_()   # the residual stream of the nets is 16-bit under autocast


def step():
    for p in blk.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x)
    y.float().abs().mean().backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::cat", "aten::contiguous", "aten::clone",
        "aten::_to_copy", "aten::add", "aten::add_", "aten::mul", "aten::sum")
rows = {}
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 0 or ev.name in ("aten::fill_", "aten::cat"):
        stack = [s for s in (ev.stack or []) if "vmambair_amd" in s or "block_launches" in s][:3]
        key = (ev.name, str(ev.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in stack)[:160])
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total
for (name, shapes, stack), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {name:18s} {us:8.1f} us  {shapes}  {stack}")
print("---- kernels")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=60).replace("  ", " "))
