"""A few launches of the dominant scan call of the headline workload (u:(8,384,4096) bf16, omni form as SS2D_1 issues
it) -- the target of tools/pmc_traffic.sh (rocprofv3 --pmc passes) so that the counter output stays small."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
# SHAPE="B,D,L" DTYPE=bf16|f16|f32: other calls of the trees (Deraining level 0: 4,48,16384; RealSR tile: 1,96,25600 f16)
B, D, L = (int(v) for v in os.environ.get("SHAPE", "8,96,4096").split(","))
N = 16
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("DTYPE", "bf16")]
x2 = torch.randn(B, 2 * D, L, device=dev).to(dt)
delta = (torch.randn(B, 4 * D, L, device=dev) * 0.5).to(dt)
A_log = torch.log(torch.arange(1, N + 1, device=dev).float()).repeat(4 * D, 1).contiguous()
Bm = torch.randn(B, 4, N, L, device=dev).to(dt)
Cm = torch.randn(B, 4, N, L, device=dev).to(dt)
Dv = torch.ones(4 * D, device=dev)
bias = torch.randn(4 * D, device=dev) * 0.1
g2 = torch.randn(B, 2 * D, L, device=dev).to(dt)
# PARTIALS=bf16: the opt-in bf16 row-tile partials of the backward (oss_scan_bwd_params.tune_partials = 2) -- A-B of their traffic
tune = (None, None, None, "bf16") if os.environ.get("PARTIALS", "") == "bf16" else None
FWD_ONLY = os.environ.get("FWD_ONLY", "0") == "1"      # inference shapes (RealSR tiles, the untiled image): no backward
for _ in range(int(os.environ.get("REPS", "6"))):
    out, st = ops.selective_scan_fwd(x2, delta, A_log, Bm, Cm, Dv, bias, True, 1, 2, 2 * D, True)
    res = [out] if FWD_ONLY else ops.selective_scan_bwd(x2, delta, A_log, Bm, Cm, Dv, bias, g2, st, True, 1, 2, 2 * D, 2 * D, True, tune=tune)
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()), float(res[0].float().abs().mean()))
