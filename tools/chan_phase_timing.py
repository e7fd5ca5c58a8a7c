#!/usr/bin/env python3
"""Phase stamps of the channel-branch kernels (experiment build only: tools/build_experiment.sh CHAN_TIMING, selected with
VMAMBAIR_LIB=...).  Prints, per d_inner, the shader-cycle deltas between the phases of oss_chan_fwd / oss_chan_bwd of image 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmambair_amd import ops  # noqa: E402
from vmambair_amd.oss_block import SS2D_1  # noqa: E402

dev = "cuda:0"
FWD = ["pool->seq", "z proj", "dt proj", "scan", "cout+LN+stores"]
BWD = ["LN bwd", "scan (+cout grads)", "dt rows + dseq", "dpool", "param sums"]
for d in (48, 96, 192, 384):
    torch.manual_seed(0)
    m = SS2D_1(d_model=d, ssm_ratio=1, variant="srgan").to(dev)
    hw = {48: 64, 96: 32, 192: 16, 384: 8}[d]
    y2 = torch.randn(8, d, hw, hw, device=dev).to(torch.bfloat16)
    args = (y2, m.conv_cin.weight, m.conv_cin.bias, m.xc_proj_weight, m.dtc_projs_weight, m.dtc_projs_bias, m.Ac_logs, m.Dsc,
            m.conv_cout.weight, m.conv_cout.bias, m.channel_norm.body.weight, m.channel_norm.body.bias, True)
    args = tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in args)
    for rep in range(3):
        out, c, pooled, zt, dts, hs, y, yc, stat = torch.ops.vmambair.chan_gate_fwd(*args)
        torch.cuda.synchronize()
        f = zt.flatten()[:len(FWD)].tolist()
        g = torch.randn_like(out)
        dy2, gr = torch.ops.vmambair.chan_gate_bwd(g, y2, c, pooled, zt, dts, hs, y, yc, stat, *args[1:])
        torch.cuda.synchronize()
        bw = gr[:len(BWD)].tolist()
    print(f"d_inner {d}: fwd " + "  ".join(f"{n}={v / 2400:.1f}us" for n, v in zip(FWD, f)) + f"  total={sum(f) / 2400:.1f}us")
    print(f"d_inner {d}: bwd " + "  ".join(f"{n}={v / 2400:.1f}us" for n, v in zip(BWD, bw)) + f"  total={sum(bw) / 2400:.1f}us")
