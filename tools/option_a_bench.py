#!/usr/bin/env python3
"""What INTEGRATION.md "Option A" buys: the reference's own Python data flow, eager, with ONLY the scan swapped.

Option A puts this repository on PYTHONPATH so that the reference's ``import selective_scan_cuda_core`` resolves to the shim at
the repository root; everything else stays the reference's Python: ``nn.Conv2d`` / einsum / flip / stack / LayerNorm through
``to_3d`` and back as separate eager launches, four materialised flattenings, one scan call per module with the standard
(non-omni) argument form, no captured graph, the optimizer as torch's.  The reference tree does not exist on the GPU box, so this
tool runs THAT data flow on this package's module tree (same parameters, same shapes): every forward of ``LayerNorm``,
``FeedForward``, ``SS2D_1`` and ``MamberBlock`` and the net-level convolutions are replaced here by plain torch restatements of the
reference's lines (cited per function), and the scan goes through ``selective_scan_cuda_core.fwd / .bwd`` exactly as the
reference's ``SelectiveScan`` autograd function calls them (MambaSISR6_arch.py:44-88).

Prints one JSON line: images/s of the BASELINE.json configs[1] training step in this mode (fp32 = the reference's precision, and
bf16 autocast), next to which BASELINE.md section 4 puts the full path's numbers.  Not the product path: a measurement of the
smallest integration step only.
Usage (GPU box): python tools/option_a_bench.py [--steps 5] [--warmup 2] [--batch 8]
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("VMAMBAIR_CONV3X3_THIN", "0")   # net-level 3x3 convolutions through torch (MIOpen), as the reference
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import selective_scan_cuda_core  # noqa: E402  (the Option A shim)
import bench  # noqa: E402
from vmambair_amd import archs, oss_block  # noqa: E402


class SelectiveScan(torch.autograd.Function):
    """the reference's autograd wrapper around the native module (MambaSISR6_arch.py:44-88): fwd keeps (u, delta, A, B, C, D,
    delta_bias, x), bwd hands them back with dout"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        u, delta, B, C = (t if t.stride(-1) == 1 else t.contiguous() for t in (u, delta, B, C))
        if D is not None:
            D = D.contiguous()
        ctx.delta_softplus = delta_softplus
        out, x, *rest = selective_scan_cuda_core.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = selective_scan_cuda_core.bwd(
            u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, 1)
        return du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None


def scan(*a):
    return SelectiveScan.apply(*a)


def ln_forward(self, x, gate=None, out_dtype=None, passthrough=False, gate_grad_into=None):
    """to_3d -> (x - mu) / sqrt(var + 1e-5) * w + b over the channel axis -> to_4d (MambaSISR6_arch.py:144-195)"""
    t = x.permute(0, 2, 3, 1)
    sigma = t.var(-1, keepdim=True, unbiased=False)
    if self.with_bias:
        y = (t - t.mean(-1, keepdim=True)) / torch.sqrt(sigma + 1e-5) * self.body.weight + self.body.bias
    else:
        y = t / torch.sqrt(sigma + 1e-5) * self.body.weight
    y = y.permute(0, 3, 1, 2)
    if out_dtype is not None:
        y = y.to(out_dtype)
    if gate is not None:
        y = y * F.silu(gate)
    return (y, x) if passthrough else y


def ffn_forward(self, x, residual=None, pre_norm=None):
    """project_in -> dwconv -> chunk -> gelu(x1) * x2 -> project_out (MambaSISR6_arch.py:201-218)"""
    x1, x2 = self.dwconv(self.project_in(x)).chunk(2, dim=1)
    return self.project_out(F.gelu(x1) * x2)


def core_xs(self, x):
    """forward_corev1: four flattenings, one einsum per projection, one scan call, un-flip and merge
    (MambaSISR6_arch.py:395-436)"""
    B, Cc, H, W = x.shape
    L = H * W
    R, N = self.dt_rank, self.d_state
    x_hwwh = torch.stack([x.view(B, -1, L), torch.transpose(x, dim0=2, dim1=3).contiguous().view(B, -1, L)], dim=1).view(B, 2, -1, L)
    xs = torch.cat([x_hwwh, torch.flip(x_hwwh, dims=[-1])], dim=1)
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs.view(B, 4, -1, L), self.x_proj_weight)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts.view(B, 4, -1, L), self.dt_projs_weight)
    out_y = scan(xs.float().view(B, -1, L), dts.contiguous().float().view(B, -1, L), -torch.exp(self.A_logs.float()),
                 Bs.float().view(B, 4, -1, L), Cs.float().view(B, 4, -1, L), self.Ds.float(),
                 self.dt_projs_bias.float().view(-1), True, 1).view(B, 4, -1, L)
    inv_y = torch.flip(out_y[:, 2:4], dims=[-1]).view(B, 2, -1, L)
    wh_y = torch.transpose(out_y[:, 1].view(B, -1, W, H), dim0=2, dim1=3).contiguous().view(B, -1, L)
    invwh_y = torch.transpose(inv_y[:, 1].view(B, -1, W, H), dim0=2, dim1=3).contiguous().view(B, -1, L)
    y = out_y[:, 0].float() + inv_y[:, 0].float() + wh_y.float() + invwh_y.float()
    y = torch.transpose(y, dim0=1, dim1=2).contiguous().view(B, H, W, -1)
    y = F.layer_norm(y, (Cc,), self.out_norm.body.weight, self.out_norm.body.bias, 1e-5).to(x.dtype)
    return y.permute(0, 3, 1, 2)


def chan_ref(self, xc):
    """cforward_corev1: pooled descriptor, 1 -> dc_inner lift, two directions, one scan call (MambaSISR6_arch.py:438-483)"""
    b, d, h, w = xc.shape
    seq = self.conv_cin(xc.mean(dim=(2, 3)).view(b, 1, d, 1)).squeeze(-1)
    Bn, Dn, L = seq.shape
    Rc, N = self.dtc_rank, self.dc_state
    xsc = torch.stack([seq, torch.flip(seq, dims=[-1])], dim=1)
    xc_dbl = torch.einsum("b k d l, k c d -> b k c l", xsc, self.xc_proj_weight)
    dts, Bs, Cs = torch.split(xc_dbl, [Rc, N, N], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts, self.dtc_projs_weight).contiguous()
    out_y = scan(xsc.reshape(Bn, -1, L).float(), dts.view(Bn, -1, L).float(), -torch.exp(self.Ac_logs.float()), Bs.float(),
                 Cs.float(), self.Dsc.float(), self.dtc_projs_bias.float().view(-1), True, 1).view(Bn, 2, -1, L)
    y = out_y[:, 0].float() + torch.flip(out_y[:, 1], dims=[-1]).float()
    y = self.conv_cout(y.unsqueeze(-1)).transpose(1, 2).contiguous()
    return F.layer_norm(y.reshape(b, d), (d,), self.channel_norm.body.weight, self.channel_norm.body.bias, 1e-5).view(b, d, 1, 1)


def ss2d_forward(self, x, residual=None, pre_norm=None):
    """SS2D_1.forward (MambaSISR6_arch.py:485-498)"""
    x, z = self.in_conv(x).chunk(2, dim=1)
    x = F.silu(self.conv2d(x))
    y2 = core_xs(self, x) * F.silu(z)
    c = chan_ref(self, y2).to(y2.dtype)
    y2 = (y2 + c) if self.gate == "add" else (y2 * c + y2)
    return self.out_conv(y2)


def block_forward(self, x):
    """MamberBlock.forward (MambaSISR6_arch.py:502-515)"""
    x = x + self.attn(self.norm1(x))
    return x + self.ffn(self.norm2(x))


def install():
    oss_block.LayerNorm.forward = ln_forward
    oss_block.FeedForward.forward = ffn_forward
    oss_block.SS2D_1.forward = ss2d_forward
    oss_block.MamberBlock.forward = block_forward
    archs.conv1x1 = lambda x, conv, residual=None: conv(x)   # reduce_chan_level*: nn.Conv2d


def run(dtype, steps, warmup, batch):
    torch.manual_seed(0)
    net = archs.build_network(bench.NET).cuda()
    ema = [p.detach().clone() for p in net.parameters()]
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99))
    step = bench.make_step(net, ema, opt, dtype, "cuda")
    lq, gt = torch.rand(batch, 3, 64, 64, device="cuda"), torch.rand(batch, 3, 256, 256, device="cuda")
    for _ in range(warmup):
        step(lq, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(lq, gt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"images_per_s": round(batch / dt, 2), "ms_per_step": round(dt * 1e3, 2), "loss": round(float(loss), 5),
            "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--miopen-find", type=int, default=1)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(args.miopen_find)   # the reference sets it (SRGAN/VmambaIR/train_pipeline.py:97)
    install()
    out = {"mode": "INTEGRATION.md Option A: reference data flow, eager torch, only selective_scan_cuda_core swapped",
           "workload": "BASELINE.json configs[1]: x4 SR 64x64 LQ, MambaSISR6 dim48 [15,1,1,1]+15, batch %d, Adam + EMA" % args.batch,
           "steps": args.steps, "warmup": args.warmup, "device": torch.cuda.get_device_name(0)}
    out["fp32"] = run(None, args.steps, args.warmup, args.batch)
    torch.cuda.reset_peak_memory_stats()
    out["bf16_autocast"] = run(torch.bfloat16, args.steps, args.warmup, args.batch)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
