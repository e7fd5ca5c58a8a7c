"""Host-side mirror of the reference's OSS block: ``LayerNorm``, ``FeedForward`` (EFFN),
``SS2D_1`` (Omni Selective Scan module: four spatial + two channel scan directions) and
``MamberBlock``.

Reference (read for behaviour, nothing copied):
  SRGAN/VmambaIR/archs/MambaSISR6_arch.py:144-218 (LayerNorm, FeedForward), :222-498 (SS2D_1),
  :502-515 (MamberBlock); Deraining/basicsr/models/archs/mamber32_arch.py:491-492 (additive
  channel gate), mamber33_arch.py:257,487-490 (dc_inner = 2); RealSR/VmambaIR/archs/
  MambaRealSR11_arch.py:478-534,582-657,806-817 (rank-ceil(D/16) channel scan with one row per
  direction, S4D-initialised Ac_logs).

Parameter names, shapes and initial distributions are the reference's, so released checkpoints
(``{'params': ...}`` state dicts) load with ``strict=True``.  Both scans go through
``vmambair_amd.selective_scan.selective_scan_fn`` -> ``torch.ops.vmambair`` (HIP only).
"""
from __future__ import annotations

import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import (effn_fwd_ok, effn_round_weights,
                  ChannelGateFn, ConvCoreFn, NormChannelGateFn, SS2DCoreFn, chan_supported, conv1x1, conv_core_ok, core_supported, dwconv3x3, dwconv3x3_gelu_gate, gelu_gate, layer_norm_nchw, ln_conv1x1, ln_conv1x1_ok,
                  split_halves)
from .selective_scan import CrossScan2, OmniScanFn, OmniScanMergeFn, SelectiveScanFP32, selective_scan_fn

#: ``VMAMBAIR_NORM_CHAN_FUSED=0``: out_norm and the channel gate stay two autograd nodes (the gate's backward writes d y2 in a pass
#: of its own) -- A-B timing of ops/channel.py: NormChannelGateFn
NORM_CHAN_FUSED = os.environ.get("VMAMBAIR_NORM_CHAN_FUSED", "1") == "1"

#: per reference tree: (dc_inner or None for the RealSR rank-R form, channel-gate mode)
VARIANTS = {
    "srgan": dict(dc_inner=4, gate="mul_add"),     # MambaSISR6_arch.py:263,495-496
    "mamber32": dict(dc_inner=4, gate="add"),      # mamber32_arch.py:260,491-492
    "mamber33": dict(dc_inner=2, gate="mul_add"),  # mamber33_arch.py:257,487-490
    "realsr": dict(dc_inner=None, gate="mul_add"), # MambaRealSR11_arch.py:589,806-817
}


class _LNBody(nn.Module):
    def __init__(self, dim: int, with_bias: bool):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        if with_bias:
            self.bias = nn.Parameter(torch.zeros(dim))
        else:
            self.bias = None


class LayerNorm(nn.Module):
    """Per-pixel LayerNorm over the channel axis of an NCHW tensor, biased variance, eps 1e-5
    (MambaSISR6_arch.py:144-195; ``BiasFree`` divides by sqrt(var + eps) without centring, :162-164)."""

    def __init__(self, dim: int, LayerNorm_type: str = "WithBias"):
        super().__init__()
        self.with_bias = LayerNorm_type != "BiasFree"
        self.body = _LNBody(dim, self.with_bias)

    def forward(self, x: torch.Tensor, gate: torch.Tensor = None, out_dtype: torch.dtype = None, passthrough: bool = False,
                gate_grad_into=None):
        """NCHW in, NCHW out, no permutes (HIP kernel).  ``gate``: fused ``* silu(gate)`` epilogue."""
        return layer_norm_nchw(x, self.body.weight, self.body.bias, gate, out_dtype, passthrough, gate_grad_into)


def _norm_then_conv(x: torch.Tensor, norm: "LayerNorm", conv: nn.Conv2d):
    """``conv(norm(x))`` and the alias of ``x`` that the block's skip connection adds back: ONE launch when the shapes allow
    (ops/pointwise.py: LNConv1x1Fn -- the LayerNorm runs on the 1x1 convolution's LDS-resident activation tile), else the
    LayerNorm launch followed by the convolution"""
    xa = x
    if torch.is_autocast_enabled("cuda") and x.is_cuda and x.dtype == torch.float32:
        xa = None   # an fp32 stream under autocast: the LayerNorm kernel narrows it itself (kept on the two-launch path)
    if xa is not None and ln_conv1x1_ok(xa, conv.weight):
        return ln_conv1x1(xa, norm.body.weight, norm.body.bias, conv)
    n, skip = norm(x, passthrough=True)
    return conv1x1(n, conv), skip


class FeedForward(nn.Module):
    """EFFN: 1x1 (D -> 2h) -> depth-wise 3x3 -> gelu(x1) * x2 -> 1x1 (h -> D), h = int(D * factor)
    (MambaSISR6_arch.py:201-218)."""

    def __init__(self, dim: int, ffn_expansion_factor: float, bias: bool):
        super().__init__()
        hidden = int(dim * ffn_expansion_factor)
        self.project_in = nn.Conv2d(dim, hidden * 2, kernel_size=1, bias=bias)
        self.dwconv = nn.Conv2d(hidden * 2, hidden * 2, kernel_size=3, stride=1, padding=1, groups=hidden * 2, bias=bias)
        self.project_out = nn.Conv2d(hidden, dim, kernel_size=1, bias=bias)

    def _rounded(self, dtype: torch.dtype):
        """the weights as the one-launch forward reads them, once per weight version (inference: constants) -- ops.ffn.effn_round_weights"""
        wi, wd, wo = self.project_in.weight, self.dwconv.weight, self.project_out.weight
        if wi.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a graph capture the one-launch rounding is captured too and replayed with the forward: a graph must not bake in
            # copies that an in-place weight update (load_state_dict, an optimizer step between validations) would leave stale
            return effn_round_weights(wi, wd, wo, dtype)
        if wi.is_inference() or wd.is_inference() or wo.is_inference():
            return effn_round_weights(wi, wd, wo, dtype)   # (inference tensors carry no version counter: nothing to key a cache on)
        key = (dtype, wi._version, wd._version, wo._version, wi.data_ptr(), wd.data_ptr(), wo.data_ptr())
        hit = getattr(self, "_rounded_cache", None)
        if hit is None or hit[0] != key:
            hit = (key, effn_round_weights(wi, wd, wo, dtype))
            self._rounded_cache = hit
        return hit[1]

    def forward(self, x: torch.Tensor, residual: torch.Tensor = None, pre_norm: "LayerNorm" = None) -> torch.Tensor:
        """``pre_norm``: x is the un-normalised stream and the block's norm2 is applied here, fused into project_in; the skip
        connection is then x itself"""
        if pre_norm is not None and not (torch.is_grad_enabled() and (x.requires_grad or self.project_in.weight.requires_grad)) \
                and self.project_in.bias is None and self.dwconv.bias is None and self.project_out.bias is None \
                and effn_fwd_ok(x, self.project_out.in_channels):
            # inference: norm2 -> project_in -> dwconv -> gate -> project_out -> + x as ONE launch (csrc/oss_effn.hip); the 2h- and
            # h-channel intermediates never reach memory
            w_in, w_dw, w_out = self._rounded(x.dtype)
            return torch.ops.vmambair.effn_fwd(x, pre_norm.body.weight, pre_norm.body.bias, w_in, w_dw, w_out, self.project_out.in_channels)
        if pre_norm is not None:
            t, residual = _norm_then_conv(x, pre_norm, self.project_in)
        else:
            t = conv1x1(x, self.project_in)
        # dwconv -> chunk -> gelu(x1) * x2 as one node that never stores the convolution (ops/dwconv.py: DWGateFn)
        return conv1x1(dwconv3x3_gelu_gate(t, self.dwconv), self.project_out, residual)


def _dt_proj_init(dt_rank: int, d_inner: int, dt_scale=1.0, dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4):
    """dt projection init (MambaSISR6_arch.py:337-362): weight U(-r^-0.5, r^-0.5); bias =
    softplus^-1 of dt ~ logU[dt_min, dt_max]."""
    std = dt_rank ** -0.5 * dt_scale
    weight = torch.empty(d_inner, dt_rank).uniform_(-std, std)
    dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
    bias = dt + torch.log(-torch.expm1(-dt))
    return weight, bias


def _a_log_init(d_state: int, rows: int) -> torch.Tensor:
    """S4D-real: A_log[d, n] = log(n + 1) (MambaSISR6_arch.py:365-379)."""
    return torch.log(torch.arange(1, d_state + 1, dtype=torch.float32)).repeat(rows, 1).contiguous()


class SS2D_1(nn.Module):
    """Omni Selective Scan module.  ``variant`` selects the reference tree (see VARIANTS)."""

    def __init__(self, d_model: int = 96, d_state: int = 16, ssm_ratio: float = 2.0, dt_rank="auto",
                 d_conv: int = 3, conv_bias: bool = True, variant: str = "srgan", **kwargs):
        super().__init__()
        cfg = VARIANTS[variant]
        self.variant = variant
        self.gate = cfg["gate"]
        d_expand = int(ssm_ratio * d_model)
        d_inner = int(min(2.0, ssm_ratio) * d_model)  # ssm_rank_ratio = 2.0 (MambaSISR6_arch.py:229,256-257)
        assert d_inner == d_expand, "low-rank SSM (d_inner < d_expand) is not used by any reference config"
        self.d_model, self.d_inner, self.d_state = d_model, d_inner, d_state
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.K, self.KC = 4, 2
        self.omni = True  # False: literal reference data flow (forward_core_xs)
        self.fused_merge = True  # scan + cross-merge as one autograd node (HIP merge kernel)
        self.fused_core = True   # ... and the flattenings and projections in front of it (SS2DCoreFn)
        self.fused_channel = True  # channel branch + gate as one autograd node (ChannelGateFn)
        R, N = self.dt_rank, d_state

        self.in_conv = nn.Conv2d(d_model, d_expand * 2, kernel_size=1)
        self.conv2d = nn.Conv2d(d_expand, d_expand, groups=d_expand, bias=conv_bias, kernel_size=d_conv,
                                padding=(d_conv - 1) // 2)
        self.out_norm = LayerNorm(d_inner, "WithBias")
        self.channel_norm = LayerNorm(d_inner, "WithBias")
        self.out_conv = nn.Conv2d(d_expand, d_model, kernel_size=1)

        # spatial branch parameters (MambaSISR6_arch.py:299-326)
        self.x_proj_weight = nn.Parameter(torch.stack(
            [nn.Linear(d_inner, R + 2 * N, bias=False).weight.detach() for _ in range(self.K)], dim=0))
        inits = [_dt_proj_init(R, d_inner) for _ in range(self.K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([w for w, _ in inits], dim=0))
        self.dt_projs_bias = nn.Parameter(torch.stack([b for _, b in inits], dim=0))
        self.A_logs = nn.Parameter(_a_log_init(N, self.K * d_inner))
        self.Ds = nn.Parameter(torch.ones(self.K * d_inner))
        self.A_logs._no_weight_decay = True
        self.Ds._no_weight_decay = True

        # channel branch parameters
        self.dc_inner = cfg["dc_inner"]
        if self.dc_inner is not None:  # SRGAN / Deraining (MambaSISR6_arch.py:263-268,306-312,330-333)
            dc, Rc = self.dc_inner, 6
            self.dtc_rank, self.dc_state = Rc, 16
            self.conv_cin = nn.Conv2d(1, dc, kernel_size=1)
            self.conv_cout = nn.Conv2d(dc, 1, kernel_size=1)
            self.xc_proj_weight = nn.Parameter(torch.stack(
                [nn.Linear(dc, Rc + 2 * 16, bias=False).weight.detach() for _ in range(self.KC)], dim=0))
            self.Dsc = nn.Parameter(torch.ones(self.KC * dc))
            self.Ac_logs = nn.Parameter(torch.randn(self.KC * dc, 16))
            self.dtc_projs_weight = nn.Parameter(torch.randn(self.KC, dc, Rc))
            self.dtc_projs_bias = nn.Parameter(torch.randn(self.KC, dc))
        else:  # RealSR (MambaRealSR11_arch.py:627-657): one row per direction, rank R, S4D init
            self.dtc_rank, self.dc_state = R, N
            self.xc_proj_weight = nn.Parameter(torch.stack(
                [nn.Linear(1, R + 2 * N, bias=False).weight.detach() for _ in range(self.KC)], dim=0))
            cinits = [_dt_proj_init(R, 1) for _ in range(self.KC)]
            self.dtc_projs_weight = nn.Parameter(torch.stack([w for w, _ in cinits], dim=0))
            self.dtc_projs_bias = nn.Parameter(torch.stack([b for _, b in cinits], dim=0))
            self.Ac_logs = nn.Parameter(_a_log_init(N, self.KC))
            self.Dsc = nn.Parameter(torch.ones(self.KC))
            self.Ac_logs._no_weight_decay = True
            self.Dsc._no_weight_decay = True

    # -- four spatial directions (MambaSISR6_arch.py:395-436; index maps: SURVEY.md Appendix B) --
    def forward_core(self, x: torch.Tensor, gate: torch.Tensor = None, gate_grad_into=None) -> torch.Tensor:
        """Omni form: two flattenings of x instead of the four of ``cross_scan_2d``; the projections of
        directions 2/3 are computed on the un-flipped rows (a column of a matmul does not depend on
        its position) and the scan kernels walk those directions backwards.  Same values as
        ``forward_core_xs`` (the literal reference data flow, kept for the bit-exact index-map test)."""
        if not self.omni:
            return self.forward_core_xs(x, gate)
        if self.fused_core and core_supported(self.d_inner, self.dt_rank, self.d_state):
            # flattenings + both projections + scan + merge as one autograd node on the HIP kernels
            y = SS2DCoreFn.apply(x, self.x_proj_weight, self.dt_projs_weight, self.A_logs, self.Ds, self.dt_projs_bias)
            return self._out_norm(y, x.dtype, gate, gate_grad_into)
        B, Cc, H, W = x.shape
        L = H * W
        R, N = self.dt_rank, self.d_state
        x2 = CrossScan2.apply(x)                                                       # (B, 2, D, L)
        z01 = torch.einsum("bjdl,jcd->bjcl", x2, self.x_proj_weight[0:2])
        z23 = torch.einsum("bjdl,jcd->bjcl", x2, self.x_proj_weight[2:4])
        x_dbl = torch.cat([z01, z23], dim=1)                                           # (B, 4, R+2N, L)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        args = (x2.view(B, -1, L), dts.contiguous().view(B, -1, L), self.A_logs, Bs, Cs, self.Ds,
                self.dt_projs_bias.view(-1))  # A = -exp(A_logs) is evaluated inside the kernels
        if self.fused_merge:
            # scan + merge ((y0 + flip y2) + T y1) + T flip y3 in fp32 as one autograd node
            y = OmniScanMergeFn.apply(*args, H, W)
        else:
            out = OmniScanFn.apply(*args).view(B, 4, -1, L)
            y = out[:, 0].float() + out[:, 2].float()
            y = y + out[:, 1].reshape(B, -1, W, H).transpose(2, 3).reshape(B, -1, L).float()
            y = y + out[:, 3].reshape(B, -1, W, H).transpose(2, 3).reshape(B, -1, L).float()
        return self._out_norm(y.view(B, Cc, H, W), x.dtype, gate)

    def _out_norm(self, y, dtype, gate, gate_grad_into=None):
        """out_norm(y).to(x.dtype) [* silu(gate)]  (MambaSISR6_arch.py:433-434,488-493)"""
        if isinstance(self.out_norm, LayerNorm):
            return self.out_norm(y, gate=gate, out_dtype=dtype, gate_grad_into=gate_grad_into)
        y = self.out_norm(y).to(dtype)  # tests swap in an Identity to look at the merge alone
        return y if gate is None else y * F.silu(gate)

    def forward_core_xs(self, x: torch.Tensor, gate: torch.Tensor = None) -> torch.Tensor:
        B, Cc, H, W = x.shape
        L = H * W
        R, N = self.dt_rank, self.d_state
        hw = x.flatten(2, 3)
        wh = x.transpose(2, 3).contiguous().flatten(2, 3)
        fwd2 = torch.stack([hw, wh], dim=1)
        xs = torch.cat([fwd2, fwd2.flip(-1)], dim=1)                                   # (B, 4, D, L)
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        out_y = selective_scan_fn(
            xs.reshape(B, -1, L), dts.contiguous().view(B, -1, L), -torch.exp(self.A_logs.float()), Bs, Cs, self.Ds,
            delta_bias=self.dt_projs_bias.view(-1), delta_softplus=True).view(B, 4, -1, L)
        # merge, in the reference's association order ((y0 + flip y2) + T y1) + T flip y3, fp32
        inv = out_y[:, 2:4].flip(-1)
        wh_y = out_y[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        invwh_y = inv[:, 1].reshape(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        y = out_y[:, 0].float() + inv[:, 0].float() + wh_y.float() + invwh_y.float()
        return self._out_norm(y.view(B, Cc, H, W), x.dtype, gate)

    # -- two channel directions over the pooled descriptor (MambaSISR6_arch.py:438-483) --
    def cforward_core(self, xc: torch.Tensor) -> torch.Tensor:
        """Omni form of the channel branch: the 1->dc_inner lift is an affine map, both directions are
        projected from the un-flipped rows and direction 1 is walked backwards inside the scan
        kernels (no stack / flip / conv launches on these tiny tensors).  Same values as
        ``cforward_core_ref`` (the literal reference data flow)."""
        if not self.omni:
            return self.cforward_core_ref(xc)
        b, d, h, w = xc.shape
        Rc, N = self.dtc_rank, self.dc_state
        # (b, 2*dc_inner, <=384) tensors: the whole branch runs in fp32 outside autocast -- the reference
        # trains in fp32 (its RealSR variant forces fp32 here even under AMP, MambaRealSR11_arch.py:513-519)
        # and every 16-bit round trip on these tiny tensors is a kernel launch, not a saving
        with torch.autocast("cuda", enabled=False):
            pooled = xc.mean(dim=(2, 3), dtype=torch.float32)                          # (b, L = d)
            if self.dc_inner is not None:
                dc = self.dc_inner
                seq = torch.addcmul(self.conv_cin.bias.float().view(1, dc, 1), pooled.unsqueeze(1),
                                    self.conv_cin.weight.float().view(1, dc, 1))       # conv_cin on a 1-channel map
            else:
                dc = 1
                seq = pooled.view(b, 1, d)
            z = torch.einsum("kcj,bjl->bkcl", self.xc_proj_weight.float(), seq)          # (b, 2, Rc+2N, L)
            dts, Bs, Cs = torch.split(z, [Rc, N, N], dim=2)
            dts = torch.einsum("bkrl,kjr->bkjl", dts, self.dtc_projs_weight.float()).reshape(b, 2 * dc, d)
            out = OmniScanFn.apply(seq, dts, self.Ac_logs, Bs, Cs, self.Dsc, self.dtc_projs_bias.view(-1)).view(b, 2, dc, d)
            y = out[:, 0] + out[:, 1]                                                   # direction 1 is stored un-flipped
            if self.dc_inner is not None:
                y = torch.matmul(self.conv_cout.weight.float().view(1, dc), y) + self.conv_cout.bias.float()  # (b, 1, L)
            y = F.layer_norm(y.reshape(b, d), (d,), self.channel_norm.body.weight.float(),
                             self.channel_norm.body.bias.float(), 1e-5)
        return y.view(b, d, 1, 1).to(xc.dtype)

    def cforward_core_ref(self, xc: torch.Tensor) -> torch.Tensor:
        b, d, h, w = xc.shape
        pooled = xc.mean(dim=(2, 3))                                                    # (b, d)
        if self.dc_inner is not None:
            seq = pooled.view(b, 1, d, 1)
            seq = self.conv_cin(seq).squeeze(-1)                                        # (b, dc, L = d)
            scan = selective_scan_fn
        else:
            seq = pooled.view(b, 1, d)                                                  # (b, 1, L = d)
            scan = SelectiveScanFP32.apply
        Bn, Dn, L = seq.shape
        Rc, N = self.dtc_rank, self.dc_state
        xsc = torch.stack([seq, seq.flip(-1)], dim=1)                                   # (b, 2, dc, L)
        xc_dbl = torch.einsum("bkdl,kcd->bkcl", xsc, self.xc_proj_weight)
        dts, Bs, Cs = torch.split(xc_dbl, [Rc, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dtc_projs_weight).contiguous()
        As = -torch.exp(self.Ac_logs.float())
        if self.dc_inner is None:  # MambaRealSR11_arch.py:513-519: everything in fp32
            args = (xsc.reshape(Bn, -1, L).float(), dts.view(Bn, -1, L).float(), As, Bs.contiguous().float(),
                    Cs.contiguous().float(), self.Dsc.float(), self.dtc_projs_bias.view(-1).float(), True, 1)
        else:
            args = (xsc.reshape(Bn, -1, L), dts.view(Bn, -1, L), As, Bs, Cs, self.Dsc,
                    self.dtc_projs_bias.view(-1), True, 1)
        out_y = scan(*args).view(Bn, 2, -1, L)
        y = out_y[:, 0].float() + out_y[:, 1].flip(-1).float()                           # (b, dc, L)
        if self.dc_inner is not None:
            y = self.conv_cout(y.unsqueeze(-1))                                         # (b, 1, L, 1)
            y = y.transpose(1, 2).contiguous()                                          # (b, L = d, 1, 1)
        else:
            y = y.transpose(1, 2).unsqueeze(2).contiguous()                             # (b, d, 1, 1)
        return F.layer_norm(y.reshape(b, d), (d,), self.channel_norm.body.weight, self.channel_norm.body.bias,
                            1e-5).view(b, d, 1, 1).to(xc.dtype)

    def forward(self, x: torch.Tensor, residual: torch.Tensor = None, pre_norm: "LayerNorm" = None) -> torch.Tensor:
        """``residual``: the block's skip connection, added in the epilogue of the out_conv kernel.  ``pre_norm``: x is the
        un-normalised stream and the block's norm1 is applied here, fused into in_conv; the skip connection is then x itself"""
        if pre_norm is not None:
            xz, residual = _norm_then_conv(x, pre_norm, self.in_conv)
        else:
            xz = conv1x1(x, self.in_conv)
        # x, z = xz.chunk(2, dim=1): the gradients of the halves are written by their producers into ONE buffer (no cat)
        x, z, pair = split_halves(xz)
        into = None if pair is None else (pair, 0)
        y_core = None
        if self.omni and self.fused_core and x.is_cuda and conv_core_ok(x, self.conv2d, self.d_inner, self.dt_rank, self.d_state):
            # (round 4) conv2d + silu + the spatial core as ONE node: the convolution writes the two flattenings the scans walk and
            # its backward reads their two gradients (ops/core.py: ConvCoreFn) -- no transpose / merge launches in between
            y_core = ConvCoreFn.apply(x, self.conv2d.weight, self.conv2d.bias, self.x_proj_weight, self.dt_projs_weight, self.A_logs,
                                      self.Ds, self.dt_projs_bias, into)
        else:
            x = dwconv3x3(x, self.conv2d, act=True, grad_into=into)  # silu in the conv's epilogue
        chan_fused = self.omni and self.fused_channel and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and \
            chan_supported(self.dc_inner or 1, self.dc_state, self.d_inner)
        lift = self.dc_inner is not None
        chan_args = (self.conv_cin.weight if lift else None, self.conv_cin.bias if lift else None, self.xc_proj_weight,
                     self.dtc_projs_weight, self.dtc_projs_bias, self.Ac_logs, self.Dsc, self.conv_cout.weight if lift else None,
                     self.conv_cout.bias if lift else None, self.channel_norm.body.weight, self.channel_norm.body.bias,
                     self.gate != "add") if chan_fused else None
        if chan_fused and NORM_CHAN_FUSED and self.fused_core and isinstance(self.out_norm, LayerNorm) and \
                core_supported(self.d_inner, self.dt_rank, self.d_state):
            # spatial core, then out_norm * silu(z) + channel branch + gate as ONE node: the gate's backward is folded into the
            # LayerNorm backward's load (ops/channel.py: NormChannelGateFn)
            y = y_core if y_core is not None else \
                SS2DCoreFn.apply(x, self.x_proj_weight, self.dt_projs_weight, self.A_logs, self.Ds, self.dt_projs_bias)
            y2 = NormChannelGateFn.apply(y, self.out_norm.body.weight, self.out_norm.body.bias, z, x.dtype,
                                         None if pair is None else (pair, 1), *chan_args)
            return conv1x1(y2, self.out_conv, residual)
        # out_norm(merge) * silu(z), fused in the LayerNorm kernel
        if y_core is not None:
            y2 = self._out_norm(y_core, x.dtype, z, None if pair is None else (pair, 1))
        else:
            y2 = self.forward_core(x, gate=z, gate_grad_into=None if pair is None else (pair, 1))
        if chan_fused and y2.dtype in (torch.float32, torch.float16, torch.bfloat16):
            # pooling + channel scans + LayerNorm + gate as one autograd node (oss_channel.hip)
            y2 = ChannelGateFn.apply(y2, *chan_args)
            return conv1x1(y2, self.out_conv, residual)
        c = self.cforward_core(y2)
        y2 = (y2 + c) if self.gate == "add" else torch.addcmul(y2, y2, c)  # y2 * c + y2
        return conv1x1(y2, self.out_conv, residual)


class MamberBlock(nn.Module):
    """x += SS2D_1(norm1(x)); x += EFFN(norm2(x))   (MambaSISR6_arch.py:502-515)."""

    def __init__(self, dim: int, num_heads: int = 1, ffn_expansion_factor: float = 2.66, bias: bool = False,
                 LayerNorm_type: str = "WithBias", variant: str = "srgan"):
        super().__init__()
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = SS2D_1(d_model=dim, ssm_ratio=1, variant=variant)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # x + f(norm(x)): the sum happens in the epilogue of f's last 1x1 conv, and the skip connection's gradient
        # re-enters through the LayerNorm node (``passthrough``), which adds it to dx inside its backward kernel
        # (round 3: the norms run inside the first 1x1 convolution of attn / ffn -- _norm_then_conv)
        x = self.attn(x, pre_norm=self.norm1)
        return self.ffn(x, pre_norm=self.norm2)
