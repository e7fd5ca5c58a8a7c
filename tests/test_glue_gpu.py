"""HIP glue kernels of the OSS block against plain PyTorch fp32 references of the same ops:
NCHW LayerNorm (+ fused silu gate) and the four-direction cross-merge (bit-exact)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from vmambair_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ln_ref(x, w, b, gate):
    """the reference's LayerNorm (MambaSISR6_arch.py:144-195): rearrange, mean/var(unbiased=False), eps 1e-5"""
    xf = x.float().permute(0, 2, 3, 1)
    if b is not None:
        mu = xf.mean(-1, keepdim=True)
        sig = xf.var(-1, keepdim=True, unbiased=False)
        y = (xf - mu) / torch.sqrt(sig + 1e-5) * w + b
    else:
        sig = xf.var(-1, keepdim=True, unbiased=False)
        y = xf / torch.sqrt(sig + 1e-5) * w
    y = y.permute(0, 3, 1, 2)
    return y if gate is None else y * F.silu(gate.float())


@pytest.mark.parametrize("shape", [(2, 48, 16, 24), (1, 96, 64, 64), (3, 384, 8, 8), (2, 7, 5, 3), (4, 48, 1, 1),
                                   (2, 768, 8, 8), (1, 400, 3, 5), (2, 192, 9, 11), (1, 100, 16, 16), (2, 24, 8, 8)])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)],
                         ids=["f32-f32", "f32-bf16", "bf16-bf16"])
@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("with_gate", [False, True])
def test_layernorm_nchw(shape, xdt, ydt, with_bias, with_gate):
    torch.manual_seed(0)
    B, C, H, W = shape
    x = (torch.randn(shape) * 2 + 0.5).to(xdt)
    w = torch.randn(C) * 0.5 + 1
    b = torch.randn(C) if with_bias else None
    gate = torch.randn(shape).to(ydt) if with_gate else None
    dy = torch.randn(shape).to(ydt)
    # reference in fp32 on the CPU
    xr = x.float().clone().requires_grad_()
    wr = w.clone().requires_grad_()
    br = b.clone().requires_grad_() if with_bias else None
    gr = gate.float().clone().requires_grad_() if with_gate else None
    yr = ln_ref(xr, wr, br, gr)
    yr.backward(dy.float())
    xd = x.detach().to(DEV).requires_grad_()
    wd = w.detach().to(DEV).requires_grad_()
    bd = b.detach().to(DEV).requires_grad_() if with_bias else None
    gd = gate.detach().to(DEV).requires_grad_() if with_gate else None
    y = ops.layer_norm_nchw(xd, wd, bd, gd, ydt)
    assert y.dtype == ydt and y.shape == x.shape
    y.backward(dy.to(DEV))
    lo = ydt == torch.float32 and xdt == torch.float32
    rt, at = (1e-4, 1e-4) if lo else (2e-2, 3e-2)
    assert_close(y, yr, rt, at, "y")
    assert_close(xd.grad, xr.grad, 1e-3 if lo else 3e-2, 1e-3 if lo else 5e-2, "dx")
    sc = max(1.0, float(wr.grad.abs().max()))
    assert_close(wd.grad, wr.grad, 1e-3 if lo else 3e-2, (1e-4 if lo else 2e-2) * sc, "dw")
    if with_bias:
        sc = max(1.0, float(br.grad.abs().max()))
        assert_close(bd.grad, br.grad, 1e-3 if lo else 3e-2, (1e-4 if lo else 2e-2) * sc, "db")
    if with_gate:
        assert_close(gd.grad, gr.grad, 1e-3 if lo else 3e-2, 1e-3 if lo else 5e-2, "dgate")


def test_layernorm_on_channel_strided_views():
    torch.manual_seed(1)
    big = torch.randn(2, 64, 8, 8, device=DEV)
    x = big[:, 16:48]
    gate_big = torch.randn(2, 64, 8, 8, device=DEV)
    gate = gate_big.chunk(2, dim=1)[1]
    w, b = torch.randn(32, device=DEV), torch.randn(32, device=DEV)
    y = ops.layer_norm_nchw(x, w, b, gate, torch.float32)
    assert_close(y, ln_ref(x.cpu(), w.cpu(), b.cpu(), gate.cpu()), 1e-4, 1e-4, "strided")


@pytest.mark.parametrize("shape", [(2, 48, 64, 64), (1, 5, 3, 7), (2, 8, 33, 70), (1, 96, 16, 8), (1, 3, 128, 128), (2, 4, 66, 130), (1, 2, 190, 64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_merge4_bit_exact(shape, dt):
    """((y0 + flip y2) + T y1) + T flip y3 of the reference (MambaSISR6_arch.py:427-430) on the omni
    layout: identical bits to the torch expression."""
    torch.manual_seed(2)
    B, D, H, W = shape
    out = torch.randn(B, 4, D, H * W).to(dt).to(DEV)
    y = ops.merge4(out, H, W)
    o = out.float()
    ref = o[:, 0] + o[:, 2]
    ref = ref + o[:, 1].reshape(B, D, W, H).transpose(2, 3).reshape(B, D, H * W)
    ref = ref + o[:, 3].reshape(B, D, W, H).transpose(2, 3).reshape(B, D, H * W)
    assert torch.equal(y, ref.view(B, D, H, W))


def test_fused_scan_merge_node_equals_separate_ops():
    from vmambair_amd.oss_block import SS2D_1
    torch.manual_seed(0)
    m = SS2D_1(d_model=48, ssm_ratio=1, variant="srgan").to(DEV)
    x = torch.randn(2, 48, 16, 24, device=DEV)
    gate = torch.randn(2, 48, 16, 24, device=DEV)
    res = []
    for fused in (True, False):
        m.fused_merge = fused
        m.zero_grad()
        xi = x.clone().requires_grad_()
        y = m.forward_core(xi, gate=gate)
        y.square().sum().backward()
        res.append((y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0])  # forward: same kernels, same order
    assert_close(res[0][1], res[1][1], 1e-4, 1e-4, "dx")
    for k in res[1][2]:
        assert_close(res[0][2][k], res[1][2][k], 1e-3, 1e-4 * max(1.0, float(res[1][2][k].abs().max())), k)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 96, 192, 64, 64), (1, 48, 96, 16, 24), (2, 127, 48, 8, 8), (1, 96, 254, 12, 10),
                                            (3, 5, 7, 3, 5), (1, 255, 96, 16, 16), (2, 384, 384, 8, 8),
                                            # round 2: weight tiles staged through LDS (K % 8 == 0) with a ragged last row tile and
                                            # fewer chunks than slots; several row tiles per wave (prefetch); K = 128 / 192 widths;
                                            # weight-gradient slabs: full (512 pixels), rolling window, ragged tail
                                            (1, 80, 100, 16, 16), (1, 40, 33, 8, 16), (2, 192, 96, 32, 32), (1, 128, 64, 16, 32),
                                            (1, 96, 510, 64, 64), (1, 48, 192, 32, 48), (1, 96, 97, 24, 24),
                                            # round 3: the workgroup-level kernel (K % 16 == 0, H W % 128 == 0) with ragged row tiles,
                                            # one / two / twelve k-steps, fewer row tiles than waves
                                            (1, 96, 97, 16, 16), (2, 32, 40, 16, 8), (1, 16, 8, 8, 16), (1, 192, 33, 16, 8), (2, 64, 255, 16, 16),
                                            # round 6: unaligned weight rows (K % 4 != 0) with an odd number of rows in the ragged last row tile --
                                            # the tile's block of rows * K floats ends in a PARTIAL quad (the staging stored it unshifted: the last
                                            # 1 - 3 weights of the matrix's last row were wrong; tools/conv_wave_check.py)
                                            (1, 193, 33, 16, 16), (1, 255, 31, 8, 8), (1, 127, 97, 8, 16), (1, 510, 65, 8, 8), (1, 201, 33, 8, 8)])
@pytest.mark.parametrize("has_bias", [True, False])
def test_conv1x1_mfma(dt, B, Cin, Cout, H, W, has_bias):
    """MFMA 1x1 convolution (fwd, input grad, weight grad) against F.conv2d in fp32 on the same
    16-bit-rounded activations (MambaSISR6_arch.py:205,211,281,329)."""
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W).to(dt)
    w = torch.randn(Cout, Cin, 1, 1) * (Cin ** -0.5)
    b = torch.randn(Cout) if has_bias else None
    dy = torch.randn(B, Cout, H, W).to(dt)
    xr = x.float().clone().requires_grad_()
    wr = w.to(dt).float().clone().requires_grad_()   # the kernel rounds the master weights to the I/O type
    br = b.clone().requires_grad_() if has_bias else None
    yr = F.conv2d(xr, wr, br)
    yr.backward(dy.float())
    xd = x.detach().to(DEV).requires_grad_()
    wd = w.detach().to(DEV).requires_grad_()
    bd = b.detach().to(DEV).requires_grad_() if has_bias else None
    y = ops.Conv1x1Fn.apply(xd, wd, bd)
    assert y.dtype == dt and y.shape == (B, Cout, H, W)
    y.backward(dy.to(DEV))
    rt, at = (2e-2, 3e-2) if dt == torch.bfloat16 else (3e-3, 5e-3)
    assert_close(y, yr, rt, at, "y")
    assert_close(xd.grad, xr.grad, rt, at * 2, "dx")
    sw = float(wr.grad.abs().max())
    assert_close(wd.grad, wr.grad, rt, 2e-3 * sw, "dw")  # fp32 accumulation of exact 16-bit products
    if has_bias:
        assert_close(bd.grad, br.grad, 1e-3, 1e-3 * float(br.grad.abs().max()), "db")


@pytest.mark.parametrize("mode", [12, 21, 22])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 96, 192, 64, 64), (1, 48, 96, 16, 24), (2, 127, 48, 8, 8), (1, 96, 254, 12, 10),
                                            (3, 5, 7, 3, 5), (2, 48, 48, 32, 32)])
def test_conv1x1_wgrad_tiles_per_wave(mode, B, Cin, Cout, H, W):
    """several 32 x 32 MFMA tiles of dW per wave: same weight / bias gradients as the one-tile kernel up to summation order"""
    torch.manual_seed(0)
    dt = torch.bfloat16
    x = torch.randn(B, Cin, H, W, device=DEV).to(dt)
    w = torch.randn(Cout, Cin, 1, 1, device=DEV) * (Cin ** -0.5)
    b = torch.randn(Cout, device=DEV)
    dy = torch.randn(B, Cout, H, W, device=DEV).to(dt)
    lib = ops._capi.load()
    res = []
    for m in (0, mode):
        lib.oss_conv1x1_wgrad_set_tile(m)
        try:
            wd, bd = w.clone().requires_grad_(), b.clone().requires_grad_()
            ops.Conv1x1Fn.apply(x, wd, bd).backward(dy)
            res.append((wd.grad.clone(), bd.grad.clone()))
        finally:
            lib.oss_conv1x1_wgrad_set_tile(0)
    for got, want, name in zip(res[1], res[0], ("dw", "db")):
        assert_close(got, want, 1e-5, 1e-5 * max(1.0, float(want.abs().max())), name)


def test_conv1x1_on_strided_views_and_autocast(monkeypatch):
    monkeypatch.setattr(ops.pointwise, "CONV1X1_IMPL", "mfma")
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(32, 48, 1).to(DEV)
    big = torch.randn(2, 64, 16, 16, device=DEV)
    x = big[:, 16:48]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.conv1x1(x, conv)
        yref = conv(x)
    assert y.dtype == torch.bfloat16
    assert_close(y, yref, 2e-2, 3e-2, "autocast conv1x1 vs vendor conv")
    y32 = ops.conv1x1(x, conv)  # fp32 activations outside autocast: vendor path, fp32 out
    assert y32.dtype == torch.float32


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 254, 16, 16), (1, 6, 3, 5), (3, 16, 8, 8), (2, 510, 4, 4)])
def test_gelu_gate(dt, shape):
    """gelu(x1) * x2 of the EFFN (MambaSISR6_arch.py:213-217) against F.gelu in fp32, forward and backward"""
    torch.manual_seed(0)
    B, C2, H, W = shape
    h = (torch.randn(shape) * 2).to(dt)
    dy = torch.randn(B, C2 // 2, H, W).to(dt)
    hr = h.float().clone().requires_grad_()
    x1, x2 = hr.chunk(2, dim=1)
    yr = F.gelu(x1) * x2
    yr.backward(dy.float())
    hd = h.detach().to(DEV).requires_grad_()
    y = ops.gelu_gate(hd)
    y.backward(dy.to(DEV))
    lo = dt == torch.float32
    assert_close(y, yr, 1e-5 if lo else 1e-2, 1e-5 if lo else 2e-2, "y")
    assert_close(hd.grad, hr.grad, 1e-4 if lo else 1e-2, 1e-5 if lo else 3e-2, "dh")


def test_gelu_gate_on_batch_strided_input():
    torch.manual_seed(1)
    big = torch.randn(2, 3, 8, 4, 4, device=DEV)
    h = big[:, 1]                                   # per-batch block contiguous, batch stride 3x larger
    x1, x2 = h.chunk(2, dim=1)
    assert_close(ops.gelu_gate(h), F.gelu(x1) * x2, 1e-5, 1e-5, "strided")


@pytest.mark.parametrize("shape", [(2, 48, 16, 16), (1, 96, 7, 5), (2, 400, 4, 4), (1, 96, 32, 32), (2, 100, 16, 16), (1, 192, 16, 16)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_layernorm_passthrough_adds_the_skip_gradient(shape, dt):
    """x + f(norm(x)): the skip connection's gradient enters the LayerNorm node and is added inside its kernel"""
    torch.manual_seed(3)
    B, C, H, W = shape
    x = torch.randn(shape).to(dt)
    w, b = torch.randn(C) * 0.5 + 1, torch.randn(C)
    dy, ds = torch.randn(shape).to(dt), torch.randn(shape).to(dt)
    xr = x.float().clone().requires_grad_()
    (ln_ref(xr, w, b, None) * dy.float()).sum().backward()
    want = xr.grad + ds.float()
    xd = x.detach().to(DEV).requires_grad_()
    y, alias = ops.layer_norm_nchw(xd, w.to(DEV), b.to(DEV), None, dt, passthrough=True)
    assert alias.data_ptr() == xd.data_ptr()
    ((y.float() * dy.to(DEV).float()).sum() + (alias.float() * ds.to(DEV).float()).sum()).backward()
    lo = dt == torch.float32
    assert_close(xd.grad, want, 1e-3 if lo else 3e-2, 1e-3 if lo else 6e-2, "dx + skip")


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 96, 48, 16, 16), (1, 127, 48, 8, 8), (2, 5, 7, 3, 5), (1, 384, 96, 4, 4)])
def test_conv1x1_with_fused_residual(B, Cin, Cout, H, W):
    torch.manual_seed(4)
    dt = torch.bfloat16
    x, res = torch.randn(B, Cin, H, W).to(dt), torch.randn(B, Cout, H, W).to(dt)
    w, b = torch.randn(Cout, Cin, 1, 1) * Cin ** -0.5, torch.randn(Cout)
    dy = torch.randn(B, Cout, H, W).to(dt)
    want = F.conv2d(x.float(), w.to(dt).float(), b) + res.float()
    xd, rd = x.to(DEV).requires_grad_(), res.to(DEV).requires_grad_()
    y = ops.Conv1x1Fn.apply(xd, w.to(DEV), b.to(DEV), rd)
    assert_close(y, want, 2e-2, 3e-2, "conv + residual")
    y.backward(dy.to(DEV))
    assert torch.equal(rd.grad.cpu(), dy), "the residual's gradient is dy itself"


@pytest.mark.parametrize("shape,gated,xdt", [((2, 96, 64, 64), True, torch.float32), ((3, 48, 16, 24), True, torch.float32),
                                            ((2, 192, 8, 8), True, torch.float32), ((2, 96, 32, 32), False, torch.bfloat16),
                                            ((1, 600, 4, 6), True, torch.float32)])
@pytest.mark.parametrize("with_mul", [True, False])
def test_layernorm_backward_with_the_channel_gate_folded_into_its_load(shape, gated, xdt, with_mul):
    """oss_ln_nchw_bwd_affine: dy * (1 + mul[b, c]) + s * add[b, c] formed on load == the same backward given the materialised
    gradient (what oss_row_affine used to write), for the register-resident and the streaming (C = 600) kernels"""
    torch.manual_seed(21)
    B, C, H, W = shape
    x = torch.randn(shape, device=DEV).to(xdt)
    w, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    gate = torch.randn(shape, device=DEV).to(torch.bfloat16) if gated else None
    dy = torch.randn(shape, device=DEV).to(torch.bfloat16)
    mul = torch.randn(B, C, device=DEV) * 0.3 if with_mul else None
    add = torch.randn(B, C, device=DEV)
    scale = 1.0 / (H * W)
    y, mean, rstd = ops.ln_nchw_fwd(x, w, b, gate, 2)
    dy_eff = dy.float() * ((1.0 + mul)[:, :, None, None] if with_mul else 1.0) + scale * add[:, :, None, None]
    ref = ops.ln_nchw_bwd(x, w, b, gate, dy_eff, mean, rstd) if not gated and xdt == torch.bfloat16 else None
    # reference: plain PyTorch fp32 autograd of LN (* silu(gate)) with the effective gradient
    xr, wr, br = x.float().clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    gr_ = gate.float().clone().requires_grad_() if gated else None
    mu = xr.mean(1, keepdim=True)
    var = xr.var(1, keepdim=True, unbiased=False)
    yr = (xr - mu) * (var + 1e-5).rsqrt() * wr.view(1, -1, 1, 1) + br.view(1, -1, 1, 1)
    if gated:
        yr = yr * torch.nn.functional.silu(gr_)
    yr.backward(dy_eff)
    dx, dgate, dw, db = torch.ops.vmambair.ln_nchw_bwd(x, w, b, gate, dy, mean, rstd, None, None, mul, add, scale)
    lo = xdt == torch.float32
    assert_close(dx, xr.grad, 2e-3 if lo else 2e-2, 2e-3 if lo else 4e-2, "dx")
    assert_close(dw, wr.grad, 2e-3, 2e-3 * float(wr.grad.abs().max()), "dw")
    assert_close(db, br.grad, 2e-3, 2e-3 * float(br.grad.abs().max()), "db")
    if gated:
        assert_close(dgate, gr_.grad, 2e-2, 4e-2, "dgate")
    assert ref is None or ref[0].shape == dx.shape


def test_norm_channel_gate_node_matches_the_two_separate_nodes():
    """NormChannelGateFn (out_norm * silu(z) -> channel branch -> gate, d y2 never materialised) against LayerNormNCHWFn followed by
    ChannelGateFn on the same tensors: every gradient to bf16 round-off of the one tensor (d y2) the fused node does not round"""
    from vmambair_amd import oss_block
    torch.manual_seed(22)
    m = oss_block.SS2D_1(d_model=48, ssm_ratio=1, variant="srgan").to(DEV)
    x = torch.randn(2, 48, 32, 32, device=DEV)
    gy = torch.randn(2, 48, 32, 32, device=DEV)
    res = []
    for fused in (True, False):
        oss_block.NORM_CHAN_FUSED = fused
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        y.backward(gy)
        res.append((y.detach().float(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    oss_block.NORM_CHAN_FUSED = True
    # (the one-node form pools y2 from the LayerNorm's per-workgroup sums: c agrees to fp32 summation order, the output to a rounding)
    assert_close(res[0][0], res[1][0], 1e-2, 1e-2 * float(res[1][0].abs().max()), "y")
    assert_close(res[0][1], res[1][1], 2e-2, 2e-2 * float(res[1][1].abs().max()), "dx")
    for k, g in res[1][2].items():
        assert_close(res[0][2][k], g, 3e-2, 3e-2 * max(float(g.abs().max()), 1e-6), k)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 96, 192, 32, 32), (1, 96, 510, 16, 16), (2, 48, 254, 16, 8), (1, 192, 96, 16, 16),
                                            (1, 16, 40, 8, 16), (2, 80, 33, 16, 8), (1, 176, 97, 8, 32)])
def test_conv1x1_workgroup_level_kernel_is_bit_identical_to_the_wave_level_ones(dt, B, Cin, Cout, H, W):
    """oss_conv1x1_wg.hip (LDS-resident activation tile, ds_read_b64_tr_b16 fragments, 64- and 128-pixel workgroups) through the
    same entry points as the wave-level kernels of oss_conv1x1.hip: forward (+ bias) and input gradient, torch.equal"""
    from vmambair_amd import _capi
    lib = _capi.load()
    torch.manual_seed(3)
    x = torch.randn(B, Cin, H, W, device=DEV).to(dt)
    w = torch.randn(Cout, Cin, 1, 1, device=DEV) * (Cin ** -0.5)
    b = torch.randn(Cout, device=DEV)
    dy = torch.randn(B, Cout, H, W, device=DEV).to(dt)
    outs = []
    try:
        for on, pix in ((0, 0), (1, 128), (1, 64), (1, 0)):
            lib.oss_conv1x1_set_wg(on, pix)
            y = ops.conv1x1_fwd(x, w, b)
            dx = ops.conv1x1_bwd(x, w, dy, True)[0]
            torch.cuda.synchronize()
            outs.append((y.clone(), dx.clone()))
    finally:
        lib.oss_conv1x1_set_wg(1, 0)
    for y, dx in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(dx, outs[0][1])
    assert_close(outs[0][0], F.conv2d(x.float(), w.to(dt).float(), b), 2e-2 if dt == torch.bfloat16 else 3e-3, 3e-2, "y vs torch")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ln_bias", [(2, 96, 192, 32, 32, True), (1, 48, 254, 16, 16, True), (2, 96, 510, 16, 8, False),
                                                     (1, 192, 384, 16, 8, True), (2, 48, 96, 64, 64, True), (1, 16, 33, 8, 16, False),
                                                     (4, 96, 510, 64, 64, True), (8, 48, 254, 32, 64, False)])
def test_layernorm_fused_into_the_1x1_convolution(dt, B, Cin, Cout, H, W, ln_bias):
    """LNConv1x1Fn (norm1 -> in_conv, norm2 -> project_in in one forward launch: the LayerNorm runs on the convolution's LDS-resident
    activation tile) against plain PyTorch fp32 and against the two separate nodes on the same tensors, incl. the skip connection's
    gradient through the alias output and both LayerNorm forms (WithBias / BiasFree)"""
    torch.manual_seed(5)
    x = (torch.randn(B, Cin, H, W, device=DEV) * 1.5 + 0.3).to(dt)
    lw = torch.randn(Cin, device=DEV) * 0.2 + 1.0
    lb = torch.randn(Cin, device=DEV) * 0.1 if ln_bias else None
    w = torch.randn(Cout, Cin, 1, 1, device=DEV) * (Cin ** -0.5)
    b = torch.randn(Cout, device=DEV)
    dy = torch.randn(B, Cout, H, W, device=DEV).to(dt)
    dskip = torch.randn(B, Cin, H, W, device=DEV).to(dt)
    conv = torch.nn.Conv2d(Cin, Cout, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
    assert ops.ln_conv1x1_ok(x, conv.weight)
    res = []
    for fused in (True, False):
        conv.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        lwi = lw.clone().requires_grad_()
        lbi = lb.clone().requires_grad_() if ln_bias else None
        if fused:
            y, skip = ops.ln_conv1x1(xi, lwi, lbi, conv)
        else:
            n, skip = ops.layer_norm_nchw(xi, lwi, lbi, None, dt, True)
            y = ops.conv1x1(n, conv)
        torch.autograd.backward([y, skip], [dy, dskip])
        res.append((y.detach().float(), xi.grad.float(), lwi.grad.clone(), lbi.grad.clone() if ln_bias else None,
                    conv.weight.grad.clone(), conv.bias.grad.clone()))
    # plain PyTorch fp32 on the same 16-bit inputs
    xr = x.float().clone().requires_grad_()
    lwr, wr, br = lw.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    lbr = lb.clone().requires_grad_() if ln_bias else None
    mu = xr.mean(1, keepdim=True)
    rs = (xr.var(1, keepdim=True, unbiased=False) + 1e-5).rsqrt()
    n = (xr - mu) * rs * lwr.view(1, -1, 1, 1) + lbr.view(1, -1, 1, 1) if ln_bias else xr * rs * lwr.view(1, -1, 1, 1)
    yr = F.conv2d(n, wr, br)
    torch.autograd.backward([yr, xr * 1.0], [dy.float(), dskip.float()])
    rt = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert_close(res[0][0], yr, 2 * rt, 4 * rt, "y vs torch")
    assert_close(res[0][1], xr.grad, 2 * rt, 4 * rt * float(xr.grad.abs().max()), "dx vs torch")
    assert_close(res[0][4], wr.grad, 2 * rt, 2 * rt * float(wr.grad.abs().max()), "dW vs torch")
    # the two HIP paths: the same to the rounding of the normalised activations (summation order of the statistics differs)
    assert_close(res[0][0], res[1][0], rt, 2 * rt, "y")
    assert_close(res[0][1], res[1][1], rt, 2 * rt * float(res[1][1].abs().max()), "dx")
    assert_close(res[0][2], res[1][2], rt, rt * float(res[1][2].abs().max()), "d ln weight")
    if ln_bias:
        assert_close(res[0][3], res[1][3], rt, rt * float(res[1][3].abs().max()), "d ln bias")
    assert_close(res[0][4], res[1][4], rt, rt * float(res[1][4].abs().max()), "dW")
    assert_close(res[0][5], res[1][5], 1e-4, 1e-4 * float(res[1][5].abs().max()), "db")


@pytest.mark.parametrize("shape,xdt", [((2, 96, 64, 64), torch.float32), ((1, 48, 16, 24), torch.float32), ((2, 192, 16, 16), torch.bfloat16),
                                       ((1, 600, 8, 16), torch.float32)])
def test_layernorm_forward_leaves_the_pooled_sums_of_its_output(shape, xdt):
    """oss_ln_nchw_fwd_pool: per 128-pixel workgroup and channel, the sum of the output values as stored -- what mean_hw(y2) of the
    channel branch (MambaSISR6_arch.py:438-441) is made of; the output itself is unchanged"""
    torch.manual_seed(31)
    B, C, H, W = shape
    x = torch.randn(shape, device=DEV).to(xdt)
    w, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    gate = torch.randn(shape, device=DEV).to(torch.bfloat16)
    y0, m0, r0 = torch.ops.vmambair.ln_nchw_fwd(x, w, b, gate, 2, False)
    y, m, r, pool = torch.ops.vmambair.ln_nchw_fwd(x, w, b, gate, 2, True)
    assert torch.equal(y, y0) and torch.equal(m, m0) and torch.equal(r, r0)
    if C > 384:   # the streaming kernel (channels not register-resident) does not take the form: an empty tensor says so
        assert pool.numel() == 0
        return
    assert pool.shape == (B, (H * W + 127) // 128, C)
    want = y.float().reshape(B, C, H * W).sum(-1)
    assert_close(pool.sum(1), want, 1e-5, 1e-4 * float(want.abs().max()), "pooled sums")
    tile0 = y.float().reshape(B, C, H * W)[:, :, :128].sum(-1)
    assert_close(pool[:, 0], tile0, 1e-5, 1e-4 * float(tile0.abs().max()), "first tile")


def test_channel_branch_from_the_pooled_sums_matches_the_pooling_pass():
    """chan_gate_fwd given the LayerNorm's per-workgroup sums (no oss_rowsum launch) against the same call with its own pooling pass:
    pooled and the gate vector agree to fp32 summation order, the gated output to one rounding"""
    from vmambair_amd import oss_block
    torch.manual_seed(32)
    m = oss_block.SS2D_1(d_model=96, ssm_ratio=1, variant="srgan").to(DEV)
    y = torch.randn(2, 96, 32, 32, device=DEV)
    z = torch.randn(2, 96, 32, 32, device=DEV).to(torch.bfloat16)
    y2, _, _, pool = torch.ops.vmambair.ln_nchw_fwd(y, m.out_norm.body.weight, m.out_norm.body.bias, z, 2, True)
    args = (m.conv_cin.weight, m.conv_cin.bias, m.xc_proj_weight, m.dtc_projs_weight, m.dtc_projs_bias, m.Ac_logs, m.Dsc,
            m.conv_cout.weight, m.conv_cout.bias, m.channel_norm.body.weight, m.channel_norm.body.bias, True)
    args = tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in args)
    a = torch.ops.vmambair.chan_gate_fwd(y2, *args, pool)
    bb = torch.ops.vmambair.chan_gate_fwd(y2, *args, None)
    assert_close(a[2], bb[2], 1e-5, 1e-6, "pooled")
    assert_close(a[1], bb[1], 1e-4, 1e-5, "c")
    assert_close(a[0], bb[0].float(), 1e-2, 1e-2, "out")


@pytest.mark.parametrize("variant,d,H,W", [("realsr", 48, 256, 256), ("realsr", 96, 128, 192), ("srgan", 96, 272, 272), ("realsr", 384, 64, 64)],
                         ids=["realsr-512tiles", "realsr-192tiles", "srgan-578tiles", "L768-more-channels-than-threads"])
def test_channel_branch_pools_many_tiles(variant, d, H, W):
    """round 6: the pooled descriptor from MANY per-workgroup sums (an untiled RealSR plane leaves 1024 per channel): up to 8 threads per
    channel add contiguous runs of the tiles, then the runs in order -- against the pooling pass and a float64 sum of the tiles"""
    from vmambair_amd import oss_block
    torch.manual_seed(33)
    m = oss_block.SS2D_1(d_model=d, ssm_ratio=2, variant=variant).to(DEV)
    D = m.d_inner
    y = torch.randn(1, D, H, W, device=DEV)
    z = torch.randn(1, D, H, W, device=DEV).to(torch.float16)
    y2, _, _, pool = torch.ops.vmambair.ln_nchw_fwd(y, m.out_norm.body.weight, m.out_norm.body.bias, z, 1, True)
    if pool.numel() == 0:   # more channels than this LayerNorm form pools: the tile sums by hand (any split of the pixels is a valid input)
        pool = y2.float().reshape(1, D, -1, 128).sum(-1).permute(0, 2, 1).contiguous()
    assert pool.shape[1] >= 32 and (pool.shape[1] >= 128 or D > 512)
    lift = m.dc_inner is not None
    if True:
        args = ((m.conv_cin.weight, m.conv_cin.bias) if lift else (None, None)) + (m.xc_proj_weight, m.dtc_projs_weight, m.dtc_projs_bias, m.Ac_logs, m.Dsc) + \
            ((m.conv_cout.weight, m.conv_cout.bias) if lift else (None, None)) + (m.channel_norm.body.weight, m.channel_norm.body.bias, True)
    args = tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in args)
    a = torch.ops.vmambair.chan_gate_fwd(y2, *args, pool)
    bb = torch.ops.vmambair.chan_gate_fwd(y2, *args, None)
    want = pool.double().sum(1) / (H * W)
    assert_close(a[2], want.float(), 1e-5, 1e-6, "pooled vs float64 sum of the tiles")
    assert_close(a[2], bb[2], 1e-4, 2e-6, "pooled")
    assert_close(a[1], bb[1], 1e-3, 1e-4, "c")
    assert_close(a[0], bb[0].float(), 1e-2, 1e-2, "out")


# ---- round 4: fp32 I/O on v_mfma_f32_32x32x2_f32 (csrc/oss_conv1x1_f32.hip) -- the reference's own precision -------------------
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 96, 384, 64, 64), (1, 48, 96, 16, 24), (2, 127, 48, 8, 8), (1, 96, 254, 12, 10),
                                            (3, 5, 7, 2, 6), (1, 255, 96, 16, 16), (2, 384, 384, 8, 8), (1, 96, 510, 32, 32),
                                            (1, 33, 65, 4, 5), (2, 192, 96, 32, 32)])
@pytest.mark.parametrize("has_bias", [True, False])
def test_conv1x1_fp32_matrix_core_kernels(B, Cin, Cout, H, W, has_bias):
    """fp32 activations, fp32 weights, true fp32 products: forward, input gradient and weight gradient against F.conv2d with TF32
    off -- odd channel counts (127 / 255 / 33: a half-empty last k-step), ragged row tiles, pixel counts that are not a multiple of
    the 128-pixel tile or of the 512-pixel weight-gradient slab, a skip connection in the epilogue"""
    torch.manual_seed(1)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 1, 1) * (Cin ** -0.5)
    b = torch.randn(Cout) if has_bias else None
    dy = torch.randn(B, Cout, H, W)
    res = torch.randn(B, Cout, H, W)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if has_bias else None
    yr = F.conv2d(xr.double(), wr.double(), None if br is None else br.double()) + res.double()
    yr.backward(dy.double())
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if has_bias else None
    assert ops.pointwise.f32_ok(xd)
    y = ops.Conv1x1Fn.apply(xd, wd, bd, res.to(DEV))
    assert y.dtype == torch.float32
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    assert_close(y, yr.float(), 2e-5, 2e-5 * float(yr.abs().max()), "y")
    assert_close(xd.grad, xr.grad, 2e-5, 2e-5 * float(xr.grad.abs().max()), "dx")
    assert_close(wd.grad, wr.grad, 1e-4, 2e-5 * float(wr.grad.abs().max()), "dw")
    if has_bias:
        assert_close(bd.grad, br.grad, 1e-4, 2e-5 * float(br.grad.abs().max()), "db")


def test_conv1x1_fp32_route_and_fallbacks(monkeypatch):
    """``conv1x1()`` sends fp32 activations (no autocast) to the matrix-core kernels, keeps the vendor convolution for pixel counts
    that are not a multiple of 4 and under VMAMBAIR_CONV1X1_F32=0; channel-slice views are taken as they are"""
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(32, 48, 1, bias=False).to(DEV)
    big = torch.randn(2, 64, 16, 16, device=DEV)
    x = big[:, 16:48].requires_grad_()
    y = ops.conv1x1(x, conv)
    assert y.grad_fn.__class__.__name__.startswith("Conv1x1Fn")
    assert_close(y, conv(x), 2e-5, 2e-5, "fp32 view")
    odd = torch.randn(1, 32, 3, 5, device=DEV)
    assert not ops.conv1x1(odd, conv).grad_fn.__class__.__name__.startswith("Conv1x1Fn")
    monkeypatch.setattr(ops.pointwise, "CONV1X1_F32", False)
    assert not ops.conv1x1(x, conv).grad_fn.__class__.__name__.startswith("Conv1x1Fn")


def test_family_byte_counters_follow_the_entry_points():
    """(round 6) ``oss_prof_family*`` (include/vmambair_oss.h): the algorithmic-bytes accounting behind bench.py's
    ``roofline.non_scan`` -- counting on: a 1x1 convolution and a LayerNorm add their formula's bytes to their own family and to no
    other; counting off: nothing moves; switching it on again clears the counters."""
    import ctypes as C
    from vmambair_amd import _capi
    lib = _capi.load()

    def read():
        out = {}
        for f in range(int(lib.oss_prof_family_count())):
            name, pat, by, calls = C.c_char_p(), C.c_char_p(), C.c_double(), C.c_longlong()
            assert lib.oss_prof_family(f, C.byref(name), C.byref(pat), C.byref(by), C.byref(calls)) == 0
            out[name.value.decode()] = (by.value, calls.value, pat.value.decode())
        return out
    B, Cin, Cout, H, W = 2, 96, 192, 32, 32
    x = torch.randn(B, Cin, H, W, device=DEV).to(torch.bfloat16)
    conv = torch.nn.Conv2d(Cin, Cout, 1, bias=True).to(DEV)
    wln, bln = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    lib.oss_prof_family_enable(1)
    ops.conv1x1(x, conv)
    ops.layer_norm_nchw(x, wln, bln, None, None, False, None)
    lib.oss_prof_family_enable(0)
    ops.conv1x1(x, conv)                       # not counted
    torch.cuda.synchronize()
    got = read()
    conv_fam = next(k for k in got if k.startswith("conv1x1"))
    ln_fam = next(k for k in got if k.startswith("LayerNorm"))
    P = H * W
    assert got[conv_fam][1] == 1 and got[conv_fam][0] == B * P * 2 * (Cin + Cout) + 4 * Cin * Cout
    assert got[ln_fam][1] == 1 and got[ln_fam][0] == B * Cin * P * (2 + 2) + 8 * B * P
    assert all(v[1] == 0 for k, v in got.items() if k not in (conv_fam, ln_fam)), got
    assert "oss_conv1x1_wg_kernel" in got[conv_fam][2] and "oss_ln_nchw" in got[ln_fam][2]
    assert lib.oss_prof_family(99, None, None, None, None) != 0
    lib.oss_prof_family_enable(1)
    lib.oss_prof_family_enable(0)
    assert all(v[1] == 0 for v in read().values())
