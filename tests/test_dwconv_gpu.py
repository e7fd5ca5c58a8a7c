"""HIP depth-wise 3x3 convolution (SS2D_1.conv2d / FeedForward.dwconv of the OSS block) through the
C ABI against a plain PyTorch fp32 reference of the same op (``F.conv2d(..., padding=1, groups=C)``,
reference modules: SRGAN/VmambaIR/archs/MambaSISR6_arch.py:209,286-294)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from vmambair_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref(x, w, b, dy):
    xx = x.detach().float().cpu().requires_grad_()
    ww = w.detach().float().cpu().requires_grad_()
    bb = None if b is None else b.detach().float().cpu().requires_grad_()
    y = F.conv2d(xx, ww, bb, padding=1, groups=xx.shape[1])
    y.backward(dy.float().cpu())
    return y.detach(), xx.grad, ww.grad, None if bb is None else bb.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 48, 64, 64), (1, 254, 16, 24), (3, 5, 7, 9), (2, 96, 12, 10), (1, 3, 1, 1), (2, 8, 33, 4),
                                   # 16-bit I/O: the 8-pixels-per-lane kernels (W / 8 lanes per row = 4, 1, 16, 64, 2 with a ragged
                                   # last workgroup) next to widths they must leave to the 4-pixel kernels (24, 12, 4)
                                   (2, 16, 32, 32), (3, 8, 8, 8), (1, 4, 6, 128), (1, 2, 5, 512), (2, 5, 37, 16),
                                   # (round 4) rows that straddle waves: W / 8 = 20, 18, 40, 5
                                   (1, 6, 24, 160), (2, 3, 13, 144), (1, 2, 9, 320), (2, 4, 11, 40),
                                   # (round 5) the RealSR bench's 256 + 16 + 16 tiles and their first down-sampling: W / 8 = 34, 17
                                   (1, 3, 10, 272), (1, 2, 9, 136)])
@pytest.mark.parametrize("has_bias", [True, False])
def test_dwconv_matches_torch(dtype, shape, has_bias):
    torch.manual_seed(0)
    B, C, H, W = shape
    x = torch.randn(shape).to(dtype)
    w = torch.randn(C, 1, 3, 3) * 0.3
    b = torch.randn(C) if has_bias else None
    dy = torch.randn(shape).to(dtype)
    xd = x.to(DEV).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    bd = None if b is None else b.to(DEV).requires_grad_()
    y = ops.DWConv3x3Fn.apply(xd, wd, bd)
    y.backward(dy.to(DEV))
    ry, rdx, rdw, rdb = ref(x, w, b, dy)
    rt, at = {torch.float32: (1e-5, 1e-5), torch.bfloat16: (2e-2, 2e-2), torch.float16: (2e-3, 2e-3)}[dtype]
    assert y.dtype == dtype
    assert_close(y, ry, rt, at, "y")
    assert_close(xd.grad, rdx, rt, at, "dx")
    sw = float(rdw.abs().max())
    assert_close(wd.grad, rdw, 1e-4 if dtype == torch.float32 else 2e-2, (1e-5 if dtype == torch.float32 else 5e-3) * max(sw, 1.0), "dw")
    if has_bias:
        assert_close(bd.grad, rdb, 1e-4 if dtype == torch.float32 else 2e-2, (1e-5 if dtype == torch.float32 else 5e-3) * max(float(rdb.abs().max()), 1.0), "db")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_dwconv_on_chunk_views(dt):
    """the block feeds the conv a channel chunk of in_conv's output (batch stride 2*C*H*W)"""
    torch.manual_seed(1)
    xz = torch.randn(2, 32, 16, 16, device=DEV).to(dt)
    x = xz.chunk(2, dim=1)[1]
    assert not x.is_contiguous()
    w = torch.randn(16, 1, 3, 3, device=DEV)
    b = torch.randn(16, device=DEV)
    y, _ = ops.dwconv3x3_fwd(x, w, b)
    tol = 1e-5 if dt == torch.float32 else 2e-2
    assert_close(y, F.conv2d(x.float().cpu(), w.cpu(), b.cpu(), padding=1, groups=16).to(dt), tol, tol, "chunk view")


@pytest.mark.parametrize("shape", [(2, 48, 16, 16), (1, 6, 5, 7), (2, 96, 8, 12), (2, 12, 64, 64), (1, 3, 9, 32)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_dwconv_with_fused_silu(shape, dt):
    """act(conv2d(x)) of SS2D_1 (MambaSISR6_arch.py:486): silu in the conv epilogue, its derivative in the weight-gradient pass"""
    torch.manual_seed(7)
    B, C, H, W = shape
    x = torch.randn(shape).to(dt)
    w, b = torch.randn(C, 1, 3, 3) * 0.3, torch.randn(C) * 0.1
    dy = torch.randn(shape).to(dt)
    xr, wr, br = x.float().clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.silu(F.conv2d(xr, wr, br, padding=1, groups=C))
    yr.backward(dy.float())
    conv = torch.nn.Conv2d(C, C, 3, padding=1, groups=C).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
    xd = x.detach().to(DEV).requires_grad_()
    y = ops.dwconv3x3(xd, conv, act=True)
    y.backward(dy.to(DEV))
    lo = dt == torch.float32
    assert_close(y, yr, 1e-5 if lo else 1e-2, 1e-5 if lo else 2e-2, "y")
    assert_close(xd.grad, xr.grad, 1e-4 if lo else 2e-2, 1e-5 if lo else 4e-2, "dx")
    assert_close(conv.weight.grad, wr.grad, 1e-4 if lo else 2e-2, (1e-5 if lo else 2e-2) * float(wr.grad.abs().max()), "dw")
    assert_close(conv.bias.grad, br.grad, 1e-4 if lo else 2e-2, (1e-5 if lo else 2e-2) * float(br.grad.abs().max()), "db")


# ---- fused forms of oss_dwconv.hip: the convolution is never stored, the backward is one launch ---------------------------
# (round 4) widths whose W / 8 lane groups do not tile a wave -- 24 (3 groups), 160 (20: the RealSR tiles), 144 (18: their edge
# tiles), 192 / 320 (the Deraining tree's progressive patches) -- take the EDGE instantiations: rows straddle waves
FUSED_SHAPES = [(2, 12, 64, 64), (1, 6, 32, 32), (3, 4, 16, 16), (2, 8, 8, 8), (1, 2, 128, 128), (2, 4, 5, 512), (2, 6, 37, 16),
                (1, 254, 16, 64), (1, 4, 16, 24), (1, 6, 160, 160), (2, 4, 144, 160), (1, 4, 9, 144), (1, 2, 48, 192), (1, 2, 7, 320),
                (1, 2, 16, 272), (1, 2, 8, 136)]


@pytest.mark.parametrize("shape", FUSED_SHAPES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("has_bias", [True, False])
def test_fused_conv_silu_one_launch_backward(shape, dt, has_bias):
    """silu(conv2d(x)) (MambaSISR6_arch.py:486) through oss_dwconv3x3_silu_fwd / _bwd against plain PyTorch fp32, and against the
    separate kernels (which store the convolution): same rounding points, so the two HIP paths agree far inside the tolerance"""
    torch.manual_seed(11)
    B, C, H, W = shape
    assert ops.dwconv.fused_ok(torch.empty(shape, dtype=dt, device=DEV), 1)
    x = torch.randn(shape).to(dt)
    w, b = torch.randn(C, 1, 3, 3) * 0.3, (torch.randn(C) * 0.1 if has_bias else None)
    dy = torch.randn(shape).to(dt)
    xr, wr = x.float().clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if has_bias else None
    yr = F.silu(F.conv2d(xr, wr, br, padding=1, groups=C))
    yr.backward(dy.float())
    xd, wd = x.to(DEV), w.to(DEV)
    bd = b.to(DEV) if has_bias else None
    y = torch.ops.vmambair.dwconv3x3_silu_fwd(xd, wd, bd)
    dx, dw, db = torch.ops.vmambair.dwconv3x3_silu_bwd(xd, wd, bd, dy.to(DEV), None)
    y0, pre = torch.ops.vmambair.dwconv3x3_fwd(xd, wd, bd, True)
    dx0, dw0, db0 = torch.ops.vmambair.dwconv3x3_bwd(xd, wd, dy.to(DEV), has_bias, pre, None)
    rt = {torch.bfloat16: 1e-2, torch.float16: 2e-3, torch.float32: 1e-4}[dt]   # (round 4) float I/O: the same kernels, 32-byte rows
    assert_close(y, yr, rt, 2 * rt, "y")
    assert torch.equal(y, y0), "forward differs from the separate kernels"
    assert_close(dx, xr.grad, 2 * rt, 4 * rt, "dx")
    assert_close(dw.cpu(), wr.grad, 2 * rt, 2 * rt * float(wr.grad.abs().max()), "dw")
    assert_close(dx, dx0.float(), rt, 2 * rt, "dx vs separate")
    assert_close(dw, dw0, rt, rt * float(dw0.abs().max()), "dw vs separate")
    if has_bias:
        assert_close(db.cpu(), br.grad, 2 * rt, 2 * rt * float(br.grad.abs().max()), "db")
        assert_close(db, db0, rt, rt * float(db0.abs().max()), "db vs separate")


def test_fused_conv_silu_writes_into_a_half_buffer_and_reruns_bit_exact():
    """the SS2D_1 use: dx lands in one half of the gradient buffer of in_conv's output (ops.PairGrad); reruns are bit-identical"""
    torch.manual_seed(12)
    B, C, H, W = 2, 24, 32, 32
    xz = torch.randn(B, 2 * C, H, W, device=DEV).to(torch.bfloat16)
    x = xz[:, :C]
    w, b = torch.randn(C, 1, 3, 3, device=DEV) * 0.3, torch.randn(C, device=DEV) * 0.1
    dy = torch.randn(B, C, H, W, device=DEV).to(torch.bfloat16)
    buf = torch.zeros(B, 2 * C, H, W, device=DEV, dtype=torch.bfloat16)
    dx, dw, db = torch.ops.vmambair.dwconv3x3_silu_bwd(x, w, b, dy, buf[:, :C])
    assert dx.numel() == 0 and float(buf[:, C:].abs().max()) == 0.0
    dx2, dw2, db2 = torch.ops.vmambair.dwconv3x3_silu_bwd(x.contiguous(), w, b, dy, None)
    assert torch.equal(buf[:, :C], dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)


# float planes: only the shapes whose TWO (H + 2) x W float planes fit a workgroup's LDS (the separate kernels take the others)
_GATE_CASES = [pytest.param(sh, dt, id=f"{name}-{'x'.join(map(str, sh))}") for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16"), (torch.float32, "f32"))
               for sh in FUSED_SHAPES if not (dt == torch.float32 and (sh[2] + 2) * sh[3] * 4 * 2 > 159 * 1024)]


@pytest.mark.parametrize("shape,dt", _GATE_CASES)
@pytest.mark.parametrize("has_bias", [False, True])
def test_fused_conv_gelu_gate(shape, dt, has_bias):
    """x1, x2 = dwconv(t).chunk(2); gelu(x1) * x2 (FeedForward.forward, MambaSISR6_arch.py:213-217) as one node: against plain
    PyTorch fp32 and against the separate kernels (dwconv -> gelu_gate; three backward launches)"""
    torch.manual_seed(13)
    B, C2, H, W = shape
    Hd = C2 // 2
    assert ops.dwconv.fused_ok(torch.empty(shape, dtype=dt, device=DEV), 2)
    t = torch.randn(shape).to(dt)
    w, b = torch.randn(C2, 1, 3, 3) * 0.3, (torch.randn(C2) * 0.1 if has_bias else None)
    dout = torch.randn(B, Hd, H, W).to(dt)
    tr, wr = t.float().clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if has_bias else None
    x1, x2 = F.conv2d(tr, wr, br, padding=1, groups=C2).chunk(2, dim=1)
    outr = F.gelu(x1) * x2
    outr.backward(dout.float())
    conv = torch.nn.Conv2d(C2, C2, 3, padding=1, groups=C2, bias=has_bias).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w)
        if has_bias:
            conv.bias.copy_(b)
    td = t.to(DEV).requires_grad_()
    out = ops.dwconv3x3_gelu_gate(td, conv)
    assert out.grad_fn.__class__.__name__.startswith("DWGateFn")
    out.backward(dout.to(DEV))
    rt = {torch.bfloat16: 1e-2, torch.float16: 2e-3, torch.float32: 1e-4}[dt]
    assert_close(out, outr, rt, 2 * rt, "out")
    assert_close(td.grad, tr.grad, 2 * rt, 6 * rt, "dt")
    assert_close(conv.weight.grad.cpu(), wr.grad, 2 * rt, 3 * rt * float(wr.grad.abs().max()), "dw")
    if has_bias:
        assert_close(conv.bias.grad.cpu(), br.grad, 2 * rt, 3 * rt * float(br.grad.abs().max()), "db")
    # the separate kernels on the same tensors
    conv0 = torch.nn.Conv2d(C2, C2, 3, padding=1, groups=C2, bias=has_bias).to(DEV)
    conv0.load_state_dict(conv.state_dict())
    t0 = t.to(DEV).requires_grad_()
    out0 = ops.gelu_gate(ops.dwconv3x3(t0, conv0))
    out0.backward(dout.to(DEV))
    assert_close(out, out0.float(), rt, 2 * rt, "out vs separate")
    assert_close(td.grad, t0.grad.float(), rt, 4 * rt, "dt vs separate")
    assert_close(conv.weight.grad, conv0.weight.grad, rt, 2 * rt * float(conv0.weight.grad.abs().max()), "dw vs separate")


def test_shapes_the_fused_forms_leave_to_the_separate_kernels():
    """widths that are not a multiple of 8 pixels, planes beyond the LDS (float planes are twice the bytes): fused_ok says no and the
    nodes fall back"""
    assert ops.dwconv.fused_ok(torch.empty((1, 4, 16, 16), dtype=torch.float32, device=DEV), 1)   # float I/O: fused since round 4
    for shape, dt, planes in (((1, 4, 16, 20), torch.float32, 1), ((1, 4, 16, 20), torch.bfloat16, 1), ((1, 4, 9, 12), torch.bfloat16, 2),
                              ((1, 2, 512, 512), torch.bfloat16, 2), ((1, 2, 256, 256), torch.float32, 1)):
        assert not ops.dwconv.fused_ok(torch.empty(shape, dtype=dt, device=DEV), planes)
    conv = torch.nn.Conv2d(8, 8, 3, padding=1, groups=8, bias=False).to(DEV)
    t = torch.randn(2, 8, 9, 12, device=DEV).to(torch.bfloat16).requires_grad_()
    out = ops.dwconv3x3_gelu_gate(t, conv)
    assert out.grad_fn.__class__.__name__.startswith("GeluGateFn")
    x1, x2 = F.conv2d(t.float(), conv.weight, None, padding=1, groups=8).chunk(2, dim=1)
    assert_close(out, F.gelu(x1) * x2, 2e-2, 4e-2, "fallback")
    assert bool(ops.dwconv.fused_ok(torch.empty((1, 2, 256, 256), dtype=torch.bfloat16, device=DEV), 1))   # 129 KiB: one plane still fits


@pytest.mark.parametrize("shape,dt", [((1, 6, 272, 272), torch.float16), ((1, 4, 512, 512), torch.float16), ((2, 6, 136, 144), torch.bfloat16),
                                      ((1, 2, 300, 200), torch.float32), ((1, 4, 272, 136), torch.float16)],
                         ids=["tile256-f16", "untiled512-f16", "bf16-136x144", "f32-300x200", "f16-272x136"])
@pytest.mark.parametrize("has_bias", [False, True])
def test_gate_streams_in_inference_on_planes_beyond_the_lds(shape, dt, has_bias):
    """round 6: without a backward to prepare, ``dwconv3x3_gelu_gate`` takes the one-launch streaming forward (``oss_dwgate_fwd``) on every
    plane ``oss_dwgate_fwd_ok`` accepts -- the RealSR tiles of 272 x 272 and the untiled 512 x 512 plane, whose two planes do not fit the
    backward's LDS-resident form (they ran dwconv -> gelu_gate: 7 plane passes instead of 3).  Against plain PyTorch fp32 and against
    the two separate kernels; with gradients enabled the same call still builds the two separate nodes"""
    torch.manual_seed(17)
    B, C2, H, W = shape
    conv = torch.nn.Conv2d(C2, C2, 3, padding=1, groups=C2, bias=has_bias).to(DEV)
    t = torch.randn(shape, device=DEV).to(dt)
    assert ops.dwconv.gate_fwd_ok(t)
    big = not ops.dwconv.fused_ok(t, 2)
    with torch.no_grad():
        out = ops.dwconv3x3_gelu_gate(t, conv)
        sep = ops.gelu_gate(ops.dwconv3x3(t, conv))
        one = ops.dwconv.dwgate_fwd(t, conv.weight, conv.bias)
    assert torch.equal(out, one), "the inference call is the one-launch form (the separate kernels round the convolution in between)"
    x1, x2 = F.conv2d(t.float(), conv.weight, conv.bias, padding=1, groups=C2).chunk(2, dim=1)
    want = F.gelu(x1) * x2
    rt = {torch.bfloat16: 1e-2, torch.float16: 2e-3, torch.float32: 1e-4}[dt]
    assert_close(out, want, rt, 2 * rt, "out")
    assert_close(out, sep.float(), rt, 2 * rt, "out vs separate")
    if big:
        tg = t.clone().requires_grad_()
        assert ops.dwconv3x3_gelu_gate(tg, conv).grad_fn.__class__.__name__.startswith("GeluGateFn")
    assert not ops.dwconv.gate_fwd_ok(torch.empty((1, 2, 16, 20), dtype=torch.float16, device=DEV))
    assert not ops.dwconv.gate_fwd_ok(torch.empty((1, 2, 16, 520), dtype=torch.float16, device=DEV))


# (round 4) SS2D_1's convolution together with cross_scan_2d's two forward flattenings (MambaSISR6_arch.py:399-404, 486)
FLAT2_SHAPES = [(2, 96, 64, 64), (1, 192, 32, 32), (2, 384, 16, 16), (1, 768, 8, 8), (1, 48, 128, 128), (1, 6, 24, 64), (2, 5, 8, 256),
                (1, 3, 40, 8)]


@pytest.mark.parametrize("shape", FLAT2_SHAPES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("has_bias", [True, False])
def test_flat2_forms_are_bit_identical_to_the_separate_launches(shape, dt, has_bias):
    """x2 = [silu(conv(x)) | its transpose] from ONE launch == dwconv3x3_silu_fwd + cross_scan2; the backward that reads the two
    flattenings' gradients == cross_merge2 + dwconv3x3_silu_bwd.  Bit for bit: the merged gradient is rounded where the merge
    launch stored it."""
    torch.manual_seed(1)
    B, C, H, W = shape
    x = torch.randn(shape, device=DEV).to(dt)
    w = (torch.randn(C, 1, 3, 3, device=DEV) * 0.3)
    b = torch.randn(C, device=DEV) if has_bias else None
    assert ops.flat2_ok(x)
    x2 = torch.ops.vmambair.dwconv3x3_silu_flat2_fwd(x, w, b)
    y = torch.ops.vmambair.dwconv3x3_silu_fwd(x, w, b)
    assert torch.equal(x2, ops.cross_scan2(y))
    # against fp32 torch as well: silu(conv) row-major and column-major
    ref = F.silu(F.conv2d(x.float(), w, b, padding=1, groups=C))
    rt = 1e-4 if dt == torch.float32 else 2e-2
    assert_close(x2[:, 0].reshape(shape), ref, rt, rt, "row-major")
    assert_close(x2[:, 1].reshape(B, C, W, H).transpose(2, 3), ref, rt, rt, "column-major")
    g2 = torch.randn(B, 2, C, H * W, device=DEV).to(dt)
    dx, dw, db = torch.ops.vmambair.dwconv3x3_silu_flat2_bwd(x, w, b, g2, None)
    dx0, dw0, db0 = torch.ops.vmambair.dwconv3x3_silu_bwd(x, w, b, ops.cross_merge2(g2, H, W), None)
    assert torch.equal(dx, dx0) and torch.equal(dw, dw0) and torch.equal(db, db0)
    # written into a strided half buffer: same bits, nothing outside the half touched
    buf = torch.full((B, 2 * C, H, W), 7.0, device=DEV).to(dt)
    r = torch.ops.vmambair.dwconv3x3_silu_flat2_bwd(x, w, b, g2, buf[:, C:])
    assert r[0].numel() == 0 and torch.equal(buf[:, C:], dx0) and bool((buf[:, :C] == 7.0).all())


def test_shapes_the_flat2_form_leaves_to_the_separate_launches():
    """H not a multiple of 8, W / 8 not a power of two or beyond 32 lane groups: flat2_ok says no"""
    for shape, dt in (((1, 4, 12, 64), torch.bfloat16), ((1, 4, 16, 160), torch.bfloat16), ((1, 4, 8, 512), torch.bfloat16),
                      ((1, 4, 12, 64), torch.float32), ((1, 4, 16, 20), torch.float16)):
        assert not ops.flat2_ok(torch.empty(shape, dtype=dt, device=DEV))
    for shape in ((1, 4, 16, 64), (1, 4, 8, 8), (1, 4, 128, 128), (1, 4, 64, 256)):
        assert ops.flat2_ok(torch.empty(shape, dtype=torch.bfloat16, device=DEV))


@pytest.mark.parametrize("variant,hw", [("srgan", (32, 32)), ("mamber32", (16, 64)), ("realsr", (8, 8))])
def test_block_with_the_conv_core_node_is_bit_identical(variant, hw, monkeypatch):
    """SS2D_1 with ConvCoreFn (convolution + flattenings + core as one node) against the separate nodes: same outputs, same
    gradients of the input and of every parameter, bit for bit, under bf16 autocast"""
    from vmambair_amd.oss_block import MamberBlock
    torch.manual_seed(2)
    blk = MamberBlock(48, variant=variant).to(DEV)
    x = torch.randn(2, 48, *hw, device=DEV)

    def run():
        blk.zero_grad()
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = blk(xi)
        names, todo = set(), [out.grad_fn]
        while todo:   # which autograd nodes built this output?
            fn = todo.pop()
            if fn is not None and fn not in names:
                names.add(fn)
                todo.extend(f for f, _ in fn.next_functions)
        out.float().square().mean().backward()
        return (out.detach().clone(), xi.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()},
                {type(f).__name__ for f in names})

    o1, g1, p1, nodes1 = run()
    monkeypatch.setattr(ops.dwconv, "DW_FLAT2", False)
    o0, g0, p0, nodes0 = run()
    assert any(n.startswith("ConvCoreFn") for n in nodes1) and not any(n.startswith("ConvCoreFn") for n in nodes0)
    assert any(n.startswith("SS2DCoreFn") for n in nodes0) and not any(n.startswith("SS2DCoreFn") for n in nodes1)
    assert torch.equal(o1, o0) and torch.equal(g1, g0)
    for n in p0:
        assert torch.equal(p1[n], p0[n]), n
