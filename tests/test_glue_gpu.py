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


@pytest.mark.parametrize("shape", [(2, 48, 16, 24), (1, 96, 64, 64), (3, 384, 8, 8), (2, 7, 5, 3), (4, 48, 1, 1)])
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


@pytest.mark.parametrize("shape", [(2, 48, 64, 64), (1, 5, 3, 7), (2, 8, 33, 70), (1, 96, 16, 8)])
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
