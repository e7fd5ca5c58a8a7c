"""Thin dense 3x3 convolutions (oss_conv3x3_thin.hip; <= 4 channels in or out) against plain PyTorch fp32: the layers the UNets
open and close with (OverlapPatchEmbed, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:520-528; the x4 tail's last layer,
archs/common.py:45-60; Mamber32.output, Deraining/basicsr/models/archs/mamber32_arch.py:608)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(shape, cout, dt, has_bias, seed=0):
    torch.manual_seed(seed)
    B, Cin, H, W = shape
    x = torch.randn(shape).to(dt)
    w = torch.randn(cout, Cin, 3, 3) / (3.0 * Cin ** 0.5)
    b = torch.randn(cout) * 0.1 if has_bias else None
    dy = torch.randn(B, cout, H, W).to(dt)
    return x, w, b, dy


def _ref(x, w, b, dy):
    xx, ww = x.float().requires_grad_(), w.clone().requires_grad_()
    bb = None if b is None else b.clone().requires_grad_()
    y = F.conv2d(xx, ww, bb, padding=1)
    y.backward(dy.float())
    return y.detach(), xx.grad, ww.grad, None if bb is None else bb.grad


# (B, Cin, H, W) -> Cout: the tail's last layer (many -> 3), patch_embed (3 -> many), widths whose lane groups tile a wave (64, 128, 8)
# and widths that make rows straddle waves (24, 160, 40), ragged last workgroups, 1 / 2 / 4 thin channels
SHAPES = [((2, 96, 64, 64), 3), ((1, 96, 37, 128), 3), ((2, 12, 16, 24), 3), ((1, 7, 5, 160), 3), ((3, 5, 9, 8), 1), ((1, 6, 11, 40), 4),
          ((2, 3, 64, 64), 48), ((1, 3, 21, 24), 48), ((1, 3, 6, 160), 10), ((2, 1, 8, 16), 5), ((1, 4, 13, 72), 9), ((1, 2, 33, 8), 17)]


@pytest.mark.parametrize("shape,cout", SHAPES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("has_bias", [True, False])
def test_thin_conv_matches_torch(shape, cout, dt, has_bias):
    from vmambair_amd import ops
    x, w, b, dy = _case(shape, cout, dt, has_bias)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, w, b, dy)
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    bd = None if b is None else b.to(DEV).requires_grad_()
    assert ops.conv3x3.thin_ok(xd, wd)
    y = ops.ThinConv3x3Fn.apply(xd, wd, bd)
    assert y.dtype == dt
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    eps = 2e-2 if dt == torch.bfloat16 else 3e-3       # one rounding of the 16-bit output / input gradient
    assert_close(y, y_ref, eps, eps * float(y_ref.abs().max()) * 0.25 + 1e-6, "y")
    assert_close(xd.grad, dx_ref, eps, eps * float(dx_ref.abs().max()) * 0.25 + 1e-6, "dx")
    # weight / bias gradients: fp32 sums of products of the SAME 16-bit inputs -> fp32 round-off only
    assert_close(wd.grad, dw_ref, 2e-4, 2e-4 * float(dw_ref.abs().max()) + 1e-6, "dw")
    if has_bias:
        assert_close(bd.grad, db_ref, 2e-4, 2e-4 * float(db_ref.abs().max()) + 1e-6, "db")


def test_thin_conv_reruns_are_bit_identical_and_strided_inputs_work():
    from vmambair_amd import ops
    x, w, b, dy = _case((2, 96, 32, 64), 3, torch.bfloat16, True, seed=3)
    big = torch.randn(2, 200, 32, 64).to(torch.bfloat16).to(DEV)
    xv = big[:, 8:104]                       # a channel slice: batch stride != Cin * H * W
    xv.copy_(x.to(DEV))
    wd, bd = w.to(DEV), b.to(DEV)
    outs = []
    for _ in range(2):
        y = torch.ops.vmambair.conv3x3_thin_fwd(xv, wd, bd)
        g = torch.ops.vmambair.conv3x3_thin_bwd(xv, wd, dy.to(DEV), True, True)
        outs.append([y] + list(g))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, w, b, dy)
    assert_close(outs[0][0], y_ref, 2e-2, 5e-3 * float(y_ref.abs().max()), "y (strided x)")
    assert_close(outs[0][2], dw_ref, 2e-4, 2e-4 * float(dw_ref.abs().max()), "dw (strided x)")


def test_layers_the_thin_kernels_leave_to_the_vendor_convolution():
    """fp32 activations, widths that are not a multiple of 8, both sides wide: conv3x3() calls the module"""
    from vmambair_amd.ops.conv3x3 import conv3x3, thin_ok
    conv = torch.nn.Conv2d(96, 3, 3, padding=1).to(DEV)
    for x in (torch.randn(1, 96, 8, 16, device=DEV), torch.randn(1, 96, 8, 12, device=DEV).to(torch.bfloat16)):
        assert not thin_ok(x, conv.weight)
    wide = torch.nn.Conv2d(96, 384, 3, padding=1).to(DEV)
    assert not thin_ok(torch.randn(1, 96, 8, 16, device=DEV).to(torch.bfloat16), wide.weight)
    x = torch.randn(1, 96, 8, 16, device=DEV)
    assert torch.equal(conv3x3(x, conv), conv(x))
    with torch.autocast("cuda", dtype=torch.bfloat16):   # an fp32 input is narrowed as autocast would, then takes the thin kernels
        y = conv3x3(x, conv)
        assert y.dtype == torch.bfloat16 and y.grad_fn.__class__.__name__.startswith("ThinConv3x3Fn")
        assert_close(y, conv(x), 2e-2, 2e-2, "autocast")
