"""Channel branch + gate of SS2D_1 as one autograd node (oss_channel.hip) against the op-by-op path it replaces
and against the literal reference data flow on the CPU oracle."""
import pytest
import torch

from conftest import assert_close
from vmambair_amd import ops
from vmambair_amd.oss_block import SS2D_1

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 5, 3, 7), (1, 48, 64, 64), (3, 16, 9, 8)])
def test_rowsum_and_row_affine(dt, shape):
    torch.manual_seed(0)
    B, C, H, W = shape
    lib = ops._capi.load()
    big = torch.randn(B, 2 * C, H, W, device=DEV).to(dt)
    a = big.chunk(2, dim=1)[1]                      # channel-strided view
    bm = torch.randn(B, C, H, W, device=DEV).to(dt)
    out = torch.empty(B, C, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for other, alpha in ((None, 1.0 / (H * W)), (bm, 1.0)):
        ops._capi.check(lib.oss_rowsum(ops._DT[dt], a.data_ptr(), None if other is None else other.data_ptr(), out.data_ptr(), B, C,
                                       H * W, a.stride(0), a.stride(1), 0 if other is None else other.stride(0),
                                       0 if other is None else other.stride(1), alpha, st), "rowsum")
        want = (a.float() * (1 if other is None else other.float())).sum(dim=(2, 3)) * alpha
        assert_close(out, want, 1e-4, 1e-4 * float(want.abs().max()) + 1e-6, "rowsum")
    mul, add = torch.randn(B, C, device=DEV), torch.randn(B, C, device=DEV)
    y = torch.empty(B, C, H, W, device=DEV, dtype=dt)
    ops._capi.check(lib.oss_row_affine(ops._DT[dt], a.data_ptr(), mul.data_ptr(), add.data_ptr(), y.data_ptr(), B, C, H * W,
                                       a.stride(0), a.stride(1), 0.5, st), "row_affine")
    want = a.float() * (1 + mul)[:, :, None, None] + 0.5 * add[:, :, None, None]
    assert_close(y, want, 1e-6 if dt == torch.float32 else 1e-2, 1e-6 if dt == torch.float32 else 2e-2, "row_affine")


def _run(m, x, dy, fused, dt):
    m.fused_channel = fused
    m.zero_grad()
    xi = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
        y = m(xi)
    y.backward(dy.to(y.dtype))
    return y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("variant", ["srgan", "mamber32", "mamber33", "realsr"])
@pytest.mark.parametrize("cfg", [(8, 7, 9), (48, 16, 16), (192, 8, 8), (384, 4, 4)], ids=lambda c: f"d{c[0]}_{c[1]}x{c[2]}")
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_fused_channel_branch_equals_op_by_op(variant, cfg, dt):
    torch.manual_seed(1)
    d_model, H, W = cfg
    m = SS2D_1(d_model=d_model, variant=variant).to(DEV)
    with torch.no_grad():  # randn-initialised channel parameters of the reference make the scan saturate; tame them
        m.Ac_logs.mul_(0.5)
        m.dtc_projs_bias.mul_(0.5)
    x = torch.randn(2, d_model, H, W, device=DEV)
    dy = torch.randn(2, d_model, H, W, device=DEV)
    (y0, dx0, g0), (y1, dx1, g1) = _run(m, x, dy, True, dt), _run(m, x, dy, False, dt)
    lo = dt == torch.float32
    assert_close(y0, y1, 2e-4 if lo else 4e-2, (2e-4 if lo else 4e-2) * float(y1.abs().max()), "y")
    assert_close(dx0, dx1, 2e-3 if lo else 6e-2, (5e-4 if lo else 6e-2) * float(dx1.abs().max()), "dx")
    assert set(g0) == set(g1)
    for k in g1:
        if k.endswith("conv_cout.bias"):
            continue  # exact gradient 0 (a constant in front of a LayerNorm): both sides are rounding noise
        sc = float(g1[k].abs().max())
        assert_close(g0[k], g1[k], 5e-3 if lo else 8e-2, (1e-3 if lo else 8e-2) * max(sc, 1e-6), k)


@pytest.mark.parametrize("variant", ["srgan", "mamber32", "realsr"])
def test_fused_channel_branch_against_oracle_twin(variant, oracle_cpu_kernel):
    """fp32 ChannelGateFn on the GPU vs the literal reference data flow on the CPU oracle (oracle/cpu_twins.py)"""
    torch.manual_seed(2)
    m = SS2D_1(d_model=16, variant=variant)
    with torch.no_grad():
        m.Ac_logs.mul_(0.5)
    lift = m.dc_inner is not None
    names = ["conv_cin.weight", "conv_cin.bias", "xc_proj_weight", "dtc_projs_weight", "dtc_projs_bias", "Ac_logs", "Dsc",
             "conv_cout.weight", "conv_cout.bias", "channel_norm.body.weight", "channel_norm.body.bias"]
    y2 = torch.randn(2, m.d_inner, 6, 5)
    g = torch.randn(2, m.d_inner, 6, 5)
    res = []
    for dev in ("cpu", DEV):
        mm = m.to(dev)
        pr = dict(mm.named_parameters())
        args = [pr[n] if (lift or not n.startswith("conv_c")) else None for n in names]
        yi = y2.detach().clone().to(dev).requires_grad_()
        mm.zero_grad()
        out = ops.ChannelGateFn.apply(yi, *args, mm.gate != "add")
        out.backward(g.to(dev))
        res.append((out.detach().cpu(), yi.grad.cpu(), {n: pr[n].grad.detach().cpu().clone() for n in names if pr.get(n) is not None and pr[n].grad is not None}))
    (oc, dc_, gc), (og, dg, gg) = res
    assert_close(og, oc, 2e-4, 2e-4 * float(oc.abs().max()), "out")
    assert_close(dg, dc_, 2e-3, 5e-4 * float(dc_.abs().max()), "dy2")
    assert set(gc) == set(gg)
    for k in gc:
        if k.endswith("conv_cout.bias"):
            continue  # exact gradient 0
        assert_close(gg[k], gc[k], 5e-3, 1e-3 * max(float(gc[k].abs().max()), 1e-6), k)
