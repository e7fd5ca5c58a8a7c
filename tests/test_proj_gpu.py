"""Projection / flattening kernels of the spatial branch (oss_proj.hip) and the fused SS2DCoreFn node against
plain PyTorch fp32 references of the same ops (the reference's einsums, MambaSISR6_arch.py:395-431)."""
import pytest
import torch

from conftest import assert_close, load_golden
from vmambair_amd import ops
from vmambair_amd.oss_block import SS2D_1

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
IDS = ["f32", "bf16", "f16"]


@pytest.mark.parametrize("shape", [(2, 5, 3, 7), (1, 48, 64, 64), (2, 8, 33, 70), (3, 96, 16, 8), (1, 4, 1, 9),
                                   (1, 3, 128, 128), (2, 4, 66, 130), (1, 2, 64, 200), (1, 2, 190, 64)])   # 64 x 64 pair tiles, ragged edges
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_cross_scan2_and_merge2_bit_exact(shape, dt):
    torch.manual_seed(0)
    B, D, H, W = shape
    x = torch.randint(-64, 64, shape).to(dt).to(DEV)
    x2 = ops.cross_scan2(x)
    assert torch.equal(x2[:, 0], x.flatten(2, 3))
    assert torch.equal(x2[:, 1], x.transpose(2, 3).flatten(2, 3))
    g2 = torch.randint(-64, 64, (B, 2, D, H * W)).to(dt).to(DEV)
    want = g2[:, 0].view(B, D, H, W) + g2[:, 1].view(B, D, W, H).transpose(2, 3)  # small integers: exact in every dtype
    assert torch.equal(ops.cross_merge2(g2, H, W), want)


def test_cross_scan2_on_channel_strided_view_and_narrowing():
    torch.manual_seed(1)
    big = torch.randn(2, 32, 6, 10, device=DEV)
    x = big.chunk(2, dim=1)[1]
    x2 = ops.cross_scan2(x, torch.bfloat16)
    xb = x.to(torch.bfloat16)
    assert torch.equal(x2[:, 0], xb.flatten(2, 3)) and torch.equal(x2[:, 1], xb.transpose(2, 3).flatten(2, 3))


def test_cross_scan2_matches_the_reference_direction_maps():
    """golden G2 (integer data through the reference's stack/flip/transposes): directions 0, 1 are x2; 2, 3 its mirror"""
    z = load_golden("g2_perm.npz")
    x2 = ops.cross_scan2(z["x"].to(DEV)).cpu()
    xs = z["xs"].view(x2.shape[0], 4, x2.shape[2], -1)
    assert torch.equal(x2, xs[:, :2]) and torch.equal(x2.flip(-1), xs[:, 2:])


def proj_ref(x2, wx, wdt, round_to):
    """fp32 reference of the omni projections with the kernels' rounding points (x_dbl rounded before dt_proj)"""
    R = wdt.shape[2]
    z = torch.cat([torch.einsum("bjdl,jcd->bjcl", x2, wx[0:2]), torch.einsum("bjdl,jcd->bjcl", x2, wx[2:4])], dim=1)
    zr = z.to(round_to).float()
    dts = torch.einsum("bkrl,kdr->bkdl", zr[:, :, :R], wdt)
    return z, dts.reshape(x2.shape[0], -1, x2.shape[3])


SHAPES = [  # (B, D, R, N, L)
    (2, 16, 1, 16, 35), (1, 96, 3, 16, 4096), (2, 192, 6, 16, 1000), (2, 384, 12, 16, 256), (2, 768, 24, 16, 64),
    (1, 20, 2, 4, 70), (1, 104, 5, 8, 129),
]


@pytest.fixture(params=["mfma", "valu"])
def proj_path(request):
    """two implementations each: matrix cores (16-bit: v_mfma_f32_32x32x16; float I/O, round 4: v_mfma_f32_32x32x2_f32 for lengths
    that are a multiple of 4) / vector ALU"""
    ops.proj_set_path(request.param == "valu")
    yield request.param
    ops.proj_set_path(False)


@pytest.mark.parametrize("shape", SHAPES, ids=[f"D{s[1]}L{s[4]}" for s in SHAPES])
@pytest.mark.parametrize("dt", DTYPES, ids=IDS)
def test_projections_forward_backward(shape, dt, proj_path):
    torch.manual_seed(2)
    B, D, R, N, L = shape
    Cc = R + 2 * N
    x2 = torch.randn(B, 2, D, L).to(dt)
    wx = torch.randn(4, Cc, D) / D ** 0.5
    wdt = torch.randn(4, D, R) / R ** 0.5
    ddts = torch.randn(B, 4 * D, L).to(dt)
    dbc = torch.randn(B, 4, 2 * N, L).to(dt)
    du = torch.randn(B, 4 * D, L).to(dt)
    lo = dt == torch.float32
    # reference (fp32 math on the same 16-bit inputs), differentiable
    x2r, wxr, wdtr = x2.float().requires_grad_(), wx.clone().requires_grad_(), wdt.clone().requires_grad_()
    z, dts_r = proj_ref(x2r, wxr, wdtr, dt)
    # forward
    xdbl, dts = ops.proj_fwd(x2.to(DEV), wx.to(DEV), wdt.to(DEV))
    rt, at = (1e-4, 1e-5) if lo else (1.2e-2, 1e-2)
    assert_close(xdbl, z, rt, at * float(z.abs().max()), "xdbl")
    assert_close(dts, dts_r, rt if lo else 2e-2, (at if lo else 3e-2) * float(dts_r.abs().max()), "dts")
    # backward: gradients of  <dts, ddts> + <xdbl[B, C rows], dbc>  (+ du passed straight through)
    loss = (dts_r * ddts.float()).sum() + (z[:, :, R:] * dbc.float()).sum()
    # the kernels round the dt rows of dxdbl to the I/O type before using them; mirror that with a straight-through hook
    gx, gwx, gwdt = torch.autograd.grad(loss, (x2r, wxr, wdtr))
    gx = gx + du.float().view(B, 2, 2, D, L).sum(1)
    dxdbl = torch.zeros(B, 4, Cc, L, dtype=dt, device=DEV)
    dxdbl[:, :, R:] = dbc.to(DEV)
    dx2 = ops.proj_dgrad(ddts.to(DEV), dxdbl, du.to(DEV), wx.to(DEV), wdt.to(DEV))
    dz_dt = torch.einsum("bkdl,kdr->bkrl", ddts.float().view(B, 4, D, L), wdt)
    assert_close(dxdbl[:, :, :R], dz_dt, rt, at * float(dz_dt.abs().max()), "dxdbl dt rows")
    assert torch.equal(dxdbl[:, :, R:].cpu(), dbc), "dB / dC rows must be left alone"
    assert_close(dx2, gx, rt if lo else 2e-2, (at if lo else 2e-2) * float(gx.abs().max()), "dx2")
    dwx, dwdt = ops.proj_wgrad(x2.to(DEV), xdbl, dxdbl, ddts.to(DEV), R)
    assert_close(dwx, gwx, 1e-3 if lo else 2e-2, (1e-4 if lo else 2e-2) * float(gwx.abs().max()), "dwx")
    assert_close(dwdt, gwdt, 1e-3 if lo else 2e-2, (1e-4 if lo else 2e-2) * float(gwdt.abs().max()), "dwdt")


def test_proj_dgrad_without_du():
    torch.manual_seed(3)
    B, D, R, N, L = 1, 16, 1, 16, 40
    wx, wdt = torch.randn(4, R + 2 * N, D, device=DEV), torch.randn(4, D, R, device=DEV)
    ddts = torch.randn(B, 4 * D, L, device=DEV)
    a = torch.randn(B, 4, R + 2 * N, L, device=DEV)
    b = a.clone()
    dx_a = ops.proj_dgrad(ddts, a, None, wx, wdt)
    dx_b = ops.proj_dgrad(ddts, b, torch.zeros(B, 4 * D, L, device=DEV), wx, wdt)
    assert torch.equal(dx_a, dx_b) and torch.equal(a, b)


@pytest.mark.parametrize("cfg", [(8, 7, 9), (16, 12, 12), (48, 16, 16), (96, 8, 8)], ids=lambda c: f"d{c[0]}_{c[1]}x{c[2]}")
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_fused_core_equals_unfused_omni_path(cfg, dt):
    """SS2DCoreFn (one node) vs the einsum + OmniScanMergeFn path it replaces, forward and every gradient"""
    torch.manual_seed(4)
    d_model, H, W = cfg
    m = SS2D_1(d_model=d_model, variant="srgan").to(DEV)
    x = torch.randn(2, m.d_inner, H, W, device=DEV).to(dt)
    gate = torch.randn(2, m.d_inner, H, W, device=DEV).to(dt)
    res = []
    for fused in (True, False):
        m.fused_core = fused
        m.zero_grad()
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt != torch.float32):
            y = m.forward_core(xi, gate=gate)
        y.float().square().sum().backward()
        res.append((y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    lo = dt == torch.float32
    assert_close(res[0][0], res[1][0], 1e-4 if lo else 3e-2, (1e-4 if lo else 3e-2) * float(res[1][0].abs().max()), "y")
    assert_close(res[0][1], res[1][1], 1e-3 if lo else 5e-2, (1e-4 if lo else 5e-2) * float(res[1][1].abs().max()), "dx")
    assert set(res[0][2]) == set(res[1][2])
    for k in res[1][2]:
        sc = float(res[1][2][k].abs().max())
        assert_close(res[0][2][k], res[1][2][k], 1e-3 if lo else 6e-2, (2e-4 if lo else 6e-2) * max(sc, 1e-6), k)


def test_fused_core_against_oracle_twin(oracle_cpu_kernel):
    """fp32 fused core on the GPU vs the literal reference data flow on the CPU oracle (oracle/cpu_twins.py)"""
    torch.manual_seed(5)
    m = SS2D_1(d_model=16, variant="srgan")
    x = torch.randn(2, m.d_inner, 9, 6)
    xc = x.clone().requires_grad_()
    args = (m.x_proj_weight, m.dt_projs_weight, m.A_logs, m.Ds, m.dt_projs_bias)
    yc = ops.SS2DCoreFn.apply(xc, *args)
    yc.square().sum().backward()
    want = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    md = m.to(DEV)
    xg = x.to(DEV).requires_grad_()
    yg = ops.SS2DCoreFn.apply(xg, md.x_proj_weight, md.dt_projs_weight, md.A_logs, md.Ds, md.dt_projs_bias)
    yg.square().sum().backward()
    assert_close(yg, yc, 1e-4, 1e-4 * float(yc.abs().max()), "y")
    assert_close(xg.grad, xc.grad, 1e-3, 2e-4 * float(xc.grad.abs().max()), "dx")
    for k, g in want.items():
        got = dict(md.named_parameters())[k].grad
        assert_close(got, g, 2e-3, 3e-4 * float(g.abs().max()), k)
