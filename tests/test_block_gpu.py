"""OSS block / UNet mirrors on the GPU with the HIP scan, against the block- and net-level golden
vectors of the reference (G3, G4) and under bf16 autocast (BASELINE.json config 2)."""
import pytest
import torch

from conftest import assert_close, load_golden
from vmambair_amd.archs import MambaSISR6, Mamber32
from vmambair_amd.oss_block import MamberBlock, SS2D_1

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _state(z):
    return {k[3:]: v for k, v in z.items() if k.startswith("sd.")}


BLOCKS = [
    ("g3_block_srgan_ss2d_d48.npz", lambda: SS2D_1(d_model=48, ssm_ratio=1, variant="srgan")),
    ("g3_block_srgan_mamber_d48.npz", lambda: MamberBlock(48, variant="srgan")),
    ("g3_block_mamber32_d48.npz", lambda: MamberBlock(48, variant="mamber32")),
    ("g3_block_mamber33_d48.npz", lambda: MamberBlock(48, variant="mamber33")),
    ("g3_block_realsr_mamber_d48.npz", lambda: MamberBlock(48, variant="realsr")),
]


@pytest.mark.parametrize("name,make", BLOCKS, ids=[b[0][9:-4] for b in BLOCKS])
def test_block_matches_reference_on_gpu(name, make):
    z = load_golden(name)
    m = make()
    m.load_state_dict(_state(z), strict=True)
    m.to(DEV)
    x = z["x"].to(DEV).requires_grad_()
    y = m(x)
    assert_close(y, z["y"], 1e-3, 1e-3, "block output")
    from vmambair_amd.ops import _common
    cats = _common.CAT_FALLBACKS
    y.backward(z["dy"].to(DEV))
    # the gradients of the two halves of xz were written in place by their producer kernels (mutated operator arguments,
    # ops/dwconv.py, ops/layernorm.py): the split's backward never had to cat
    assert _common.CAT_FALLBACKS == cats
    assert_close(x.grad, z["dx"], 3e-3, 3e-3, "input grad")
    for k, p in m.named_parameters():
        ref = z["grad." + k]
        if k.endswith("conv_cout.bias"):
            continue  # exact gradient is 0 (constant before a LayerNorm); both sides return noise
        scale = max(1.0, float(ref.abs().max()))
        assert_close(p.grad, ref, 5e-3, 1e-3 * scale, f"grad {k}")


@pytest.mark.parametrize("name,cls", [("g4_net_mambasisr6_d8.npz", MambaSISR6), ("g4_net_mamber32_d8.npz", Mamber32)])
def test_net_matches_reference_on_gpu(name, cls):
    z = load_golden(name)
    net = cls(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net.load_state_dict(_state(z), strict=True)
    net.to(DEV)
    with torch.no_grad():
        y = net(z["x"].to(DEV))
    assert_close(y, z["y"], 1e-3, 1e-3, "net output")


@pytest.mark.selfcheck
def test_block_under_bf16_autocast_trains():
    """config 2 runs the arch under bf16 autocast: the scan receives bf16 u/delta/B/C with fp32
    A/D/bias; output must stay close to the fp32 run."""
    z = load_golden("g3_block_srgan_mamber_d48.npz")
    m = MamberBlock(48, variant="srgan")
    m.load_state_dict(_state(z), strict=True)
    m.to(DEV)
    x = z["x"].to(DEV).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    assert y.dtype == torch.float32  # residual with the fp32 input
    assert_close(y, z["y"], 5e-2, 1e-1, "bf16 autocast output")
    y.backward(z["dy"].to(DEV))
    assert torch.isfinite(x.grad).all()
    ref = z["dx"]
    rel = (x.grad.cpu() - ref).norm() / ref.norm()
    assert rel < 0.1, f"bf16 input grad relative error {rel:.3f}"


def test_fused_adam_ema_matches_torch_adam():
    """oss_adam_ema_step vs torch.optim.Adam + the foreach EMA of the reference's optimize_parameters"""
    from vmambair_amd.optim import FusedAdamEMA
    torch.manual_seed(0)
    shapes = [(5,), (3, 7), (2049,), (48, 96, 1, 1), (1,), (4096,)]
    pa = [torch.randn(s, device=DEV) for s in shapes]
    pb = [p.clone() for p in pa]
    ema_a = [p.clone() for p in pa]
    ema_b = [p.clone() for p in pb]
    opt = torch.optim.Adam(pb, lr=2e-2, betas=(0.9, 0.99))
    fo = FusedAdamEMA(pa, ema_a, lr=2e-2, betas=(0.9, 0.99), ema_decay=0.9)
    for step in range(4):
        grads = [torch.randn(s, device=DEV) * (step + 1) for s in shapes]
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g.clone(), g.clone()
        fo.step()
        opt.step()
        torch._foreach_mul_(ema_b, 0.9)
        torch._foreach_add_(ema_b, pb, alpha=0.1)
    for p, q, e, f in zip(pa, pb, ema_a, ema_b):
        assert_close(p, q, 1e-5, 1e-6, "param")
        assert_close(e, f, 1e-5, 1e-6, "ema")


@pytest.mark.parametrize("dim,hw", [(48, (32, 24)), (96, (64, 64))])
def test_grouped_weight_gradient_launch_is_bit_identical(dim, hw):
    """weight-gradient products recorded during the backward and run as ONE grouped launch (oss_flush_wgrads) vs one launch per
    product.  With the same 512 pixels per partial product (oss_conv1x1_wgrad_set_span(1)): same tiles, same partials, same
    finishing sums -> torch.equal on every parameter gradient and on dx.  With the default span (fewer, longer partial products)
    the fp32 summation order differs: equal to round-off, and bit-identical from run to run."""
    from vmambair_amd import _capi, ops
    torch.manual_seed(0)
    m = MamberBlock(dim, variant="srgan").to(DEV)
    x = torch.randn(2, dim, *hw, device=DEV)
    gy = torch.randn(2, dim, *hw, device=DEV)
    lib = _capi.load()
    res = []
    for grouped, span in ((False, 1), (True, 1), (True, 1), (True, 4), (True, 4)):
        lib.oss_conv1x1_wgrad_set_span(span)
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        with ops.deferred_finishes(wgrads=grouped):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(xi)
            y.backward(gy)
            n_rec = ops.pending_wgrads()
            assert (n_rec > 0) == grouped, "the 1x1 / projection weight-gradient products are recorded only in grouped mode"
            if grouped:
                wt = ops.WgradTable(DEV, ops.pending_wgrad_table_bytes())
                ops.flush_wgrads(wt)
                assert ops.pending_wgrads() == 0
            ft = ops.FinishTable(DEV, ops.pending_finish_chunks())
            ops.flush_finishes(ft)
        torch.cuda.synchronize()
        res.append((xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, n_rec))
    assert res[1][2] >= 5, "in_conv, out_conv, project_in, project_out and the two projection products of one block"
    for other in (res[1], res[2]):
        assert torch.equal(res[0][0], other[0])
        for k, g in res[0][1].items():
            assert torch.equal(g, other[1][k]), k
    assert torch.equal(res[0][0], res[3][0])
    for k, g in res[0][1].items():
        assert torch.equal(res[3][1][k], res[4][1][k]), k
        assert_close(res[3][1][k], g, 1e-4, 1e-5 * max(1.0, float(g.abs().max())), k)
    assert lib.oss_deferred_wgrads() == 0
