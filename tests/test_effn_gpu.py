"""The second half of an OSS block as ONE forward launch (csrc/oss_effn.hip, inference only):
``x + project_out(gelu(x1) * x2)``, ``x1, x2 = dwconv(project_in(norm2(x))).chunk(2, 1)``
(SRGAN/VmambaIR/archs/MambaSISR6_arch.py:201-218 FeedForward, :513-516 the block, :144-195 LayerNorm) against plain PyTorch fp32 of
the same ops and against the launch-per-layer chain of this repo (LayerNorm inside project_in, dwconv + gate, project_out + skip)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, golden_files, load_golden
from vmambair_amd import ops, oss_block
from vmambair_amd.ops import ffn as ffn_ops

pytestmark = [pytest.mark.gpu, pytest.mark.tier(1)]   # one op against plain PyTorch fp32 (and against the chain of this repo)
DEV = "cuda:0"


def reference(x, norm, ff):
    """fp32 PyTorch on the values the kernel reads (x and the two 1x1 weights as rounded to the I/O type)"""
    dt = x.dtype
    xf = x.float()
    mu = xf.mean(1, keepdim=True)
    var = xf.var(1, keepdim=True, unbiased=False)
    w = norm.body.weight.float().view(1, -1, 1, 1)
    if norm.with_bias:
        n = (xf - mu) / torch.sqrt(var + 1e-5) * w + norm.body.bias.float().view(1, -1, 1, 1)
    else:
        n = xf / torch.sqrt(var + 1e-5) * w
    n = n.to(dt).float()                                              # the chain stores norm2(x) in the I/O type
    t = F.conv2d(n, ff.project_in.weight.to(dt).float()).to(dt).float()
    t = F.conv2d(t, ff.dwconv.weight.float(), None, padding=1, groups=t.shape[1])
    x1, x2 = t.chunk(2, dim=1)
    g = (F.gelu(x1) * x2).to(dt).float()
    return xf + F.conv2d(g, ff.project_out.weight.to(dt).float())


_CASES = [  # (B, D, H, W, LayerNorm type)
    (1, 48, 64, 64, "WithBias"),      # RealSR level 1 (encoder), hidden 127
    (2, 96, 40, 48, "WithBias"),      # dim 96, hidden 255: 16 chunks, the last one ragged
    (1, 96, 20, 24, "WithBias"),      # H not a multiple of the 8-row tile, W a multiple of 8 only
    (1, 32, 8, 8, "BiasFree"),        # one tile, half of it outside the image; the un-centred LayerNorm form
    (1, 64, 17, 40, "WithBias"),
    (3, 48, 9, 16, "BiasFree"),
    (1, 96, 136, 144, "WithBias"),    # a RealSR corner tile (tile 128 + halo 16: 144 wide) at level 1
]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("B,D,H,W,ln", _CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}x{c[3]}-{c[4]}" for c in _CASES])
def test_fused_effn_forward_against_pytorch_fp32_and_the_chain(B, D, H, W, ln, dt):
    torch.manual_seed(5)
    norm = oss_block.LayerNorm(D, ln).to(DEV)
    ff = oss_block.FeedForward(D, 2.66, False).to(DEV)
    with torch.no_grad():
        norm.body.weight.uniform_(0.5, 1.5)
        if norm.with_bias:
            norm.body.bias.normal_(0, 0.3)
        ff.dwconv.weight.mul_(2.0)
    x = (torch.randn(B, D, H, W, device=DEV) * 1.5 + 0.2).to(dt)
    hidden = ff.project_out.in_channels
    assert ffn_ops.effn_fwd_ok(x, hidden)
    with torch.no_grad():
        want = reference(x, norm, ff)
        got = ff(x, pre_norm=norm)
        w_in, w_dw, w_out = ops.effn_round_weights(ff.project_in.weight, ff.dwconv.weight, ff.project_out.weight, dt)
        direct = torch.ops.vmambair.effn_fwd(x, norm.body.weight, norm.body.bias, w_in, w_dw, w_out, hidden)
        ffn_ops.EFFN_FUSED = False
        try:
            chain = ff(x, pre_norm=norm)
        finally:
            ffn_ops.EFFN_FUSED = True
    assert torch.equal(got, direct), "FeedForward.forward without a backward to prepare IS the one-launch form"
    assert got.dtype == dt and got.shape == x.shape
    # one rounding of the result to the I/O type (+ the differences of two roundings of intermediates that fall on the other side)
    rt = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]
    scale = float(want.abs().max())
    assert_close(got, want, rt, rt * scale * 0.5, "fused vs PyTorch fp32")
    assert_close(got, chain.float(), rt, rt * scale * 0.5, "fused vs the launch-per-layer chain")
    # and the chain is as close to the reference as the fused form is (the tolerance is not hiding a bias of the new kernel)
    e_f = float((got.float() - want).abs().mean()), float((chain.float() - want).abs().mean())
    assert e_f[0] <= 1.5 * e_f[1] + 1e-6, f"mean |error| fused {e_f[0]:.3e} vs chain {e_f[1]:.3e}"


def test_fused_effn_leaves_training_and_unsupported_streams_to_the_chain():
    """with gradients enabled the module builds the autograd chain; fp32 streams, widths that are not a multiple of 8 and channel counts
    without an instantiation are not taken (effn_fwd_ok) and the raw op refuses them loudly"""
    torch.manual_seed(6)
    norm = oss_block.LayerNorm(48, "WithBias").to(DEV)
    ff = oss_block.FeedForward(48, 2.66, False).to(DEV)
    x = torch.randn(1, 48, 16, 16, device=DEV).to(torch.bfloat16)
    out = ff(x.clone().requires_grad_(), pre_norm=norm)
    assert out.grad_fn is not None
    assert not ffn_ops.effn_fwd_ok(x.float(), 127)
    assert not ffn_ops.effn_fwd_ok(torch.empty(1, 48, 16, 20, device=DEV, dtype=torch.float16), 127)
    assert not ffn_ops.effn_fwd_ok(torch.empty(1, 192, 16, 16, device=DEV, dtype=torch.float16), 510)
    assert not ffn_ops.effn_fwd_ok(torch.empty(1, 48, 16, 16, device=DEV, dtype=torch.float16)[:, :, :, ::2], 127)
    w_in, w_dw, w_out = ops.effn_round_weights(ff.project_in.weight, ff.dwconv.weight, ff.project_out.weight, torch.float16)
    assert w_in.shape == (256, 48) and w_dw.shape == (256, 9) and w_out.shape == (48, 128)
    assert float(w_in[127:128].abs().sum()) == 0 and float(w_in[255:].abs().sum()) == 0 and float(w_out[:, 127:].abs().sum()) == 0
    with pytest.raises(RuntimeError):
        torch.ops.vmambair.effn_fwd(x.float(), norm.body.weight, norm.body.bias, w_in, w_dw, w_out, 127)
    with pytest.raises(RuntimeError):   # weights of the wrong type
        torch.ops.vmambair.effn_fwd(x, norm.body.weight, norm.body.bias, w_in, w_dw, w_out, 127)
    with pytest.raises(RuntimeError):   # the unpadded depth-wise weight
        torch.ops.vmambair.effn_fwd(x.half(), norm.body.weight, norm.body.bias, w_in, ff.dwconv.weight.reshape(254, 9), w_out, 127)


def test_rounded_weights_follow_the_parameters():
    """the module keeps the rounded copies per weight version: an in-place update (an optimizer step, load_state_dict) refreshes them"""
    torch.manual_seed(7)
    norm = oss_block.LayerNorm(48, "WithBias").to(DEV)
    ff = oss_block.FeedForward(48, 2.66, False).to(DEV)
    x = torch.randn(1, 48, 16, 16, device=DEV).to(torch.float16)
    with torch.no_grad():
        a = ff(x, pre_norm=norm)
        first = ff._rounded(torch.float16)
        assert ff._rounded(torch.float16)[0] is first[0]
        ff.project_out.weight.mul_(2.0)
        b = ff(x, pre_norm=norm)
    assert ff._rounded(torch.float16)[0] is not first[0]
    assert_close((b.float() - x.float()), 2.0 * (a.float() - x.float()), 5e-3, 5e-3 * float(a.abs().max()), "doubled project_out")


def test_round_weights_kernel_matches_the_torch_construction():
    """oss_effn_round_weights (one launch) against slicing / padding / casting with torch ops"""
    torch.manual_seed(8)
    for D, dt in ((48, torch.float16), (96, torch.bfloat16), (32, torch.float16)):
        ff = oss_block.FeedForward(D, 2.66, False).to(DEV)
        h = ff.project_out.in_channels
        hp = (h + 15) // 16 * 16
        w_in, w_dw, w_out = ops.effn_round_weights(ff.project_in.weight, ff.dwconv.weight, ff.project_out.weight, dt)
        want_in = torch.zeros(2, hp, D, device=DEV, dtype=dt)
        want_in[:, :h] = ff.project_in.weight.detach().reshape(2, h, D).to(dt)
        want_dw = torch.zeros(2, hp, 9, device=DEV)
        want_dw[:, :h] = ff.dwconv.weight.detach().reshape(2, h, 9)
        want_out = torch.zeros(D, hp, device=DEV, dtype=dt)
        want_out[:, :h] = ff.project_out.weight.detach().reshape(D, h).to(dt)
        assert torch.equal(w_in, want_in.reshape(2 * hp, D)) and torch.equal(w_dw, want_dw.reshape(2 * hp, 9)) and torch.equal(w_out, want_out)


def test_a_captured_inference_graph_follows_in_place_weight_updates():
    """the rounded weight copies are cached per weight version in eager mode; inside a graph capture the rounding launch is captured
    with the forward, so a replay after an in-place update of the parameters (load_state_dict, an optimizer step between two
    validations) computes with the NEW weights -- as every other kernel of the net does, reading the fp32 parameters directly"""
    torch.manual_seed(9)
    norm = oss_block.LayerNorm(48, "WithBias").to(DEV)
    ff = oss_block.FeedForward(48, 2.66, False).to(DEV)
    x = torch.randn(2, 48, 24, 32, device=DEV).to(torch.float16)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ff(x, pre_norm=norm)   # warm-up: fills the eager cache
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = ff(x, pre_norm=norm)
        g.replay()
        torch.cuda.synchronize()
        first = out.clone()
        assert torch.equal(first, ff(x, pre_norm=norm))
        ff.project_in.weight.mul_(1.5)
        ff.dwconv.weight.add_(0.05)
        ff.project_out.weight.mul_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        fresh = ff(x, pre_norm=norm)
    assert not torch.equal(out, first)
    assert torch.equal(out, fresh), "the replay used the parameters as they are now"


def test_fused_effn_under_inference_mode_with_inference_parameters():
    """a module built under ``torch.inference_mode()`` has parameters without a version counter: the rounded copies are then made per call"""
    torch.manual_seed(10)
    with torch.inference_mode():
        norm = oss_block.LayerNorm(48, "WithBias").to(DEV)
        ff = oss_block.FeedForward(48, 2.66, False).to(DEV)
        x = torch.randn(1, 48, 16, 24, device=DEV).to(torch.float16)
        got = ff(x, pre_norm=norm)
        want = reference(x, norm, ff)
    assert_close(got, want, 2e-3, 1e-3 * float(want.abs().max()), "inference mode")


def _g9(name, device):
    """the reference's ``norm2`` + ``ffn`` of one G9 fixture as this repo's modules (state dict loaded strictly), input, reference output"""
    z = load_golden(name)
    dim, ln = int(z["dim"]), str(z["ln"])
    norm = oss_block.LayerNorm(dim, ln)
    ff = oss_block.FeedForward(dim, 2.66, False)
    norm.load_state_dict({k[len("sd.norm2."):]: v.float() for k, v in z.items() if k.startswith("sd.norm2.")}, strict=True)
    ff.load_state_dict({k[len("sd.ffn."):]: v.float() for k, v in z.items() if k.startswith("sd.ffn.")}, strict=True)
    return norm.to(device), ff.to(device), z["x"].float(), z["y"].float()


@pytest.mark.tier(0)
@pytest.mark.parametrize("name", golden_files("g9_effn_"))
def test_fused_effn_forward_against_the_reference_golden_vectors(name):
    """G9 (tests/golden/make_golden.py: ``x + ffn(norm2(x))`` of the reference's own MamberBlock, SRGAN / RealSR / mamber32 trees, fp16-exact
    weights and input, fp32 arithmetic): the one-launch forward on the fp16 input must land within one fp16 rounding of the result plus
    the roundings of the three intermediates the chain also rounds -- relative L2 <= 1e-3 (the stated fp16 output tolerance of the block
    fixtures, tests/test_configs_gpu.py), and elementwise 3e-3 of the output scale"""
    norm, ff, x, want = _g9(name, DEV)
    xh = x.to(DEV).half()
    assert ffn_ops.effn_fwd_ok(xh, ff.project_out.in_channels)
    with torch.no_grad():
        got = ff(xh, pre_norm=norm)
        ffn_ops.EFFN_FUSED = False
        try:
            chain = ff(xh, pre_norm=norm)
        finally:
            ffn_ops.EFFN_FUSED = True
    rel = float((got.float().cpu() - want).norm() / want.norm())
    relc = float((chain.float().cpu() - want).norm() / want.norm())
    print(f"[g9] {name}: rel-L2 vs the reference: one launch {rel:.2e}, chain {relc:.2e}")
    assert rel <= 1e-3, rel
    assert_close(got, want, 3e-3, 3e-3 * float(want.abs().max()), "fused EFFN vs the reference's output")
