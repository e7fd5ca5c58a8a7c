"""Parity on the workloads BASELINE.json names and on the production widths (round-2 additions).

  configs[0]  x2 SR 48x48 LQ, ONE OSS block, batch 2            -> g3_block_srgan_mamber_cfg1.npz (reference run, 69 s of
                                                                   selective_scan_ref on the build container's CPU)
  configs[1/2] widths D = 96 (decoder-1 / refinement), 192, 384 (latent; EFFN hidden 1021, dt_rank 24)
                                                                -> g3_block_*_d96 / _d192 / _d384.npz
  configs[3]  Deraining 128x128 patches: L = 16 384, and the progressive schedule's largest patch 384x384: L = 147 456
              (Deraining/basicsr/train.py:213-271) -> scan vs the oracle at those lengths, Mamber32 block at 128x128
              vs the CPU oracle twins
  configs[4]  RealSR fp16: g3_block_realsr_fp16_d48.npz (fp16-exact weights and input) under fp16 autocast
All fixtures hold fp16-exact weights / inputs (stored as float16) so that 16-bit runs start from the same numbers.

16-bit tolerances (stated, measured margins in profiles/r02_*): against the fp32 reference, relative L2 error
  bf16 autocast: y <= 5e-3, dx <= 8e-3, parameter gradients <= 5e-2     (bf16 eps = 3.9e-3; measured 2.0e-3 / 2.9e-3 / 2.3e-2)
  fp16 autocast: y <= 1e-3, dx <= 1e-3, parameter gradients <= 5e-3     (fp16 eps = 4.9e-4; measured 2.1e-4 / 2.9e-4 / 1.2e-3)
  (worst case over the fixtures, profiles/r02_16bit_parity_margins.txt)
A parameter gradient's error is taken relative to max(its own norm, 5 % of the largest gradient norm in the block): the
channel branch's gradients are sums of large cancelling terms with norms 100-1000x below the conv weights', and the
rounding noise of the activations they are formed from does not shrink with them.
"""
import pytest
import torch

from conftest import assert_close, load_golden
from vmambair_amd.oss_block import MamberBlock

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

COMPACT = {
    "d96": ("g3_block_srgan_mamber_d96.npz", 96, "srgan"),
    "d384": ("g3_block_srgan_mamber_d384.npz", 384, "srgan"),
    "m32_d192": ("g3_block_mamber32_d192.npz", 192, "mamber32"),
    "realsr_fp16": ("g3_block_realsr_fp16_d48.npz", 48, "realsr"),
    "cfg1": ("g3_block_srgan_mamber_cfg1.npz", 48, "srgan"),
}


def _load(tag):
    name, dim, variant = COMPACT[tag]
    z = load_golden(name)
    m = MamberBlock(dim, variant=variant)
    m.load_state_dict({k[3:]: v.float() for k, v in z.items() if k.startswith("sd.")}, strict=True)
    return z, m.to(DEV)


def _run(z, m, autocast=None):
    x = z["x"].float().to(DEV).requires_grad_()
    with torch.autocast("cuda", dtype=autocast, enabled=autocast is not None):
        y = m(x)
    y.backward(z["dy"].float().to(DEV))
    s, gs = int(z["io_stride"]), int(z["grad_stride"])
    grads = {k: p.grad.reshape(-1)[::gs] for k, p in m.named_parameters()}
    return y.detach()[..., ::s, ::s], x.grad[..., ::s, ::s], grads


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("tag", list(COMPACT))
def test_block_fp32_matches_reference(tag):
    """HIP fp32 vs the reference-generated fixture: output 1e-3 (x max|y|), input gradient 3e-3, parameter gradients
    5e-3 relative + 1e-3 of the tensor's largest entry (the G3 tolerances of round 1, scaled by magnitude because the
    wide blocks sum over up to 1021 channels)"""
    z, m = _load(tag)
    y, dx, grads = _run(z, m)
    assert_close(y, z["y"], 1e-3, 1e-3 * max(1.0, float(z["y"].abs().max())), "block output")
    assert_close(dx, z["dx"], 3e-3, 3e-3 * max(1.0, float(z["dx"].abs().max())), "input grad")
    for k, g in grads.items():
        if k.endswith("conv_cout.bias"):
            continue  # exact gradient is 0 (a constant in front of a LayerNorm); both sides return rounding noise
        assert_close(g, z["grad." + k], 5e-3, 1e-3 * max(1.0, float(z["gradmax." + k])), f"grad {k}")


LIMITS = {torch.bfloat16: (5e-3, 8e-3, 5e-2), torch.float16: (1e-3, 1e-3, 5e-3)}


@pytest.mark.parametrize("tag,dt", [("d96", torch.bfloat16), ("d384", torch.bfloat16), ("m32_d192", torch.bfloat16),
                                    ("cfg1", torch.bfloat16), ("realsr_fp16", torch.float16), ("d96", torch.float16)],
                         ids=lambda v: str(v).replace("torch.", ""))
def test_block_16bit_autocast_within_stated_tolerance(tag, dt, record_property):
    z, m = _load(tag)
    y, dx, grads = _run(z, m, autocast=dt)
    ly, ldx, lg = LIMITS[dt]
    ey, edx = rel_l2(y, z["y"]), rel_l2(dx, z["dx"])
    worst, wk = 0.0, ""
    big = max(float(z["grad." + k].norm()) for k in grads)
    floor = 0.05 * big
    # The error of a tensor is taken relative to max(its own norm, 5 % of the largest gradient norm): that criterion alone loses
    # power on small tensors (one whose norm is < 0.25 % of the largest would pass even if it were all zeros -- VERDICT r2 weak
    # #2), so every tensor must also POINT the right way: cosine with the reference >= 0.99 (>= 0.9 for tensors below 0.1 % of
    # the largest norm, where 16-bit rounding of the activations is a visible share of the tensor itself).
    wcos, wck = 1.0, ""
    for k, g in grads.items():
        ref = z["grad." + k]
        if k.endswith("conv_cout.bias"):
            continue
        gc = g.detach().float().cpu()
        e = float((gc - ref).norm()) / max(float(ref.norm()), floor)
        if e > worst:
            worst, wk = e, k
        rn = float(ref.norm())
        if rn > 0:
            c = float((gc * ref).sum() / (gc.norm().clamp_min(1e-30) * rn))
            lim = 0.99 if rn >= 1e-3 * big else 0.9
            if c - lim < wcos - (0.99 if wck == "" else wlim):
                wcos, wck, wlim = c, k, lim
            assert c >= lim, f"{k}: cosine {c:.4f} < {lim} (|ref| = {rn:.3e}, largest {big:.3e})"
    print(f"[16bit] {tag} {dt}: rel-L2 y {ey:.2e} dx {edx:.2e} worst parameter gradient {worst:.2e} ({wk}); "
          f"tightest cosine {wcos:.5f} ({wck})")
    record_property("rel_l2", dict(y=ey, dx=edx, grad=worst, grad_key=wk, min_cos=wcos, min_cos_key=wck))
    assert ey <= ly and edx <= ldx and worst <= lg, (ey, edx, worst, wk)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=lambda v: str(v).replace("torch.", ""))
def test_cfg1_block_inference_forward_within_the_16bit_tolerance_of_the_reference(dt, monkeypatch):
    """(round 6) BASELINE.json configs[0] -- the reference's own CPU-runnable case: ONE OSS block, (2, 48, 48, 48) -- forward only under
    ``no_grad`` on a 16-bit stream (as the blocks see it inside the nets under autocast).  This is the path that takes the inference-only
    forms: the EFFN half as ONE launch (csrc/oss_effn.hip), the streaming gate, the pooled sums.  Held to the stated 16-bit output
    tolerance against the reference's fp32 output (fp16: the fixture input is fp16-exact; bf16: + the rounding of the input itself, so
    twice the limit) and to the launch-per-layer chain on the same input at twice the limit.  (The other G3 fixtures are 10 / 12
    pixels wide: the one-launch form needs rows of a multiple of 8 pixels -- its own fixtures are G9, tests/test_effn_gpu.py.)"""
    from vmambair_amd import oss_block
    from vmambair_amd.ops import ffn as ffn_ops
    z, m = _load("cfg1")
    x = z["x"].float().to(DEV).to(dt)
    s = int(z["io_stride"])
    took = []
    real_ok = oss_block.effn_fwd_ok
    monkeypatch.setattr(oss_block, "effn_fwd_ok", lambda t, h: (took.append(real_ok(t, h)) or took[-1]))
    with torch.no_grad(), torch.autocast("cuda", dtype=dt):
        y = m(x)
        ffn_ops.EFFN_FUSED = False
        try:
            y_chain = m(x)
        finally:
            ffn_ops.EFFN_FUSED = True
    assert took and took[0], "the one-launch EFFN forward takes d = 48 on a 16-bit stream"
    ey = rel_l2(y[..., ::s, ::s], z["y"])
    ec = rel_l2(y, y_chain)
    print(f"[inference] cfg1 {dt}: rel-L2 vs the reference {ey:.2e}, vs the chain {ec:.2e}")
    assert ey <= LIMITS[dt][0] * (2 if dt == torch.bfloat16 else 1), ey
    assert ec <= 2 * LIMITS[dt][0], ec


# ------------------------------------------------------------------------------------------------------------------
# configs[3]: sequence lengths of the Deraining workload
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,itype", [(16384, torch.float32), (16384, torch.bfloat16), (147456, torch.float32), (65536, torch.bfloat16),
                                     (262144, torch.float16)],
                         ids=["L16384-f32", "L16384-bf16", "L147456-f32", "L65536-bf16", "L262144-f16"])
def test_scan_at_deraining_lengths(L, itype):
    """u (1, 192, L), B/C (1, 4, 16, L): encoder level 1 of Mamber32 at 128x128, at the 256x256 stage and at the 384x384 patches the
    progressive schedule ends on (Deraining_mamber33.yml:27-30); (round 6) L = 262 144 in fp16: the scan rows of the RealSR net on an
    UNTILED 512 x 512 image, the reference's default tile = 0 (RealSR/VmambaIR/utils.py:32).  Every output and gradient against the
    oracle (the reference test's tolerances)"""
    from test_scan_gpu import check_fwd_bwd, make_inputs
    check_fwd_bwd(make_inputs(1, 192, 16, 4, L, itype, seed=3, delta_scale=0.5), True, itype)


def test_scan_long_sequence_properties():
    """(2, 384, 16384) -- the 5x u(1,384,16384) calls of config 4 at batch 2: causality at an x-chunk boundary, linearity
    in dout, bit-stable reruns, workspace query covers the call"""
    import vmambair_amd
    from vmambair_amd import _capi
    torch.manual_seed(1)
    Bsz, KD, N, G, L = 2, 384, 16, 4, 16384
    u = torch.randn(Bsz, KD, L, device=DEV)
    delta = 0.5 * torch.rand(Bsz, KD, L, device=DEV)
    A = -0.5 * torch.rand(KD, N, device=DEV)
    Bm, Cm = torch.randn(Bsz, G, N, L, device=DEV), torch.randn(Bsz, G, N, L, device=DEV)
    D, bias = torch.randn(KD, device=DEV), 0.5 * torch.rand(KD, device=DEV)
    h = 256 * 23
    lib = _capi.load()

    def both():
        o, xs = vmambair_amd.selective_scan_fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
        oh, xh = vmambair_amd.selective_scan_fwd(u[..., :h].contiguous(), delta[..., :h].contiguous(), A, Bm[..., :h].contiguous(),
                                                  Cm[..., :h].contiguous(), D, bias, True, 1)
        return o, xs, oh, xh
    # one workgroup per row tile walking the whole row (the reference's order of operations): the prefix is bit-identical
    lib.oss_scan_set_segments(1, 1)
    try:
        out, x, out_h, x_h = both()
        assert torch.equal(out[..., :h], out_h) and torch.equal(x[:, :, :23], x_h)
    finally:
        lib.oss_scan_set_segments(-1, -1)
    # default launch: this call is cut into time segments (768 rows leave CUs idle) and the two lengths are cut differently;
    # a segment's entering state is a fold of segment-local states, so causality holds to fp32 round-off instead of bit for bit
    out, x, out_h, x_h = both()
    assert lib.oss_scan_last_segments(0) >= 1
    assert_close(out[..., :h], out_h, 1e-5, 1e-5 * float(out_h.abs().max()), "causality (segmented launch)")
    assert_close(x[:, :, :23, 1::2], x_h[..., 1::2], 1e-5, 1e-5 * float(x_h[..., 1::2].abs().max()), "saved states")
    dout = torch.randn(Bsz, KD, L, device=DEV)
    g1 = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    g1b = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1)
    assert all(torch.equal(a, b) for a, b in zip(g1, g1b))
    g2 = vmambair_amd.selective_scan_bwd(u, delta, A, Bm, Cm, D, bias, 2 * dout, x, True, 1)
    for a, b, n in zip(g1, g2, ["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"]):
        assert_close(b, 2 * a, 2e-3, 1e-3 * float(a.abs().max()) + 1e-3, "bwd linearity " + n)
    assert int(_capi.load().oss_scan_bwd_workspace_bytes(Bsz, KD, L, N, G)) < (2 << 30)


@pytest.mark.parametrize("hw", [128, 384])
def test_mamber32_block_at_128x128_matches_cpu_twin(hw):
    """one Deraining OSS block (dim 48, additive channel gate) on a 128x128 patch -- and (round 6) on the 384x384 patch the
    progressive schedule ends on (L = 147 456: time-segmented scans both ways, the depth-wise kernels' separate forms) --
    forward + all gradients, HIP vs the CPU oracle twins of every op (oracle/cpu_twins.py)"""
    from conftest import install_oracle_cpu_kernel
    install_oracle_cpu_kernel()
    torch.manual_seed(4)
    m = MamberBlock(48, variant="mamber32")
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith(("body.weight", "body.bias", "Ds", "Dsc")):
                p_.add_(0.1 * torch.randn_like(p_))
    x = torch.randn(1, 48, hw, hw)
    dy = torch.randn(1, 48, hw, hw)
    xc = x.clone().requires_grad_()
    yc = m(xc)
    yc.backward(dy)
    want = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    m.to(DEV)
    xg = x.to(DEV).requires_grad_()
    yg = m(xg)
    yg.backward(dy.to(DEV))
    assert_close(yg, yc, 1e-3, 1e-3 * float(yc.abs().max()), f"block output at L = {hw * hw}")
    assert_close(xg.grad, xc.grad, 3e-3, 3e-3 * float(xc.grad.abs().max()), "input grad")
    for k, p in m.named_parameters():
        if k.endswith("conv_cout.bias"):
            continue
        assert_close(p.grad, want[k], 5e-3, 2e-3 * max(1.0, float(want[k].abs().max())), f"grad {k}")


@pytest.mark.parametrize("hw", [272, 512])
def test_realsr_block_at_the_272x272_tile_matches_cpu_twin(hw):
    """(round 5) configs[4] with tile 256 + halo 16: one RealSR OSS block (dim 48, rank-R channel scan) on a 272 x 272 tile -- L = 73 984,
    the forward scan in two time segments with its local pass in pieces, depth-wise kernels at W / 8 = 34 lane groups -- inference
    forward in fp32 and under fp16 autocast against the CPU oracle twins; (round 6) 512 x 512: the reference's default tile = 0
    (RealSR/VmambaIR/utils.py:32) gives the first level of the net the WHOLE image, L = 262 144"""
    from conftest import install_oracle_cpu_kernel
    install_oracle_cpu_kernel()
    torch.manual_seed(6)
    m = MamberBlock(48, variant="realsr").eval()
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith(("body.weight", "body.bias", "Ds", "Dsc")):
                p_.add_(0.1 * torch.randn_like(p_))
        x = torch.randn(1, 48, hw, hw)
        want = m(x)
        m.to(DEV)
        got = m(x.to(DEV))
        assert_close(got, want, 1e-3, 1e-3 * float(want.abs().max()), f"fp32 block output at L = {hw * hw}")
        with torch.autocast("cuda", dtype=torch.float16):
            got16 = m(x.to(DEV))
        e = rel_l2(got16, want)
        print(f"[realsr tile {hw}] fp16 autocast rel-L2 {e:.2e} (limit {LIMITS[torch.float16][0]:.0e})")
        assert e <= LIMITS[torch.float16][0], e


# ------------------------------------------------------------------------------------------------------------------
# configs[1]: the whole dim-48 net, bf16 autocast against fp32 on the same HIP kernels and against the CPU twins
# ------------------------------------------------------------------------------------------------------------------
def test_whole_net_dim48_fp32_vs_cpu_twin_and_bf16_vs_fp32():
    """MambaSISR6 dim 48 [2,1,1,1]+2 (all four widths 48..384 and the x4 tail), batch 2, 64x64 LQ, L1 loss step:
    (a) HIP fp32 vs CPU oracle twins: output 2e-3 of max|y|, every parameter gradient rel-L2 <= 2e-3 (measured 1e-4);
    (b) bf16 autocast (what bench.py times) vs HIP fp32: output rel-L2 <= 1.5e-2 (measured 5.9e-3), loss within 5e-3
        relative (2.5e-3), gradient of the whole parameter vector rel-L2 <= 4e-2 (1.1e-2) and cosine >= 0.999."""
    from conftest import install_oracle_cpu_kernel
    from vmambair_amd.archs import MambaSISR6
    import torch.nn.functional as F
    install_oracle_cpu_kernel()
    torch.manual_seed(5)
    net = MambaSISR6(dim=48, num_blocks=[2, 1, 1, 1], num_refinement_blocks=2)
    lq, gt = torch.rand(2, 3, 64, 64), torch.rand(2, 3, 256, 256)

    def step(n, a, b, acdt=None):
        n.zero_grad()
        with torch.autocast(a.device.type, dtype=acdt, enabled=acdt is not None):
            out = n(a)
        loss = F.l1_loss(out.float(), b)
        loss.backward()
        return out.detach().float().cpu(), float(loss), {k: p.grad.detach().float().cpu().clone() for k, p in n.named_parameters()}

    y_c, l_c, g_c = step(net, lq, gt)
    net.to(DEV)
    y_f, l_f, g_f = step(net, lq.to(DEV), gt.to(DEV))
    assert_close(y_f, y_c, 2e-3, 2e-3 * float(y_c.abs().max()), "fp32 net output vs CPU twins")
    assert abs(l_f - l_c) <= 1e-4 * abs(l_c)
    bad = [(k, rel_l2(g_f[k], g_c[k])) for k in g_c if not k.endswith("conv_cout.bias") and float(g_c[k].norm()) > 1e-7]
    worst = max(bad, key=lambda t: t[1])
    print(f"[net] fp32 HIP vs CPU twins: worst parameter-gradient rel-L2 {worst[1]:.2e} ({worst[0]})")
    assert worst[1] <= 2e-3, worst
    y_b, l_b, g_b = step(net, lq.to(DEV), gt.to(DEV), torch.bfloat16)
    ey = rel_l2(y_b, y_f)
    keys = [k for k in g_f if not k.endswith("conv_cout.bias")]
    vf, vb = torch.cat([g_f[k].reshape(-1) for k in keys]), torch.cat([g_b[k].reshape(-1) for k in keys])
    eg, cos = rel_l2(vb, vf), float(F.cosine_similarity(vb, vf, dim=0))
    print(f"[net] bf16 vs fp32: output rel-L2 {ey:.2e}, loss {l_b:.6f} vs {l_f:.6f}, gradient rel-L2 {eg:.2e}, cosine {cos:.5f}")
    assert ey <= 1.5e-2 and abs(l_b - l_f) <= 5e-3 * abs(l_f) and eg <= 4e-2 and cos >= 0.999, (ey, l_b, l_f, eg, cos)
