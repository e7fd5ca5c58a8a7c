"""Whole-net parity at the depth bench.py times (VERDICT r3 missing #3 / next #6): MambaSISR6 dim 48 [15,1,1,1]+15
(SRGAN/options/MambaSISR15_x4.yml:55-65, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:557-643), batch 1, 64x64 LQ, fp32, one
L1-loss step -- against golden set G8: the REFERENCE's own arch file run on the same weights and inputs with
selective_scan_ref as the scan (tests/golden/make_golden.py: make_g8; weights / inputs are functions of (seed, name),
tests/conftest.py: reseed_parameters, so the fixture holds only samples of the results).  `small` = the same net at
[2,1,1,1]+2; `full32` = the full depth on a 32x32 input (a quarter of the reference's step-by-step scan: hours instead of half a day of
host time for the fixture).  CPU: the host mirrors + CPU twins (oracle/) against the reference; GPU: the HIP path against the reference, fp32
at the limits below, and bf16 autocast (what bench.py times) against the fp32 HIP run."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, reseed_parameters, seeded_tensor

HAS_GPU = torch.cuda.is_available()


def _load(tag):
    path = os.path.join(GOLDEN, f"g8_net_mambasisr6_{tag}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} has not been generated (tests/golden/make_golden.py g8)")
    return np.load(path)


def _step(z, device, acdt=None):
    from vmambair_amd.archs import MambaSISR6
    seed, hw = int(z["seed"]), int(z["hw"])
    net = MambaSISR6(inp_channels=3, out_channels=3, dim=48, num_blocks=[int(v) for v in z["num_blocks"]],
                     num_refinement_blocks=int(z["refine"]), heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
                     LayerNorm_type="WithBias")
    assert sorted(n for n, _ in net.named_parameters()) == sorted(str(n) for n in z["names"]), "parameter names differ from the reference net"
    reseed_parameters(net, seed).to(device)
    lq = seeded_tensor("g8.lq", (1, 3, hw, hw), seed).to(device).requires_grad_()
    gt = seeded_tensor("g8.gt", (1, 3, 4 * hw, 4 * hw), seed).to(device)
    with torch.autocast(torch.device(device).type, dtype=acdt, enabled=acdt is not None):
        out = net(lq)
    loss = F.l1_loss(out.float(), gt)
    loss.backward()
    grads = {k: (p.grad.detach().float().cpu() if p.grad is not None else torch.zeros(p.shape)) for k, p in net.named_parameters()}
    return out.detach().float().cpu(), float(loss), lq.grad.detach().float().cpu(), grads


def _compare(z, got, lim_y, lim_g, what):
    y, loss, dlq, grads = got
    ymax, dmax = float(z["y_absmax"]), float(z["dlq_absmax"])
    ey = float((y[..., ::8, ::8] - torch.from_numpy(z["y"])).abs().max()) / ymax
    ed = float((dlq[..., ::2, ::2] - torch.from_numpy(z["dlq"])).abs().max()) / dmax
    el = abs(loss - float(z["loss"])) / abs(float(z["loss"]))
    worst, worst_n, n_checked = (0.0, ""), (0.0, ""), 0
    num = den = 0.0
    for k in z["names"]:
        k = str(k)
        ref, (gnorm, gmax, st) = torch.from_numpy(z["grad." + k]), z["gstat." + k]
        g = grads[k].reshape(-1)
        num += float((g[::int(st)] - ref).square().sum())
        den += float(ref.square().sum())
        if k.endswith("conv_cout.bias"):   # a constant added right before a LayerNorm over the same axis (MambaSISR6_arch.py:476-479):
            continue                         # its true gradient is 0, both sides hold round-off noise (as in tests/test_configs_gpu.py)
        if gmax <= 1e-9:
            assert float(g.abs().max()) <= 1e-6, k
            continue
        n_checked += 1
        e = float((g[::int(st)] - ref).abs().max()) / float(gmax)
        en = abs(float(g.norm()) - float(gnorm)) / float(gnorm)
        worst = max(worst, (e, k))
        worst_n = max(worst_n, (en, k))
    eg = (num / max(den, 1e-30)) ** 0.5
    print(f"[g8 {what}] output {ey:.2e} of max|y|, loss rel {el:.2e}, d lq {ed:.2e} of its max; {n_checked} parameter gradients: "
          f"sampled rel-L2 over all {eg:.2e}, worst sample error {worst[0]:.2e} of the tensor's max ({worst[1]}), worst norm error "
          f"{worst_n[0]:.2e} ({worst_n[1]})")
    assert ey <= lim_y and el <= lim_y and ed <= lim_g, (ey, el, ed)
    assert eg <= lim_g and worst[0] <= 10 * lim_g and worst_n[0] <= 10 * lim_g, (eg, worst, worst_n)


@pytest.mark.parametrize("tag", ["small", "full32", "full"])
def test_cpu_twins_match_the_reference_whole_net(tag, oracle_cpu_kernel):
    """host mirrors + CPU twins (fused data flow, C oracle as the scan) vs the reference's arch + selective_scan_ref: fp32
    round-off only.  The full-depth run is ~10 s of host time per step."""
    z = _load(tag)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    _compare(z, _step(z, "cpu"), 2e-4, 2e-3, f"{tag} cpu twins")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "full32", "full"])
def test_hip_whole_net_matches_the_reference(tag):
    """the product path (fused HIP blocks, omni scans) vs the reference run: fp32, and bf16 autocast vs the fp32 HIP run"""
    z = _load(tag)
    got = _step(z, "cuda:0")
    _compare(z, got, 5e-4, 5e-3, f"{tag} hip fp32")
    y_b, l_b, d_b, g_b = _step(z, "cuda:0", torch.bfloat16)
    y_f, l_f, d_f, g_f = got
    keys = [k for k in g_f if float(g_f[k].norm()) > 0 and not k.endswith("conv_cout.bias")]
    vf = torch.cat([g_f[k].reshape(-1) for k in keys]).double()
    vb = torch.cat([g_b[k].reshape(-1) for k in keys]).double()
    ey = float((y_b - y_f).norm() / y_f.norm())
    eg, cos = float((vb - vf).norm() / vf.norm()), float((vb @ vf) / (vb.norm() * vf.norm()))
    print(f"[g8 {tag} bf16 vs fp32] output rel-L2 {ey:.2e}, loss {l_b:.6f} vs {l_f:.6f}, gradient rel-L2 {eg:.2e}, cosine {cos:.5f}")
    # Stated limits for 16-bit activations through this net: the name-seeded weights (conftest.reseed_parameters) are not a
    # trained net's -- |y| reaches 45 and every block adds its bf16 rounding to a residual stream that large -- so the limits
    # are wider than test_configs_gpu.py's for the default initialisation (measured there 5e-3 / 8e-3; here, [2,1,1,1]+2:
    # 3.1e-2 output, 1.2e-1 gradient, cosine 1.000).  The direction of the gradient is what the optimizer consumes: cosine.
    # 50 blocks deep (measured on the MI355X -- full32: output 9.3e-2, loss 10.4816 vs 10.3527 = 1.2e-2, gradient 1.2e-1, cosine 0.9994;
    # full: 4.4e-2, 13.029 vs 12.983, 6.2e-2, 0.998)
    lim_y, lim_g, lim_c, lim_l = (6e-2, 2.5e-1, 0.97, 1e-2) if tag == "small" else (1.5e-1, 2.5e-1, 0.99, 3e-2)
    assert ey <= lim_y and abs(l_b - l_f) <= lim_l * abs(l_f) and eg <= lim_g and cos >= lim_c, (ey, l_b, l_f, eg, cos)
