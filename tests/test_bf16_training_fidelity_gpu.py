"""bf16 autocast -- the dtype BASELINE.json configs[1] names and bench.py's headline runs -- against the reference's own precision
(it trains in fp32: no autocast / GradScaler anywhere) over a TRAINING TRAJECTORY, not one step
(VERDICT r5 next #3b: "PSNR-matching" needs the two precisions to learn the same thing, and a single-step gradient comparison
-- tests/test_full_depth_net.py -- cannot show that rounding errors do not compound).

MambaSISR6 dim 48 [2,1,1,1]+2 (all four level widths and the x4 tail), one seed, a fixed synthetic x4 SR set whose HR images are
smooth random fields (so that there is something to learn: LQ = area-downsampled HR), 200 optimizer steps of the reference's step
(L1, Adam 2e-4 (0.9, 0.99), EMA 0.999; SRGAN/options/MambaSISR15_x4.yml:75-90, MambaSISR_model.py:120-147) through
``GraphedTrainStep`` once in fp32 and once under bf16 autocast.

Compared (limits next to the asserts; the measured values are printed for the record):
  * the loss curves, window by window (mean of 25 steps): |bf16 / fp32 - 1| <= 2 %;
  * PSNR of the final EMA weights on a held-out batch: |difference| <= 0.1 dB;
  * the first update from identical weights: cosine of the two update vectors >= 0.99;
  * the total displacement w_200 - w_0 of all weights: cosine >= 0.95.
This is a self-comparison of two precisions of THIS repo (the fp32 path is what the G8 fixtures pin to the reference).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu]
DEV = "cuda:0"
STEPS, WINDOW = 200, 25


def _smooth_images(n, size, gen):
    """random low-frequency colour fields in [0, 1]: an 8x8 random grid upsampled bicubically plus a little fine texture"""
    coarse = torch.rand(n, 3, 8, 8, generator=gen)
    img = F.interpolate(coarse, size=(size, size), mode="bicubic", align_corners=False)
    img = img + 0.05 * F.interpolate(torch.rand(n, 3, size // 4, size // 4, generator=gen) - 0.5, size=(size, size), mode="bilinear",
                                     align_corners=False)
    return img.clamp(0, 1)


def _psnr(a, b):
    return 10.0 * math.log10(1.0 / float((a.float() - b.float()).square().mean()))


def _train(acdt, hr, lq, hr_val, lq_val):
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(0)
    net = MambaSISR6(dim=48, num_blocks=[2, 1, 1, 1], num_refinement_blocks=2).to(DEV)
    w0 = torch.cat([p.detach().flatten() for p in net.parameters()]).double()
    step = GraphedTrainStep(net, autocast_dtype=acdt, warmup=1, lr=2e-4, betas=(0.9, 0.99), ema_decay=0.999)
    nb = hr.shape[0] // 2
    losses, first = [], None
    for it in range(STEPS):
        k = it % nb
        losses.append(float(step(lq[2 * k:2 * k + 2], hr[2 * k:2 * k + 2])))
        if it == 0:
            first = torch.cat([p.detach().flatten() for p in net.parameters()]).double() - w0
    w = torch.cat([p.detach().flatten() for p in net.parameters()]).double()
    # validation with the EMA weights, as the reference's validation does (param_key 'params_ema')
    live = [p.detach().clone() for p in net.parameters()]
    with torch.no_grad():
        for p, e in zip(net.parameters(), step.ema):
            p.copy_(e)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=acdt is not None):
            out = net(lq_val)
        psnr = _psnr(out.clamp(0, 1), hr_val)
        for p, l in zip(net.parameters(), live):
            p.copy_(l)
    return losses, first, w - w0, psnr


def test_bf16_autocast_training_trajectory_follows_fp32():
    g = torch.Generator().manual_seed(11)
    hr = _smooth_images(16, 128, g).to(DEV)                 # 8 batches of 2, cycled: 25 passes over the set
    lq = F.interpolate(hr, scale_factor=0.25, mode="area")
    hr_val = _smooth_images(4, 128, g).to(DEV)
    lq_val = F.interpolate(hr_val, scale_factor=0.25, mode="area")
    l32, f32_first, d32, p32 = _train(None, hr, lq, hr_val, lq_val)
    l16, f16_first, d16, p16 = _train(torch.bfloat16, hr, lq, hr_val, lq_val)

    def cos(a, b):
        return float((a * b).sum() / (a.norm() * b.norm()))
    win32 = [sum(l32[i:i + WINDOW]) / WINDOW for i in range(0, STEPS, WINDOW)]
    win16 = [sum(l16[i:i + WINDOW]) / WINDOW for i in range(0, STEPS, WINDOW)]
    ratios = [b / a for a, b in zip(win32, win16)]
    c_first, c_total = cos(f16_first, f32_first), cos(d16, d32)
    print(f"[bf16 vs fp32 trajectory] windows of {WINDOW} steps: fp32 {[round(v, 5) for v in win32]} bf16 {[round(v, 5) for v in win16]} "
          f"ratio {[round(r, 4) for r in ratios]}; first-update cosine {c_first:.5f}; displacement cosine after {STEPS} steps {c_total:.5f}; "
          f"|displacement| fp32 {float(d32.norm()):.4f} bf16 {float(d16.norm()):.4f}; EMA-weights PSNR on held-out images fp32 {p32:.3f} dB "
          f"bf16 {p16:.3f} dB")
    assert win32[-1] < 0.8 * win32[0], "the set is learnable: the fp32 loss must fall, or the comparison says nothing"
    for r in ratios:
        assert abs(r - 1.0) <= 0.02, ratios
    assert abs(p16 - p32) <= 0.1, (p16, p32)
    assert c_first >= 0.99, c_first
    assert c_total >= 0.95, c_total
    assert abs(float(d16.norm()) / float(d32.norm()) - 1.0) <= 0.05
