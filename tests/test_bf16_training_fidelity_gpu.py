"""bf16 autocast -- the dtype BASELINE.json configs[1] names and bench.py's headline runs -- against the reference's own precision
(it trains in fp32: no autocast / GradScaler anywhere) over a TRAINING TRAJECTORY, not one step
(VERDICT r5 next #3b: "PSNR-matching" needs the two precisions to learn the same thing, and a single-step gradient comparison
at random initial weights -- tests/test_full_depth_net.py -- cannot show that rounding errors do not compound).

MambaSISR6 dim 48 [2,1,1,1]+2 (all four level widths and the x4 tail), one seed, a fixed synthetic x4 SR set whose HR images are
smooth random fields (so that there is something to learn: LQ = area-downsampled HR), the reference's step (L1, Adam 2e-4
(0.9, 0.99), EMA 0.999; SRGAN/options/MambaSISR15_x4.yml:75-90, MambaSISR_model.py:120-147) through ``GraphedTrainStep``.

Design (the first version compared two runs from the random initial weights and learnt something else: the first ~50 Adam steps of
this net are spiky -- the CPU twins show single-step losses jumping 0.12 -> 0.65 -> 0.17 -- so ANY perturbation, a different fp32
summation order included, moves the 25-step loss means by tens of percent; that measures chaos, not precision):
  1. burn-in: 120 fp32 steps, shared by everything below (past the spikes);
  2. from that ONE state (weights, Adam moments, EMA, step count) three branches of 80 steps: fp32 again (A), bf16 autocast (B), and
     fp32 with the batch as two micro-batch branches (C: the same arithmetic in another summation order -- the control that shows
     how far two runs of the SAME precision drift apart);
  3. asserted: the loss curve of B stays with A -- the first two window means (20 steps each) within 2 %, every window within
     max(6 %, 5 x the control's drift): the branches drift apart with time whatever the precision (the control reaches 0.7 % by
     itself) -- PSNR of the EMA weights on held-out images within 0.5 dB, and the total displacement of the 80 steps points the
     same way (cosine >= 0.9 or the control's own value);
  4. teacher-forced gradients: at the branch point AND at the end of branch A the bf16-autocast gradient of the SAME weights and
     batch has cosine >= 0.98 and relative L2 error <= 0.25 against the fp32 gradient (flat vector over all parameters) -- the
     single-step agreement of test_full_depth_net.py holds at TRAINED weights too, so there is no drift for errors to compound in.
Measured on the MI355X in five runs of the round (the fp32 trajectory itself is not reproducible to the last bit from run to run --
the vendor library's 3x3 convolutions of the skeleton -- so every run is another sample; profiles/r06_pytest_gpu_final.txt is one):
bf16 / fp32 window means between 0.972 and 1.005 (first two windows 0.998 .. 1.005; control 0.994 .. 1.007), displacement cosine
0.986 .. 0.999 (control 0.952 .. 1.000), PSNR of bf16 within -0.21 .. +0.24 dB of fp32 (control within 0.001), teacher-forced gradient
cosine 0.990 .. 0.999 with relative L2 error 5.4e-2 .. 1.43e-1 (the gradient shrinks as the loss falls, the bf16 rounding of the
activations does not).  The limits leave a factor ~2 over the worst sample.
This is a self-comparison of two precisions of THIS repo (the fp32 path is what the G8 fixtures pin to the reference).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu]
DEV = "cuda:0"
BURN, BRANCH, WINDOW = 120, 80, 20


def _smooth_images(n, size, gen):
    """random low-frequency colour fields in [0, 1]: an 8x8 random grid upsampled bicubically plus a little fine texture"""
    coarse = torch.rand(n, 3, 8, 8, generator=gen)
    img = F.interpolate(coarse, size=(size, size), mode="bicubic", align_corners=False)
    img = img + 0.05 * F.interpolate(torch.rand(n, 3, size // 4, size // 4, generator=gen) - 0.5, size=(size, size), mode="bilinear",
                                     align_corners=False)
    return img.clamp(0, 1)


def _psnr(a, b):
    return 10.0 * math.log10(1.0 / float((a.float() - b.float()).square().mean()))


def _flat(net):
    return torch.cat([p.detach().flatten() for p in net.parameters()]).double()


def _make_net():
    from vmambair_amd.archs import MambaSISR6
    torch.manual_seed(0)
    return MambaSISR6(dim=48, num_blocks=[2, 1, 1, 1], num_refinement_blocks=2).to(DEV)


def _run(step, net, hr, lq, first_it, n):
    nb = hr.shape[0] // 2
    losses = []
    for it in range(first_it, first_it + n):
        k = it % nb
        losses.append(float(step(lq[2 * k:2 * k + 2], hr[2 * k:2 * k + 2])))
    return losses


def _eager_grad(net, lq, hr, acdt):
    for p in net.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=acdt is not None):
        out = net(lq)
    F.l1_loss(out.float(), hr).backward()
    g = torch.cat([p.grad.detach().flatten() for p in net.parameters()]).double()
    for p in net.parameters():
        p.grad = None
    return g


def test_bf16_autocast_training_trajectory_follows_fp32():
    from vmambair_amd.train_graph import GraphedTrainStep
    g = torch.Generator().manual_seed(11)
    hr = _smooth_images(16, 128, g).to(DEV)                 # 8 batches of 2, cycled
    lq = F.interpolate(hr, scale_factor=0.25, mode="area")
    hr_val = _smooth_images(4, 128, g).to(DEV)
    lq_val = F.interpolate(hr_val, scale_factor=0.25, mode="area")
    kw = dict(warmup=1, lr=2e-4, betas=(0.9, 0.99), ema_decay=0.999)

    # 1. burn-in (fp32)
    net0 = _make_net()
    st0 = GraphedTrainStep(net0, autocast_dtype=None, **kw)
    burn = _run(st0, net0, hr, lq, 0, BURN)
    weights = [p.detach().clone() for p in net0.parameters()]
    state = st0.state_dict()
    w_branch = _flat(net0)

    def cos(a, b):
        return float((a * b).sum() / (a.norm() * b.norm()))

    def grad_agreement(net, tag):
        gf = _eager_grad(net, lq[:2], hr[:2], None)
        gb = _eager_grad(net, lq[:2], hr[:2], torch.bfloat16)
        c, e = cos(gb, gf), float((gb - gf).norm() / gf.norm())
        print(f"[bf16 vs fp32 trajectory] teacher-forced gradient {tag}: cosine {c:.5f}, rel-L2 {e:.3e}")
        return c, e

    ga = grad_agreement(net0, f"at the branch point (step {BURN})")

    # 2. three branches from the same state
    def branch(acdt, micro):
        net = _make_net()
        with torch.no_grad():
            for p, w in zip(net.parameters(), weights):
                p.copy_(w)
        st = GraphedTrainStep(net, autocast_dtype=acdt, micro_streams=micro, **kw)
        st.capture(lq[:2], hr[:2])             # capture first (its warm-up step is undone), THEN load the trained state
        with torch.no_grad():
            for p, w in zip(net.parameters(), weights):
                p.copy_(w)
        st.load_state_dict(state)
        losses = _run(st, net, hr, lq, BURN, BRANCH)
        live = [p.detach().clone() for p in net.parameters()]
        with torch.no_grad():                  # validation with the EMA weights, as the reference's validation does ('params_ema')
            for p, e in zip(net.parameters(), st.ema):
                p.copy_(e)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=acdt is not None):
                out = net(lq_val)
            psnr = _psnr(out.clamp(0, 1), hr_val)
            for p, l in zip(net.parameters(), live):
                p.copy_(l)
        return net, losses, _flat(net) - w_branch, psnr

    netA, lA, dA, pA = branch(None, 1)
    netB, lB, dB, pB = branch(torch.bfloat16, 1)
    netC, lC, dC, pC = branch(None, 2)
    ge = grad_agreement(netA, f"at the end of the fp32 branch (step {BURN + BRANCH})")

    def windows(l):
        return [sum(l[i:i + WINDOW]) / WINDOW for i in range(0, BRANCH, WINDOW)]
    wA, wB, wC = windows(lA), windows(lB), windows(lC)
    rB, rC = [b / a for a, b in zip(wA, wB)], [c / a for a, c in zip(wA, wC)]
    drift_c = max(abs(r - 1.0) for r in rC)
    lim = max(0.06, 5.0 * drift_c)
    print(f"[bf16 vs fp32 trajectory] burn-in loss {burn[0]:.4f} -> {sum(burn[-WINDOW:]) / WINDOW:.4f}; branch windows of {WINDOW} steps: "
          f"fp32 {[round(v, 5) for v in wA]} bf16 {[round(v, 5) for v in wB]} fp32-other-order {[round(v, 5) for v in wC]}; "
          f"bf16 / fp32 {[round(r, 4) for r in rB]} control / fp32 {[round(r, 4) for r in rC]} (limit +-{lim:.3f}); displacement cosine "
          f"bf16 {cos(dB, dA):.4f} control {cos(dC, dA):.4f}; |displacement| fp32 {float(dA.norm()):.4f} bf16 {float(dB.norm()):.4f}; "
          f"EMA-weights PSNR on held-out images fp32 {pA:.3f} dB bf16 {pB:.3f} dB control {pC:.3f} dB")
    assert sum(burn[-WINDOW:]) / WINDOW < 0.5 * sum(burn[:WINDOW]) / WINDOW, "the set is learnable: the loss must fall, or nothing is compared"
    for r in rB[:2]:
        assert abs(r - 1.0) <= 0.02, (rB, rC)
    for r in rB:
        assert abs(r - 1.0) <= lim, (rB, rC)
    assert abs(pB - pA) <= max(0.5, 3.0 * abs(pC - pA)), (pA, pB, pC)
    assert cos(dB, dA) >= min(0.9, cos(dC, dA) - 0.05), (cos(dB, dA), cos(dC, dA))
    assert abs(float(dB.norm()) / float(dA.norm()) - 1.0) <= 0.05
    for c, e in (ga, ge):
        assert c >= 0.98 and e <= 0.25, (c, e)
