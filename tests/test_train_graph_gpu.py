"""The captured training step beyond the headline config (round 2): the Deraining step (AdamW + clip_grad_norm_ 0.01, no EMA:
Deraining/basicsr/models/image_restoration_model.py:121-167), one graph per input shape for the progressive patch schedule
(Deraining/basicsr/train.py:213-271), and the two-rank RCCL exchange (self-skips with fewer than two GPUs)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
selfcheck = pytest.mark.selfcheck   # graph replay vs the eager loop of this repo: collected last (tests/conftest.py)
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_net(seed=0, dim=8):
    from vmambair_amd.archs import Mamber32
    torch.manual_seed(seed)
    return Mamber32(dim=dim, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)


def eager_deraining_step(net, opt, lq, gt):
    opt.zero_grad(set_to_none=True)
    loss = F.l1_loss(net(lq), gt)
    loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(net.parameters(), 0.01)
    opt.step()
    return float(loss), float(norm)


def test_adamw_clip_step_matches_torch():
    """fused launch with weight decay + device-side clip coefficient vs torch.optim.AdamW after clip_grad_norm_"""
    from vmambair_amd.optim import FusedAdamEMA
    torch.manual_seed(0)
    shapes = [(5,), (3, 7), (2049,), (48, 96, 1, 1), (1,), (4096,)]
    pa = [torch.randn(s, device=DEV) for s in shapes]
    pb = [p.clone().requires_grad_() for p in pa]
    opt = torch.optim.AdamW(pb, lr=3e-2, betas=(0.9, 0.999), weight_decay=1e-2)
    fo = FusedAdamEMA(pa, None, lr=3e-2, betas=(0.9, 0.999), ema_decay=0.0, weight_decay=1e-2, clip_grad_norm=0.5)
    for step in range(4):
        grads = [torch.randn(s, device=DEV) * (0.02 if step == 2 else 1.0) for s in shapes]   # step 2: below the clip norm
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g.clone(), g.clone()
        want_norm = float(torch.nn.utils.clip_grad_norm_(pb, 0.5))
        fo.step()
        opt.step()
        assert float(fo.total_norm) == pytest.approx(want_norm, rel=1e-5)
        assert float(fo.grad_scale) == pytest.approx(min(1.0, 0.5 / (want_norm + 1e-6)), rel=1e-5)
    for p, q in zip(pa, pb):
        assert torch.allclose(p, q.detach(), rtol=1e-5, atol=1e-6)


@selfcheck
def test_graphed_deraining_step_matches_eager():
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(3)
    lq, gt = torch.rand(2, 3, 32, 32, device=DEV), torch.rand(2, 3, 32, 32, device=DEV)
    net_g, net_e = make_net(), make_net()
    step = GraphedTrainStep(net_g, lr=3e-4, betas=(0.9, 0.999), ema_decay=0.0, autocast_dtype=None, warmup=1,
                            loss_fn=F.l1_loss, weight_decay=1e-4, clip_grad_norm=0.01)
    assert step.ema is None
    opt = torch.optim.AdamW(net_e.parameters(), lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    for i in range(3):
        lg = float(step(lq, gt))
        le, norm = eager_deraining_step(net_e, opt, lq, gt)
        assert lg == pytest.approx(le, rel=2e-3), i
        assert float(step.fopt.total_norm) == pytest.approx(norm, rel=5e-3)
    assert norm > 0.01, "the clip must be active for the test to mean anything"
    for (k, p), q in zip(net_g.named_parameters(), net_e.parameters()):
        assert float((p - q).abs().max()) <= 2 * 3e-4 * 3 + 1e-5, k


@selfcheck
def test_one_graph_per_shape_progressive_schedule():
    """patch size and batch change during Deraining training: the step keeps one forward+backward graph per shape and ONE
    optimizer graph; alternating shapes gives the same trajectory as the eager loop"""
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(4)
    batches = [(torch.rand(2, 3, 32, 32, device=DEV), torch.rand(2, 3, 32, 32, device=DEV)),
               (torch.rand(1, 3, 48, 40, device=DEV), torch.rand(1, 3, 48, 40, device=DEV))]
    net_g, net_e = make_net(1), make_net(1)
    step = GraphedTrainStep(net_g, lr=3e-4, betas=(0.9, 0.999), ema_decay=0.0, autocast_dtype=None, warmup=1,
                            weight_decay=1e-4, clip_grad_norm=0.01, multi_shape=True)
    opt = torch.optim.AdamW(net_e.parameters(), lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    order = [0, 1, 0, 0, 1, 1, 0]
    for i in order:
        lg = float(step(*batches[i]))
        le, _ = eager_deraining_step(net_e, opt, *batches[i])
        assert lg == pytest.approx(le, rel=3e-3), i
    assert step.n_graphs == 2 and step.graph_opt is not None
    for (k, p), q in zip(net_g.named_parameters(), net_e.parameters()):
        assert float((p - q).abs().max()) <= 2 * 3e-4 * len(order) + 1e-5, k
    # without multi_shape a second shape is an error, not a silent eager fallback
    single = GraphedTrainStep(make_net(2), autocast_dtype=None, warmup=1)
    single(*batches[0])
    with pytest.raises(RuntimeError):
        single(*batches[1])


@selfcheck
def test_warmup_leaves_no_trace():
    """capture() warms up with real steps and restores parameters, EMA, moments and the step count"""
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(0)
    net = MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)
    before = [p.detach().clone() for p in net.parameters()]
    step = GraphedTrainStep(net, autocast_dtype=None, warmup=3)
    step.capture(torch.rand(2, 3, 16, 16, device=DEV), torch.rand(2, 3, 64, 64, device=DEV))
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))
    assert all(torch.equal(a, e) for a, e in zip(before, step.ema))
    assert float(step.fopt.state[0]) == 0.0 and all(float(m.abs().max()) == 0.0 for m in step.fopt.exp_avg)


# ---------------------------------------------------------------------------------------------------------------------
# learning-rate schedule under graph replay, training-state save / resume
# ---------------------------------------------------------------------------------------------------------------------
@selfcheck
def test_set_lr_reaches_the_captured_optimizer_launch():
    """the reference changes the rate during a run (update_learning_rate, Deraining/basicsr/models/base_model.py:183-205); the
    fused launch reads it from device memory, so a replayed graph follows ``set_lr``: trajectory = eager AdamW driven by the
    reference's own scheduler object"""
    from vmambair_amd import lr_schedule
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(5)
    lq, gt = torch.rand(2, 3, 32, 32, device=DEV), torch.rand(2, 3, 32, 32, device=DEV)
    net_g, net_e = make_net(7), make_net(7)
    kw = dict(periods=[3, 5], restart_weights=[1, 0.5], eta_mins=[3e-4, 1e-5])
    step = GraphedTrainStep(net_g, lr=3e-4, betas=(0.9, 0.999), ema_decay=0.0, autocast_dtype=None, warmup=1,
                            weight_decay=1e-4, clip_grad_norm=0.01)
    opt = torch.optim.AdamW(net_e.parameters(), lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    lrs = []
    for it in range(1, 8):
        lr = lr_schedule.cosine_restart_cyclic(it, 3e-4, **kw)
        lrs.append(lr)
        step.set_lr(lr)
        for g in opt.param_groups:
            g["lr"] = lr
        lg = float(step(lq, gt))
        le, _ = eager_deraining_step(net_e, opt, lq, gt)
        assert lg == pytest.approx(le, rel=3e-3), it
    assert len(set(lrs)) > 3 and step.iteration == 7 and float(step.fopt.state[3]) == pytest.approx(lrs[-1])
    for (k, p), q in zip(net_g.named_parameters(), net_e.parameters()):
        assert float((p - q).abs().max()) <= 2 * 3e-4 * 7 + 1e-5, k
    # a constant-rate twin must end somewhere else: the schedule really acted
    net_c = make_net(7)
    const = GraphedTrainStep(net_c, lr=3e-4, betas=(0.9, 0.999), ema_decay=0.0, autocast_dtype=None, warmup=1,
                             weight_decay=1e-4, clip_grad_norm=0.01)
    for _ in range(7):
        const(lq, gt)
    assert max(float((p - q).abs().max()) for p, q in zip(net_c.parameters(), net_g.parameters())) > 1e-5


def test_fused_adam_state_dict_round_trips_with_torch_adam():
    """``FusedAdamEMA.state_dict()`` has torch.optim.Adam's layout: a torch optimizer loads it and continues identically, and
    the fused optimizer loads a torch state (what the reference's .state files hold, base_model.py:312-351)"""
    from vmambair_amd.optim import FusedAdamEMA
    torch.manual_seed(1)
    shapes = [(7,), (3, 5), (2050,), (1,)]
    pa = [torch.randn(s, device=DEV) for s in shapes]
    fo = FusedAdamEMA(pa, None, lr=1e-2, betas=(0.9, 0.99), ema_decay=0.0)
    grads = [[torch.randn(s, device=DEV) for s in shapes] for _ in range(6)]
    for k in range(3):
        for p, g in zip(pa, grads[k]):
            p.grad = g.clone()
        fo.step()
    sd = fo.state_dict()
    assert float(sd["state"][0]["step"]) == 3.0 and sd["param_groups"][0]["lr"] == 1e-2
    # torch continues from the fused state
    pb = [p.detach().clone().requires_grad_() for p in pa]
    topt = torch.optim.Adam(pb, lr=123.0, betas=(0.5, 0.5))
    topt.load_state_dict(sd)
    # ... and a fresh fused optimizer continues from torch's state
    pc = [p.detach().clone() for p in pa]
    fo2 = FusedAdamEMA(pc, None, lr=5.0, betas=(0.9, 0.99), ema_decay=0.0)
    fo2.load_state_dict(topt.state_dict())
    assert fo2.lr == 1e-2 and float(fo2.state[0]) == 3.0
    for k in range(3, 6):
        for p, q, r, g in zip(pa, pb, pc, grads[k]):
            p.grad, q.grad, r.grad = g.clone(), g.clone(), g.clone()
        fo.step(); topt.step(); fo2.step()
    for p, q, r in zip(pa, pb, pc):
        assert torch.allclose(p, q.detach(), rtol=1e-5, atol=1e-6) and torch.equal(p, r)


@selfcheck
def test_training_state_save_and_resume(tmp_path):
    """save_training_state / resume_training (base_model.py:312-351): a run interrupted after 2 steps and resumed in a NEW
    process-like object (fresh net from the saved weights, fresh graphs) ends where the uninterrupted run ends"""
    from vmambair_amd import checkpoint
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(2)
    lq, gt = torch.rand(2, 3, 16, 16, device=DEV), torch.rand(2, 3, 64, 64, device=DEV)

    def fresh():
        torch.manual_seed(11)
        return MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)
    net_a = fresh()
    a = GraphedTrainStep(net_a, autocast_dtype=None, warmup=1)
    for _ in range(4):
        a(lq, gt)
    net_b = fresh()
    b = GraphedTrainStep(net_b, autocast_dtype=None, warmup=1)
    for _ in range(2):
        b(lq, gt)
    path = checkpoint.save_training_state(b, str(tmp_path / "training_states"), epoch=0, current_iter=2)
    assert path.endswith("2.state") and checkpoint.save_training_state(b, str(tmp_path), 0, -1) is None
    wpath = checkpoint.save_network(net_b, str(tmp_path / "net_g_2.pth"))
    net_c = fresh()
    with torch.no_grad():
        for p in net_c.parameters():
            p.add_(1.0)                       # definitely not the saved weights
    checkpoint.load_network(net_c, wpath, strict=True)
    c = GraphedTrainStep(net_c, autocast_dtype=None, warmup=1)
    c.capture(lq, gt)                         # graphs exist BEFORE the resume: tensors are restored in place
    where = checkpoint.resume_training(c, path)
    assert where == {"epoch": 0, "iter": 2} and c.iteration == 2
    for _ in range(2):
        c(lq, gt)
    for (k, p), q, e1, e2 in zip(net_a.named_parameters(), net_c.parameters(), a.ema, c.ema):
        if k.endswith("conv_cout.bias"):
            continue   # exact gradient 0 (a constant in front of a LayerNorm): Adam turns the rounding noise into +-lr steps
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), k
        assert torch.allclose(e1, e2, rtol=1e-5, atol=1e-6), k


@selfcheck
def test_second_shape_capture_keeps_the_torch_optimizer_state():
    """with the torch fallback optimizer (fused_optimizer=False) a second input shape captured in the middle of a
    run must not reset the Adam moments / step count"""
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(6)
    batches = [(torch.rand(2, 3, 32, 32, device=DEV), torch.rand(2, 3, 32, 32, device=DEV)),
               (torch.rand(1, 3, 48, 40, device=DEV), torch.rand(1, 3, 48, 40, device=DEV))]
    net_g, net_e = make_net(3), make_net(3)
    step = GraphedTrainStep(net_g, lr=3e-4, betas=(0.9, 0.999), ema_decay=0.0, autocast_dtype=None, warmup=2,
                            weight_decay=1e-4, multi_shape=True, fused_optimizer=False)
    assert step.opt is not None and step.fopt is None
    opt = torch.optim.AdamW(net_e.parameters(), lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    order = [0, 0, 0, 1, 0, 1]
    for n, i in enumerate(order):
        lg = float(step(*batches[i]))
        opt.zero_grad(set_to_none=True)
        loss = F.l1_loss(net_e(batches[i][0]), batches[i][1])
        loss.backward()
        opt.step()
        assert lg == pytest.approx(float(loss), rel=3e-3), n
    steps = {float(st["step"]) for st in step.opt.state.values()}
    assert steps == {float(len(order))}, steps
    for (k, p), q in zip(net_g.named_parameters(), net_e.parameters()):
        assert float((p - q).abs().max()) <= 2 * 3e-4 * len(order) + 1e-5, k
    step.set_lr(1e-5)
    assert float(step.opt.param_groups[0]["lr"]) == pytest.approx(1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# two ranks over RCCL: the flat-buffer exchange between the two graphs
# ---------------------------------------------------------------------------------------------------------------------
def _nccl_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(0)
    net = MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(dev)
    g = torch.Generator().manual_seed(9)
    lq_all, gt_all = torch.rand(4, 3, 16, 16, generator=g), torch.rand(4, 3, 64, 64, generator=g)
    sl = slice(rank * 2, rank * 2 + 2)
    step = GraphedTrainStep(net, autocast_dtype=None, warmup=1)
    assert step.split and step.world == 2
    step.time_allreduce()
    for _ in range(2):
        loss = step(lq_all[sl].to(dev), gt_all[sl].to(dev))
    torch.cuda.synchronize()
    ms = step.collect_allreduce_ms()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = bool(torch.equal(gathered[0], gathered[1]))
    if rank == 0:
        torch.save({"params": [p.detach().cpu() for p in net.parameters()], "same": same, "ms": ms, "lq": lq_all, "gt": gt_all},
                   os.path.join(tmp, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


@selfcheck
def test_two_rank_rccl_step_equals_single_process_whole_batch():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU node); the exchange itself is covered by tests/test_ddp_gloo.py")
    import tempfile
    import torch.multiprocessing as mp
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_nccl_worker, args=(2, port, tmp), nprocs=2, join=True)
        blob = torch.load(os.path.join(tmp, "r0.pt"))
    assert blob["same"], "both ranks must hold identical weights after the step"
    assert blob["ms"] is not None and blob["ms"] > 0
    torch.manual_seed(0)
    net = MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)
    step = GraphedTrainStep(net, autocast_dtype=None, warmup=1, split_graphs=True)
    for _ in range(2):
        step(blob["lq"].to(DEV), blob["gt"].to(DEV))     # whole batch of 4: mean loss = mean of the two rank means
    for p, q in zip(net.parameters(), blob["params"]):
        assert float((p.detach().cpu() - q).abs().max()) <= 2 * 2e-4 * 2 + 1e-5


@selfcheck
def test_resume_from_a_reference_style_state_needs_the_ema_weights(tmp_path):
    """a ``.state`` file as the REFERENCE writes it has no EMA weights (they live in ``net_g_<iter>.pth`` under
    ``params_ema``, Deraining/basicsr/models/base_model.py:234-244,312-334).  ``resume_training`` must not continue silently from
    freshly initialised EMA weights: it raises unless ``ema_from`` names that checkpoint -- and then restores them by name"""
    from vmambair_amd import checkpoint
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(4)
    lq, gt = torch.rand(2, 3, 16, 16, device=DEV), torch.rand(2, 3, 64, 64, device=DEV)

    def fresh():
        torch.manual_seed(12)
        return MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)
    net_b = fresh()
    b = GraphedTrainStep(net_b, autocast_dtype=None, warmup=1)
    for _ in range(3):
        b(lq, gt)
    sd = b.state_dict()
    ref_state = {"epoch": 0, "iter": 3, "optimizers": sd["optimizers"], "schedulers": [{"last_epoch": 2}]}   # no 'ema'
    names = [n for n, p in net_b.named_parameters() if p.requires_grad]
    wpath = str(tmp_path / "net_g_3.pth")
    torch.save({"params": net_b.state_dict(), "params_ema": {**net_b.state_dict(), **{n: e.clone() for n, e in zip(names, b.ema)}}}, wpath)
    net_c = fresh()
    checkpoint.load_network(net_c, wpath, strict=True)
    c = GraphedTrainStep(net_c, autocast_dtype=None, warmup=1)
    with pytest.raises(ValueError, match="params_ema"):
        checkpoint.resume_training(c, ref_state)
    assert checkpoint.resume_training(c, ref_state, ema_from=wpath) == {"epoch": 0, "iter": 3}
    for e1, e2 in zip(b.ema, c.ema):
        assert torch.equal(e1, e2)
    d = GraphedTrainStep(fresh(), autocast_dtype=None, warmup=1, ema_decay=0.0)      # a step without EMA weights takes the file as it is
    assert checkpoint.resume_training(d, ref_state)["iter"] == 3


@selfcheck
def test_train_loop_drives_the_schedule_and_saves_states(tmp_path):
    """``checkpoint.train_loop``: current_iter += 1, schedule(current_iter) -> set_lr BEFORE the step (update_learning_rate,
    base_model.py:183-205), a training state every ``save_every`` iterations"""
    from vmambair_amd import checkpoint, lr_schedule
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep
    torch.manual_seed(5)
    net = MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)
    step = GraphedTrainStep(net, autocast_dtype=None, warmup=1)
    lq, gt = torch.rand(2, 3, 16, 16, device=DEV), torch.rand(2, 3, 64, 64, device=DEV)
    seen = []
    sched = lambda it: lr_schedule.multistep(it, 2e-4, [3, 5], 0.5)   # noqa: E731
    last = checkpoint.train_loop(step, [(lq, gt)] * 10, sched, total_iters=6, save_every=2, states_dir=str(tmp_path),
                                 on_iter=lambda it, loss: seen.append((it, step.lr, float(loss))))
    assert last == 6 and step.iteration == 6
    assert [round(lr / 2e-4, 3) for _, lr, _ in seen] == [1.0, 1.0, 1.0, 0.5, 0.5, 0.25]   # milestones at last_epoch 3 and 5
    assert sorted(p.name for p in tmp_path.iterdir()) == ["2.state", "4.state", "6.state"]
    assert all(torch.isfinite(torch.tensor(l)) for _, _, l in seen)


@pytest.mark.parametrize("acdt", [None, torch.bfloat16], ids=["fp32", "bf16"])
def test_bucketed_gradient_flush_is_bit_identical_to_the_single_flush(acdt):
    """(round 6) ``grad_buckets=3``: the backward graph stops before the grouped weight-gradient launch; three small graphs each flush
    the products / finishing sums of one reverse-order bucket and pack it (between them the multi-GPU step issues that bucket's
    all-reduce on a side stream).  Same kernels on the same operands as the single flush of the two-graph step, so three training
    steps must leave bit-identical weights, EMA and losses; the buckets cover every parameter once, in hand-over order, and their
    cuts are monotone."""
    from vmambair_amd.archs import MambaSISR6
    from vmambair_amd.train_graph import GraphedTrainStep

    def make():
        torch.manual_seed(0)
        return MambaSISR6(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1).to(DEV)

    torch.manual_seed(3)
    lq = torch.rand(2, 3, 32, 32, device=DEV)
    gt = torch.rand(2, 3, 128, 128, device=DEV)
    res = []
    for buckets in (1, 3):
        net = make()
        st = GraphedTrainStep(net, autocast_dtype=acdt, warmup=2, split_graphs=True, grad_buckets=buckets)
        losses = [float(st(lq, gt)) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, [p.detach().clone() for p in net.parameters()], [e.clone() for e in st.ema], st))
    (l1, w1, e1, _), (l3, w3, e3, st3) = res
    assert l1 == l3, (l1, l3)
    for a, b in zip(w1, w3):
        assert torch.equal(a, b)
    for a, b in zip(e1, e3):
        assert torch.equal(a, b)
    fl = st3._flat
    assert fl.n_buckets == 3 and len(st3.graph_flush) == 3
    assert sorted(i for m in fl.bucket_members for i in m) == list(range(len(st3.params)))
    assert all(a[0] <= b[0] and a[1] <= b[1] for a, b in zip(st3._cuts, st3._cuts[1:])), st3._cuts
    # 16-bit: the weight-gradient products are recorded (grouped launch); fp32 launches them at once and only the finishing sums wait
    assert st3._cuts[0][0 if acdt is not None else 1] > 0, "the first bucket holds nothing recorded: nothing would overlap"
    sizes = [b - a for a, b in fl.bucket_ranges]
    assert min(sizes) > 0.1 * sum(sizes), sizes   # (the dim-16 test net: one tail convolution is 60 % of the first bucket)
