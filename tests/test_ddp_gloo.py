"""N > 1 path on the CPU: two gloo processes, image batch sharded by rank, DDP gradient mean must
equal the single-process gradient of the whole batch (the reference has no such test; its DDP is
only ever exercised on 8 real GPUs -- SURVEY.md section 4)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from conftest import install_oracle_cpu_kernel
    install_oracle_cpu_kernel()
    from vmambair_amd import ddp
    from vmambair_amd.oss_block import MamberBlock

    r, w, device = ddp.init_distributed()
    assert (r, w) == (rank, world) and device.type == "cpu"
    torch.manual_seed(0)
    net = MamberBlock(16, variant="srgan")
    ref_state = {k: v.clone() for k, v in net.state_dict().items()}
    model = ddp.wrap_ddp(net, device)
    g = torch.Generator().manual_seed(123)
    x_all = torch.randn(4, 16, 6, 5, generator=g)
    idx = ddp.shard_indices(4, rank, world, epoch=0)
    assert len(idx) == 2
    y = model(x_all[idx])
    loss = y.square().mean()
    loss.backward()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    red = ddp.reduce_loss_dict({"l_pix": loss})
    # the captured-step path: local backward without DDP hooks + ONE flat all-reduce must give the same
    net2 = MamberBlock(16, variant="srgan")
    net2.load_state_dict(ref_state)
    net2(x_all[idx]).square().mean().backward()
    ddp.allreduce_grads_flat(list(net2.parameters()))
    for (k, p2) in net2.named_parameters():
        assert torch.allclose(p2.grad, grads[k], rtol=1e-5, atol=1e-7), k
    # ... and so must the persistent flat buffer of the captured step (pack once per step, all-reduce in place,
    # the optimizer reads views of the buffer)
    net3 = MamberBlock(16, variant="srgan")
    net3.load_state_dict(ref_state)
    fg = None
    for _ in range(2):   # two steps: the buffer is built at the first and re-used
        for p3 in net3.parameters():
            p3.grad = None
        net3(x_all[idx]).square().mean().backward()
        if fg is None:
            fg = ddp.FlatGrads(list(net3.parameters()))
        fg.pack()
        fg.allreduce_mean()
        base = fg.flat.untyped_storage().data_ptr()
        for (k, p3) in net3.named_parameters():
            assert p3.grad.untyped_storage().data_ptr() == base, k
            assert torch.allclose(p3.grad, grads[k], rtol=1e-5, atol=1e-7), k
    # (round 6) the bucketed exchange of the captured step (train_graph.py: grad_buckets): the flat buffer laid out in REVERSE
    # parameter order (the order in which a backward hands gradients over), three buckets, each packed and all-reduced on its own
    # -- any interleaving of pack_bucket / allreduce_bucket over the buckets gives the gradient mean
    net4 = MamberBlock(16, variant="srgan")
    net4.load_state_dict(ref_state)
    net4(x_all[idx]).square().mean().backward()
    ps4 = list(net4.parameters())
    fb = ddp.FlatGrads(ps4, order=list(range(len(ps4)))[::-1], n_buckets=3)
    assert fb.n_buckets == 3 and sorted(i for m in fb.bucket_members for i in m) == list(range(len(ps4)))
    assert fb.bucket_ranges[0][0] == 0 and fb.bucket_ranges[-1][1] == fb.flat.numel()
    assert all(a[1] == b[0] for a, b in zip(fb.bucket_ranges, fb.bucket_ranges[1:]))
    for k in range(fb.n_buckets):
        fb.pack_bucket(k)
        fb.allreduce_bucket(k)
    for (k, p4) in net4.named_parameters():
        assert p4.grad.untyped_storage().data_ptr() == fb.flat.untyped_storage().data_ptr(), k
        assert torch.allclose(p4.grad, grads[k], rtol=1e-5, atol=1e-7), k
    # the last parameter's gradient (first to be handed over by a backward) sits at the front of the buffer: bucket 0
    assert ps4[-1].grad.data_ptr() == fb.flat.data_ptr()
    if rank == 0:
        torch.save({"grads": grads, "state": ref_state, "x": x_all, "loss": red["l_pix"]}, os.path.join(tmp, "r0.pt"))
    # both ranks hold the same averaged gradient
    flat = torch.cat([v.flatten() for v in grads.values()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    # shards are disjoint and cover the batch
    all_idx = [torch.zeros_like(idx) for _ in range(world)]
    dist.all_gather(all_idx, idx)
    assert sorted(torch.cat(all_idx).tolist()) == [0, 1, 2, 3]
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_match_single_process():
    port = 29000 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        blob = torch.load(os.path.join(tmp, "r0.pt"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import install_oracle_cpu_kernel
    install_oracle_cpu_kernel()
    from vmambair_amd.oss_block import MamberBlock
    net = MamberBlock(16, variant="srgan")
    net.load_state_dict(blob["state"])
    y = net(blob["x"])
    loss = y.square().mean()
    loss.backward()
    assert abs(float(loss) - blob["loss"]) < 1e-5 * max(1.0, abs(float(loss)))
    for k, p in net.named_parameters():
        ref = p.grad
        got = blob["grads"][k]
        tol = 1e-5 + 1e-4 * float(ref.abs().max())
        assert (got - ref).abs().max() <= tol, k


def _deraining_worker(rank, world, port, tmp):
    """the Deraining step over two ranks, in the order the captured step runs it (train_graph.py): local backward, pack into the
    flat buffer, ONE all-reduce (mean), THEN clip_grad_norm_(0.01) on the averaged gradient, AdamW
    (Deraining/basicsr/models/image_restoration_model.py:144-173 + base_model.py:76-82)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from conftest import install_oracle_cpu_kernel
    install_oracle_cpu_kernel()
    from vmambair_amd import ddp
    from vmambair_amd.oss_block import MamberBlock
    ddp.init_distributed()
    blob = torch.load(os.path.join(tmp, "in.pt"))
    per_rank = ddp.split_global_batch(blob["x"].shape[0], world)     # --global-batch: fixed total, equal shards
    sl = slice(rank * per_rank, (rank + 1) * per_rank)
    net = MamberBlock(16, variant="mamber32")
    net.load_state_dict(blob["state"])
    params = list(net.parameters())
    opt = torch.optim.AdamW(params, lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    fg, norms = None, []
    for step in range(2):
        for p in params:
            p.grad = None
        torch.nn.functional.l1_loss(net(blob["x"][sl]), blob["y"][sl]).backward()
        if fg is None:   # (round 6) the bucketed order of the captured step: reverse parameter order, three buckets
            fg = ddp.FlatGrads(params, order=list(range(len(params)))[::-1], n_buckets=3)
        local_sq = 0.0
        for k in range(fg.n_buckets):          # bucket k is packed, then exchanged (on a side stream in train_graph.py) ...
            fg.pack_bucket(k)
            local_sq += float(fg.flat[slice(*fg.bucket_ranges[k])].square().sum())   # this rank's own gradient, before the exchange
            fg.allreduce_bucket(k)
        local_norm = local_sq ** 0.5
        # ... and the clip runs only after EVERY bucket came back: on the averaged gradient of the whole net
        norms.append((local_norm, float(torch.nn.utils.clip_grad_norm_(params, 0.01))))   # on the AVERAGED gradient
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), "both ranks must take the same clipped step"
    if rank == 0:
        torch.save({"params": [p.detach().clone() for p in params], "norms": norms}, os.path.join(tmp, "out.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_deraining_step_equals_single_process_on_the_whole_batch():
    """AdamW + clip computed on the all-reduced gradient: two ranks on halves of a global batch of 4 take the same clipped
    steps as one process on all 4 images"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import install_oracle_cpu_kernel
    install_oracle_cpu_kernel()
    from vmambair_amd import ddp
    from vmambair_amd.oss_block import MamberBlock
    torch.manual_seed(0)
    net = MamberBlock(16, variant="mamber32")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(4, 16, 6, 5, generator=g), torch.randn(4, 16, 6, 5, generator=g)
    port = 31000 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as tmp:
        torch.save({"state": state, "x": x, "y": y}, os.path.join(tmp, "in.pt"))
        mp.spawn(_deraining_worker, args=(2, port, tmp), nprocs=2, join=True)
        out = torch.load(os.path.join(tmp, "out.pt"))
    params = list(net.parameters())
    opt = torch.optim.AdamW(params, lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.l1_loss(net(x), y).backward()
        n = float(torch.nn.utils.clip_grad_norm_(params, 0.01))
        opt.step()
        local, avg = out["norms"][step]
        assert avg == pytest.approx(n, rel=1e-4), "the clip saw the whole-batch gradient norm"
        assert n > 0.01, "the clip must be active for the test to mean anything"
        assert abs(local - avg) > 1e-6 * avg, "per-rank norms differ from the averaged one: clipping per rank would be wrong"
    for (k, p), q in zip(net.named_parameters(), out["params"]):
        if k.endswith("conv_cout.bias"):
            continue   # exact gradient 0 (a constant in front of a LayerNorm): Adam turns summation-order noise into +-lr steps
        assert torch.allclose(p.detach(), q, rtol=1e-5, atol=1e-6), k
    with pytest.raises(ValueError):
        ddp.split_global_batch(32, 3)
    assert [ddp.split_global_batch(32, w) for w in (1, 2, 4, 8)] == [32, 16, 8, 4]


def test_shard_indices_follow_the_reference_sampler():
    from vmambair_amd import ddp
    # data_sampler.py:36-43 with ratio 1: randperm(total) seeded by epoch, rank-strided
    g = torch.Generator()
    g.manual_seed(3)
    perm = torch.randperm(12, generator=g)
    for r in range(4):
        assert torch.equal(ddp.shard_indices(10, r, 4, epoch=3), (perm % 10)[r::4])


def test_bench_gpus_n_starts_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher around it (the reference's launcher line is
    SRGAN/train_S1.sh:1-8): the script re-executes itself under torch.distributed.run, the two ranks rendezvous on
    127.0.0.1 (gloo here: no GPU), rank 0's JSON line is the last stdout line and the exit code is the ranks'."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rendezvous-only"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["rendezvous"] == "ok" and line["n_ranks"] == 2 and line["sum_of_ranks_plus_1"] == 3.0
    assert line["master"].startswith("127.0.0.1:")
    assert "starting the ranks" in r.stderr


def test_bench_gpus_n_fails_on_the_device_count_not_on_the_launcher():
    """On a box with fewer GPUs than ranks the self-launched run must end with the device-count message (or the no-GPU
    message here), never with 'needs torch.distributed.run'."""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the run would start")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "needs torch.distributed.run" not in r.stderr
    assert ("needs a GPU" in r.stderr) or ("GPU(s) are visible" in r.stderr), r.stderr[-1500:]
