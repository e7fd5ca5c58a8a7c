"""Data-parallel scale-out of the OSS nets: one process per GPU, image batch sharded by rank,
gradients all-reduced (mean) by DistributedDataParallel over RCCL/xGMI.

Mirrors what the reference does and nothing more: ``init_dist`` picks the device
from the rank and calls ``init_process_group(backend='nccl')`` (*/utils/dist_util.py:10-25),
``model_to_device`` wraps the net in DDP (Deraining/basicsr/models/base_model.py:76-82), the
sampler hands rank ``r`` the indices ``r, r+world, ...`` of a seeded permutation
(Deraining/basicsr/data/data_sampler.py:30-43).  The scan itself needs no collective: rows of
different images are independent.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 48.1 MB fp32 gradient of MambaSISR6 is one
~0.1-0.6 ms all-reduce, far below a step, so the only tuning is a bucket large enough to keep the
number of RCCL calls small (``bucket_cap_mb``).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def dist_info() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun / torch.distributed.launch environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend: str | None = None) -> Tuple[int, int, torch.device]:
    """Reference ``_init_dist_pytorch``: device = rank % n_gpus, then init_process_group.
    ``backend`` defaults to "nccl" (= RCCL on ROCm) on GPUs and "gloo" on the CPU (tests)."""
    rank, world, local_rank = dist_info()
    use_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    if use_gpu:
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, device


def wrap_ddp(net: torch.nn.Module, device: torch.device, bucket_cap_mb: int = 50,
             find_unused_parameters: bool = False) -> torch.nn.Module:
    """``model_to_device`` of the reference (base_model.py:67-85) for the distributed case."""
    net = net.to(device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return net
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(net, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                                     find_unused_parameters=find_unused_parameters,
                                                     gradient_as_bucket_view=True)


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, ratio: int = 1) -> torch.Tensor:
    """EnlargedSampler.__iter__ (data_sampler.py:36-43): epoch-seeded global permutation of
    ``ceil(n*ratio/world)*world`` indices, rank-strided slice, modulo the dataset size."""
    import math
    num_samples = math.ceil(n_items * ratio / world)
    total = num_samples * world
    g = torch.Generator()
    g.manual_seed(epoch)
    idx = torch.randperm(total, generator=g)
    return (idx % n_items)[rank:total:world]


def split_global_batch(global_batch: int, world: int) -> int:
    """per-rank batch of a FIXED global batch (BASELINE.json configs[2]: 32 images over 1 / 2 / 4 / 8 GPUs = 32 / 16 / 8 / 4
    each; the reference fixes ``batch_size_per_gpu`` instead, Deraining/basicsr/data/__init__.py:80-95).  The shards must
    be equal: the gradient mean over ranks equals the whole-batch gradient only then."""
    if global_batch <= 0 or world <= 0 or global_batch % world:
        raise ValueError(f"global batch {global_batch} does not split evenly over {world} ranks")
    return global_batch // world


def reduce_loss_dict(losses: dict) -> dict:
    """``reduce_loss_dict`` (base_model.py:353-378): reduce to rank 0 and average there."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: float(v) for k, v in losses.items()}
    keys = sorted(losses)
    t = torch.stack([losses[k].detach().float() for k in keys])
    dist.reduce(t, dst=0)
    if dist.get_rank() == 0:
        t /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, t)}


class FlatGrads:
    """One persistent flat fp32 buffer holding every gradient of ``params`` (the captured training step,
    train_graph.py): ``pack`` copies the freshly produced gradients into it with one multi-tensor copy and
    re-points ``p.grad`` at views of the buffer, so that the data-parallel exchange is a single all-reduce of
    ``flat`` with no per-step loop over the ~1450 tensors and nothing to unpack afterwards (the optimizer reads
    the views).  Built from the first set of gradients it sees; shapes must not change afterwards."""

    def __init__(self, params):
        self.params = list(params)
        grads = [p.grad for p in self.params]
        assert all(g is not None and g.dtype == torch.float32 for g in grads), "FlatGrads needs an fp32 grad per parameter"
        self.flat = torch.empty(sum(g.numel() for g in grads), dtype=torch.float32, device=grads[0].device)
        self.views = [v.view_as(g) for v, g in zip(self.flat.split([g.numel() for g in grads]), grads)]

    def pack(self) -> None:
        with torch.no_grad():
            torch._foreach_copy_(self.views, [p.grad for p in self.params])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def allreduce_mean(self) -> None:
        """mean over ranks, in place (a no-op without an initialised process group / with one rank)"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        gloo_on_gpu = self.flat.is_cuda and dist.get_backend() == "gloo"
        if gloo_on_gpu:
            # gloo stages device tensors through the host.  Fed from a stream with a whole training step still queued it ran at
            # 10-23 s per step on the MI355X box (two ranks sharing the GPU, `bench.py --share-gpu`), fenced at 11 ms: gloo is a
            # flow check here, never the product's collective (RCCL enqueues on the stream), so fence it.
            torch.cuda.synchronize()
        dist.all_reduce(self.flat)
        if gloo_on_gpu:
            torch.cuda.synchronize()
        self.flat.div_(dist.get_world_size())


def allreduce_grads_flat(params) -> None:
    """Average the gradients of ``params`` over all ranks with ONE collective: pack into a flat fp32
    buffer, all-reduce (sum), divide, scatter back in place.  Used between the two hipGraphs of the
    captured training step (train_graph.py); 48.1 MB for MambaSISR6 -> one RCCL call instead of DDP's
    per-bucket calls."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat)
    flat.div_(dist.get_world_size())
    torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
