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
    the views).  Built from the first set of gradients it sees; shapes must not change afterwards.

    (round 6) ``order`` / ``n_buckets``: the buffer is laid out in ``order`` (indices into ``params``: the order in which the
    backward finishes the gradients, last layers first) and cut into ``n_buckets`` contiguous slices of about equal size --
    DDP's reverse-order buckets (Deraining/basicsr/models/base_model.py:76-82 wraps the net in DistributedDataParallel, whose
    reducer all-reduces a bucket as soon as its gradients exist).  ``pack_bucket(k)`` / ``allreduce_bucket(k)`` work on one slice,
    so bucket k's exchange can run on a side stream while the step finishes the gradients of bucket k + 1.
    ``via_host``: stage every exchange through pinned host memory and the CPU backend (gloo) -- the fall-back of bench.py when
    the RCCL preflight fails, and what ``--share-gpu`` uses; a flow that always works, never the fast path."""

    def __init__(self, params, order=None, n_buckets: int = 1, via_host: bool = False):
        self.params = list(params)
        grads = [p.grad for p in self.params]
        assert all(g is not None and g.dtype == torch.float32 for g in grads), "FlatGrads needs an fp32 grad per parameter"
        n = len(grads)
        order = list(range(n)) if order is None else list(order)
        assert sorted(order) == list(range(n)), "order must be a permutation of the parameter indices"
        self.order = order
        self.flat = torch.empty(sum(g.numel() for g in grads), dtype=torch.float32, device=grads[0].device)
        pieces = self.flat.split([grads[i].numel() for i in order])
        self.views = [None] * n
        for piece, i in zip(pieces, order):
            self.views[i] = piece.view_as(grads[i])
        # bucket boundaries: cut after the parameter at which the running size passes the bucket's share; the share is re-computed from
        # what is LEFT after every cut (one large tensor -- the x4 tail's 3x3 convolutions sit at the front of the order -- must not
        # leave the following bucket with a sliver)
        n_buckets = max(1, min(int(n_buckets), n))
        total, run, k = self.flat.numel(), 0, 1
        self.bucket_members, cur, self.bucket_ranges, start = [], [], [], 0
        goal = total / n_buckets
        for i in order:
            cur.append(i)
            run += grads[i].numel()
            if k < n_buckets and run >= goal:
                self.bucket_members.append(cur)
                self.bucket_ranges.append((start, run))
                cur, start, k = [], run, k + 1
                goal = run + (total - run) / (n_buckets - k + 1)
        if cur or not self.bucket_members:
            self.bucket_members.append(cur)
            self.bucket_ranges.append((start, total))
        self.n_buckets = len(self.bucket_members)
        self.via_host = bool(via_host)
        self._host = None

    def pack(self) -> None:
        with torch.no_grad():
            torch._foreach_copy_(self.views, [p.grad for p in self.params])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def pack_bucket(self, k: int) -> None:
        idx = self.bucket_members[k]
        with torch.no_grad():
            torch._foreach_copy_([self.views[i] for i in idx], [self.params[i].grad for i in idx])
        for i in idx:
            self.params[i].grad = self.views[i]

    def _exchange(self, t: torch.Tensor) -> None:
        world = dist.get_world_size()
        staged = t.is_cuda and (self.via_host or dist.get_backend() == "gloo")
        if staged:
            # host staging (gloo has no device path worth using: fed from a stream with a whole training step still queued it ran at
            # 10-23 s per step on the MI355X box; fenced and staged through pinned memory it is 11 ms).  A flow check / fall-back.
            if self._host is None or self._host.numel() < self.flat.numel():
                self._host = torch.empty(self.flat.numel(), dtype=torch.float32).pin_memory()
            h = self._host[:t.numel()]
            torch.cuda.current_stream().synchronize()
            h.copy_(t)
            dist.all_reduce(h)
            h.div_(world)
            t.copy_(h, non_blocking=False)
            return
        dist.all_reduce(t)
        t.div_(world)

    def allreduce_mean(self) -> None:
        """mean over ranks, in place (a no-op without an initialised process group / with one rank)"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        self._exchange(self.flat)

    def allreduce_bucket(self, k: int) -> None:
        """mean over ranks of bucket ``k`` only (on the CURRENT stream: the caller picks the side stream)"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        a, b = self.bucket_ranges[k]
        self._exchange(self.flat[a:b])


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
