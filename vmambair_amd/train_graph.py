"""hipGraph-captured training step for the OSS nets.

At 64×64 LQ the reference's step is launch-latency bound: one MamberBlock forward is ~140 aten ops
 and the net has 50 of them, so an eager step issues >10^4 tiny kernels.  All shapes
are static during training (fixed patch size and per-GPU batch,
SRGAN/options/MambaSISR15_x4.yml:26-33), so the whole step -- forward, L1 loss, backward, Adam,
EMA: exactly ``optimize_parameters`` of the reference (SRGAN/VmambaIR/models/MambaSISR_model.py:120-147)
-- is captured once into a hipGraph and replayed.

Gradients are never accumulated: ``p.grad`` is cleared (a host-side pointer reset, no kernel) before
the captured backward, so autograd hands every parameter its gradient tensor as-is instead of
launching one add per parameter tensor (the net has ~1450 of them).

Multi-GPU: the graph holds forward + backward only; after the replay the gradients are packed into
ONE flat buffer, all-reduced with a single RCCL call (48.1 MB for MambaSISR6 -- xGMI-link-bound
~0.1-0.6 ms) and scattered back, then a second graph applies Adam + EMA.  No DDP hooks
inside a capture, no per-bucket calls.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


class GraphedTrainStep:
    def __init__(self, net: torch.nn.Module, lr: float = 2e-4, betas=(0.9, 0.99), ema_decay: float = 0.999,
                 autocast_dtype: Optional[torch.dtype] = torch.bfloat16,
                 loss_fn: Callable = torch.nn.functional.l1_loss, warmup: int = 3):
        self.net = net
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.device = self.params[0].device
        assert self.device.type == "cuda", "graph capture needs a GPU"
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.autocast_dtype = autocast_dtype
        self.loss_fn = loss_fn
        self.ema_decay = ema_decay
        self.warmup = warmup
        self.ema = [p.detach().clone() for p in self.params]
        self.opt = torch.optim.Adam(self.params, lr=lr, betas=betas, fused=True, capturable=True)
        self.graph_fb: Optional[torch.cuda.CUDAGraph] = None
        self.graph_opt: Optional[torch.cuda.CUDAGraph] = None
        self.static_lq = self.static_gt = self.static_loss = None

    # ---- pieces ------------------------------------------------------------------------------
    def _fwd_bwd(self):
        for p in self.params:  # host-side only: the backward then assigns instead of accumulating
            p.grad = None
        with torch.autocast("cuda", dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
            out = self.net(self.static_lq)
        loss = self.loss_fn(out.float(), self.static_gt)
        loss.backward()
        return loss.detach()

    def _opt_ema(self):
        self.opt.step()
        with torch.no_grad():  # model_ema(decay) of the reference
            torch._foreach_mul_(self.ema, self.ema_decay)
            torch._foreach_add_(self.ema, [p.detach() for p in self.params], alpha=1.0 - self.ema_decay)

    def _allreduce(self):
        if self.world > 1:
            from .ddp import allreduce_grads_flat
            allreduce_grads_flat(self.params)

    # ---- capture -----------------------------------------------------------------------------
    def capture(self, lq: torch.Tensor, gt: torch.Tensor) -> None:
        self.static_lq = lq.clone()
        self.static_gt = gt.clone()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture: lazy inits, MIOpen find, LDS attributes
            for _ in range(self.warmup):
                self._fwd_bwd()
                self._allreduce()
                self._opt_ema()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self.static_loss = self._fwd_bwd()
            if self.world == 1:
                self._opt_ema()
        if self.world > 1:
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt):
                self._opt_ema()

    # ---- replay ------------------------------------------------------------------------------
    def __call__(self, lq: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        if self.graph_fb is None:
            self.capture(lq, gt)
        self.static_lq.copy_(lq, non_blocking=True)
        self.static_gt.copy_(gt, non_blocking=True)
        self.graph_fb.replay()
        if self.world > 1:
            self._allreduce()
            self.graph_opt.replay()
        return self.static_loss
