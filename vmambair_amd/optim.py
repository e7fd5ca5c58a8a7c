"""Adam + EMA of the training step as one HIP launch (``oss_adam_ema_step``, vmambair_amd/csrc/oss_optim.hip).

Mirrors what the reference's ``optimize_parameters`` does after ``backward`` (SRGAN/VmambaIR/models/
MambaSISR_model.py:138-147): ``optimizer_g.step()`` with ``torch.optim.Adam(lr, betas)`` (no weight decay, no
amsgrad -- options/MambaSISR15_x4.yml:71-75) followed by ``model_ema(decay)``.  State layout and arithmetic equal
``torch.optim.Adam``'s (``exp_avg``, ``exp_avg_sq``, step count), so a run can switch between the two.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _capi


class FusedAdamEMA:
    def __init__(self, params: Sequence[torch.Tensor], ema: Optional[Sequence[torch.Tensor]] = None, lr: float = 2e-4,
                 betas=(0.9, 0.99), eps: float = 1e-8, ema_decay: float = 0.999, weight_decay: float = 0.0,
                 clip_grad_norm: Optional[float] = None):
        """``weight_decay`` > 0: torch.optim.AdamW (decoupled); ``clip_grad_norm``: ``clip_grad_norm_(params, value)``
        in front of the update -- the Deraining step (image_restoration_model.py:121-167; Options/Deraining_mamber33.yml:
        AdamW 3e-4, decay 1e-4, betas (0.9, 0.999), use_grad_clip)."""
        self.params: List[torch.Tensor] = list(params)
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params), \
            "FusedAdamEMA: contiguous fp32 GPU parameters only"
        self.ema = list(ema) if ema is not None else None
        assert self.ema is None or len(self.ema) == len(self.params)
        self.lr, self.betas, self.eps, self.ema_decay = lr, betas, eps, ema_decay
        self.weight_decay, self.clip = float(weight_decay), clip_grad_norm
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        dev = self.params[0].device
        # step, 1 - beta1^t, 1 - beta2^t, learning rate: the kernel reads the rate from here (a captured launch must follow
        # the reference's schedulers: MultiStepLR / CosineAnnealingRestartCyclicLR stepped per iteration,
        # Deraining/basicsr/models/base_model.py:183-193) -- ``set_lr`` rewrites it between replays
        self.state = torch.tensor([0.0, 0.0, 0.0, float(lr)], dtype=torch.float32, device=dev)
        self._lr_host = torch.tensor([float(lr)], dtype=torch.float32).pin_memory()
        self._lr_copied: Optional[torch.cuda.Event] = None
        self._sig = None            # grad pointers the device table was built for
        self._n = sum((p.numel() + _capi.ADAM_CHUNK - 1) // _capi.ADAM_CHUNK for p in self.params)
        nbytes = self._n * C.sizeof(_capi.AdamChunk)
        # allocated up front (no allocation may happen inside a stream capture); the pinned copy stays alive because a
        # captured host-to-device memcpy node re-reads it on every replay
        self._host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self._table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._copied: Optional[torch.cuda.Event] = None   # the last eager host-to-device copy of the table
        self.grad_scale = torch.ones(1, dtype=torch.float32, device=dev)   # clip coefficient of the current step
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)  # gradient norm before clipping (logging)

    def _build(self):
        rows = []
        for i, p in enumerate(self.params):
            g = p.grad
            assert g is not None and g.dtype == torch.float32 and g.is_contiguous(), "FusedAdamEMA: every parameter needs an fp32 grad"
            e = self.ema[i].data_ptr() if self.ema is not None else 0
            for off in range(0, p.numel(), _capi.ADAM_CHUNK):
                n = min(_capi.ADAM_CHUNK, p.numel() - off)
                b = off * 4
                rows.append((p.data_ptr() + b, g.data_ptr() + b, self.exp_avg[i].data_ptr() + b,
                             self.exp_avg_sq[i].data_ptr() + b, e + b if e else 0, n, 0))
        assert len(rows) == self._n
        arr = (_capi.AdamChunk * len(rows))(*[_capi.AdamChunk(*r) for r in rows])
        capturing = torch.cuda.is_current_stream_capturing()
        if self._copied is not None and not capturing:
            # eager steps rebuild the table whenever autograd hands out new gradient tensors: the previous asynchronous
            # copy out of the pinned buffer must have executed before the buffer is overwritten (a GPU running a step
            # behind the host would otherwise read the NEXT step's pointers)
            self._copied.synchronize()
        C.memmove(self._host.data_ptr(), C.addressof(arr), C.sizeof(arr))
        self._table.copy_(self._host, non_blocking=True)   # inside a stream capture this becomes a memcpy node of the graph
        if not capturing:
            self._copied = torch.cuda.Event()
            self._copied.record()

    # ---- learning rate / state (torch.optim.Adam(W)-compatible) --------------------------------
    @torch.no_grad()
    def set_lr(self, lr: float) -> None:
        """new learning rate for the following steps -- eager or replayed from a hipGraph: a 4-byte host-to-device copy
        into ``state[3]`` on the current stream, no capture involved (``param_group['lr'] = lr`` of the reference's
        ``_set_lr`` / scheduler.step(), base_model.py:164-193)."""
        lr = float(lr)
        if lr == self.lr:
            return
        assert not torch.cuda.is_current_stream_capturing(), "set_lr belongs between graph replays, not inside a capture"
        if self._lr_copied is not None:
            self._lr_copied.synchronize()      # the previous copy out of the pinned scalar has executed
        self._lr_host[0] = lr
        self.state[3:4].copy_(self._lr_host, non_blocking=True)
        self._lr_copied = torch.cuda.Event()
        self._lr_copied.record()
        self.lr = lr

    def state_dict(self) -> dict:
        """``torch.optim.Adam(W).state_dict()`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, one param
        group), so that the ``optimizers`` entry of a reference ``.state`` file (``save_training_state``,
        Deraining/basicsr/models/base_model.py:312-334) round-trips between this class and torch's optimizer."""
        step = self.state[0].detach().clone().cpu()
        st = {i: {"step": step.clone(), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
              for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))} if float(step) > 0 else {}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(self.params)))}
        if "initial_lr" in self.__dict__:
            group["initial_lr"] = self.initial_lr
        return {"state": st, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> None:
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        assert len(ids) == len(self.params), "optimizer state has a different number of parameters"
        g0 = groups[0]
        assert all(abs(g["lr"] - g0["lr"]) == 0 and tuple(g["betas"]) == tuple(g0["betas"]) for g in groups), \
            "FusedAdamEMA keeps ONE set of hyper-parameters (the reference uses one param group, base_model.py:121-135)"
        self.betas, self.eps = tuple(g0["betas"]), float(g0["eps"])
        self.weight_decay = float(g0.get("weight_decay", 0.0))
        if "initial_lr" in g0:
            self.initial_lr = float(g0["initial_lr"])
        steps = set()
        for k, pid in enumerate(ids):
            ent = sd["state"].get(pid)
            if ent is None:
                self.exp_avg[k].zero_(); self.exp_avg_sq[k].zero_()
                steps.add(0.0)
                continue
            self.exp_avg[k].copy_(ent["exp_avg"])
            self.exp_avg_sq[k].copy_(ent["exp_avg_sq"])
            steps.add(float(ent["step"]))
        assert len(steps) == 1, "FusedAdamEMA keeps one step count for all parameters"
        t = steps.pop()
        b1, b2 = self.betas
        self.state.copy_(torch.tensor([t, 1.0 - b1 ** t, 1.0 - b2 ** t, float(g0["lr"])], dtype=torch.float32))
        self.lr = float(g0["lr"])

    @torch.no_grad()
    def step(self) -> None:
        sig = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self.params)
        if sig != self._sig:   # eager mode: new grad tensors every step; under a hipGraph the addresses are fixed
            self._build()
            self._sig = sig
        lib = _capi.load()
        scale = None
        if self.clip is not None:
            # clip_grad_norm_ (torch/nn/utils/clip_grad.py): total 2-norm of all gradients, coefficient clamped at 1 -- on
            # the device, no read-back; the multiplication itself happens inside the optimizer launch
            norms = torch._foreach_norm([p.grad for p in self.params])
            total = torch.linalg.vector_norm(torch.stack(norms))
            self.total_norm.copy_(total.reshape(1))
            self.grad_scale.copy_((self.clip / (total + 1e-6)).clamp(max=1.0).reshape(1))
            scale = self.grad_scale.data_ptr()
        with torch.cuda.device(self.params[0].device):
            # lr = -1: the kernel reads state[3] (set_lr), so a captured launch follows the schedule
            _capi.check(lib.oss_adamw_ema_step(self._table.data_ptr(), self._n, self.state.data_ptr(), -1.0, self.betas[0],
                                               self.betas[1], self.eps, self.weight_decay, self.ema_decay, scale,
                                               torch.cuda.current_stream().cuda_stream), "oss_adamw_ema_step")
