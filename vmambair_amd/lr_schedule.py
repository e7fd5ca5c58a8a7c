"""Learning-rate schedules of the two reference trainings, as closed-form functions of the training iteration.

The reference drives ``torch.optim.lr_scheduler`` objects from ``update_learning_rate(current_iter)``: the scheduler is
stepped once before every iteration except the first (Deraining/basicsr/models/base_model.py:183-205), so at training
iteration ``it`` (1-based, ``current_iter`` of Deraining/basicsr/train.py:226-238) the optimizer runs with the schedule
evaluated at ``last_epoch = it - 1``.  A hipGraph-replayed step cannot carry a Python scheduler inside the graph; the host
evaluates the schedule and hands the value to ``GraphedTrainStep.set_lr`` / ``FusedAdamEMA.set_lr`` (one 4-byte copy).

  * ``cosine_restart_cyclic``: ``CosineAnnealingRestartCyclicLR`` (Deraining/basicsr/models/lr_scheduler.py:186-232) with the
    Deraining YAML's ``periods [144000, 288000]``, ``restart_weights [1, 1]``, ``eta_mins [3e-4, 1e-6]``
    (Deraining/Deraining/Options/Deraining_mamber33.yml:81-85);
  * ``multistep``: ``torch.optim.lr_scheduler.MultiStepLR`` as configured by SRGAN/options/MambaSISR15_x4.yml:84-87
    (milestones [50000, 70000], gamma 0.5).
Values are pinned by tests/golden/g7_lr.npz (produced by running the reference's scheduler classes).
"""
from __future__ import annotations

import bisect
import math
from typing import Sequence


def _last_epoch(iteration: int) -> int:
    if iteration < 1:
        raise ValueError("training iterations are counted from 1 (current_iter of the reference's loop)")
    return iteration - 1


def cosine_restart_cyclic(iteration: int, base_lr: float, periods: Sequence[int], restart_weights: Sequence[float] = (1,),
                          eta_mins: Sequence[float] = (0.0,)) -> float:
    """learning rate the optimizer uses at training iteration ``iteration`` (>= 1)"""
    assert len(periods) == len(restart_weights) == len(eta_mins), "periods, restart_weights and eta_mins go together"
    e = _last_epoch(iteration)
    ends = []
    total = 0
    for p in periods:
        total += p
        ends.append(total)
    idx = bisect.bisect_left(ends, e)        # first cycle whose end is >= e (lr_scheduler.py:115-133)
    if idx >= len(periods):
        raise ValueError(f"iteration {iteration} is past the last period ({ends[-1]})")
    start = 0 if idx == 0 else ends[idx - 1]
    lo = eta_mins[idx]
    return lo + restart_weights[idx] * 0.5 * (base_lr - lo) * (1.0 + math.cos(math.pi * ((e - start) / periods[idx])))


def multistep(iteration: int, base_lr: float, milestones: Sequence[int], gamma: float = 0.1) -> float:
    """MultiStepLR: the rate is multiplied by ``gamma`` once per milestone reached (a repeated milestone counts twice)"""
    e = _last_epoch(iteration)
    return base_lr * gamma ** sum(1 for m in milestones if m <= e)
