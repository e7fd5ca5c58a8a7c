"""Checkpoint files of the reference's training loops, read and written by the drop-in modules.

The reference stores a network as ``torch.save({param_key: state_dict, ...})`` with ``param_key`` in
{``'params'``, ``'params_ema'``} and every ``module.`` prefix (DataParallel / DDP wrappers) removed
(``BaseModel.save_network``, Deraining/basicsr/models/base_model.py:213-244; the SRGAN / RealSR trees use
pip-basicsr's identical method, and ``RealESRGANer`` reads the same files preferring ``params_ema``,
RealSR/VmambaIR/utils.py:57-63).  ``load_network`` (:281-309) picks ``param_key`` (falling back to
``'params'`` when the file has no ``'params_ema'``), strips ``module.`` again and calls
``load_state_dict(strict=...)``; with ``strict=False`` entries whose SIZE differs are set aside instead of
raising (``_print_different_keys_loading``, :246-279).

Because the modules under ``vmambair_amd`` keep the reference's parameter names and shapes, a released
``net_g_*.pth`` loads with ``strict=True`` and a file written here loads in the reference (checked against
the reference's own reader by tests/golden/make_golden.py, fixture ``g5_ckpt_mambasisr6_d8.pth``).
"""
from __future__ import annotations

import logging
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Union

import torch

log = logging.getLogger("vmambair_amd.checkpoint")


def bare_model(net: torch.nn.Module) -> torch.nn.Module:
    """the module under DataParallel / DistributedDataParallel (base_model.py:135-142)"""
    if isinstance(net, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        net = net.module
    return net


def _strip_module(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def network_state(net: torch.nn.Module) -> "OrderedDict[str, torch.Tensor]":
    """CPU state dict without ``module.`` prefixes: what ``save_network`` stores per key (:231-236)"""
    return OrderedDict((k, v.detach().cpu()) for k, v in _strip_module(bare_model(net).state_dict()).items())


def save_network(net: Union[torch.nn.Module, Sequence[torch.nn.Module]], save_path: str,
                 param_key: Union[str, Sequence[str]] = "params") -> str:
    """``BaseModel.save_network``: one file, one state dict per ``param_key`` (``[net_g, net_g_ema]`` with
    ``['params', 'params_ema']`` is what the training loops write).  Written atomically."""
    nets = list(net) if isinstance(net, (list, tuple)) else [net]
    keys = list(param_key) if isinstance(param_key, (list, tuple)) else [param_key]
    if len(nets) != len(keys):
        raise ValueError("The lengths of net and param_key should be the same.")
    blob = {k: network_state(n) for n, k in zip(nets, keys)}
    os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
    tmp = save_path + ".tmp"
    torch.save(blob, tmp)
    os.replace(tmp, save_path)
    return save_path


def save_iteration(net, models_dir: str, net_label: str, current_iter: int, param_key="params") -> str:
    """file naming of the reference: ``{label}_{iter}.pth``, ``latest`` for iteration -1 (:225-228)"""
    it = "latest" if current_iter == -1 else current_iter
    return save_network(net, os.path.join(models_dir, f"{net_label}_{it}.pth"), param_key)


def read_state(load_path: str, param_key: Optional[str] = "params") -> "OrderedDict[str, torch.Tensor]":
    """the state dict ``load_network`` would hand to ``load_state_dict`` (key choice + ``module.`` stripping)"""
    blob = torch.load(load_path, map_location="cpu", weights_only=True)
    if param_key is not None:
        if param_key not in blob and "params" in blob:
            log.info("Loading: %s does not exist, use params.", param_key)
            param_key = "params"
        blob = blob[param_key]
    return _strip_module(blob)


def different_keys(net: torch.nn.Module, state: Dict[str, torch.Tensor], strict: bool = True) -> Dict[str, List[str]]:
    """``_print_different_keys_loading``: names only in the net / only in the file; with ``strict=False`` the
    same-name-different-size entries are renamed ``<key>.ignore`` in ``state`` so that they are not loaded."""
    cur = bare_model(net).state_dict()
    res = {"missing_in_file": sorted(set(cur) - set(state)), "unexpected_in_file": sorted(set(state) - set(cur)),
           "size_mismatch": []}
    for v in res["missing_in_file"]:
        log.warning("Current net - loaded net: %s", v)
    for v in res["unexpected_in_file"]:
        log.warning("Loaded net - current net: %s", v)
    if not strict:
        for k in sorted(set(cur) & set(state)):
            if cur[k].size() != state[k].size():
                log.warning("Size different, ignore [%s]: crt_net: %s; load_net: %s", k, tuple(cur[k].shape), tuple(state[k].shape))
                state[k + ".ignore"] = state.pop(k)
                res["size_mismatch"].append(k)
    return res


def load_network(net: torch.nn.Module, load_path: str, strict: bool = True, param_key: Optional[str] = "params"):
    """``BaseModel.load_network``.  Raises ``RuntimeError`` on any name / size difference when ``strict``
    (``load_state_dict``'s own error), returns its ``_IncompatibleKeys`` otherwise."""
    state = read_state(load_path, param_key)
    different_keys(net, state, strict)
    return bare_model(net).load_state_dict(state, strict=strict)


def load_for_inference(net: torch.nn.Module, load_path: str) -> str:
    """``RealESRGANer.__init__``: prefer ``params_ema``, strict (RealSR/VmambaIR/utils.py:57-63) -> key used"""
    blob = torch.load(load_path, map_location="cpu", weights_only=True)
    key = "params_ema" if "params_ema" in blob else "params"
    bare_model(net).load_state_dict(_strip_module(blob[key]), strict=True)
    net.eval()
    return key


# ---- training state (resume) ---------------------------------------------------------------------
def save_training_state(step, states_dir: str, epoch: int, current_iter: int) -> Optional[str]:
    """``BaseModel.save_training_state`` (Deraining/basicsr/models/base_model.py:312-334): ``{current_iter}.state`` holding
    ``{'epoch', 'iter', 'optimizers': [optimizer.state_dict()], 'schedulers': [...]}``; nothing is written for
    ``current_iter == -1`` (the "latest" save at the end of training).  ``step``: a ``GraphedTrainStep`` (or anything with
    ``state_dict()`` returning ``{'iter', 'optimizers', 'ema'}``).  The optimizer entry has ``torch.optim.Adam(W)``'s layout
    (vmambair_amd/optim.py), so the reference's ``resume_training`` can load it into its torch optimizer and this module can
    load a ``.state`` file the reference wrote.  ``schedulers`` holds the one thing a closed-form schedule needs -- the
    iteration -- in ``_LRScheduler.state_dict()``'s field name: ``last_epoch`` = scheduler steps taken so far = ``iter - 1`` (the
    reference steps its scheduler before every iteration but the first, base_model.py:183-193)."""
    if current_iter == -1:
        return None
    sd = step.state_dict()
    state = {"epoch": int(epoch), "iter": int(current_iter), "optimizers": sd["optimizers"],
             "schedulers": [{"last_epoch": max(int(current_iter) - 1, 0)}], "ema": sd.get("ema")}
    os.makedirs(states_dir, exist_ok=True)
    path = os.path.join(states_dir, f"{current_iter}.state")
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)
    return path


def resume_training(step, resume_state: Union[str, dict], ema_from: Optional[str] = None) -> dict:
    """``BaseModel.resume_training`` (base_model.py:336-351) + the iteration bookkeeping of ``train.py`` (:176-190): puts the
    optimizer moments, step count and learning rate back -> ``{'epoch', 'iter'}`` to continue from.  ``resume_state``: a path
    or the loaded dict.

    EMA weights: a ``.state`` file written by THIS package carries them (``'ema'``); one written by the reference does not -- the
    reference keeps its EMA net in the network checkpoint (``params_ema`` of ``net_g_<iter>.pth``, base_model.py:234-244) and
    restores it through ``load_network``.  For a step that tracks EMA weights such a file therefore needs ``ema_from`` = that
    ``.pth`` (its ``params_ema`` is loaded into the step's EMA tensors by parameter name); without it this raises instead of
    silently continuing from freshly initialised EMA weights."""
    if isinstance(resume_state, str):
        resume_state = torch.load(resume_state, map_location="cpu", weights_only=False)
    opts = resume_state["optimizers"]
    assert len(opts) == 1, "Wrong lengths of optimizers"   # the reference's own assertion (base_model.py:344-345)
    ema = resume_state.get("ema")
    tracks_ema = getattr(step, "ema", None) is not None
    if ema is None and tracks_ema:
        if ema_from is None:
            raise ValueError("the training state has no 'ema' entry (a .state file written by the reference: its EMA weights live in "
                             "net_g_<iter>.pth as 'params_ema'); pass ema_from=<that .pth>, or build the step with ema_decay=0")
        raw = torch.load(ema_from, map_location="cpu", weights_only=True)
        if "params_ema" not in raw:   # read_state would fall back to 'params' (load_network's rule): not for the EMA weights
            raise KeyError(f"{ema_from} has no 'params_ema' entry: the EMA weights would be seeded from the raw weights")
        blob = _strip_module(raw["params_ema"])
        names = [n for n, p in bare_model(step.net).named_parameters() if p.requires_grad]
        missing = [n for n in names if n not in blob]
        if missing:
            raise KeyError(f"{ema_from}: params_ema lacks {len(missing)} parameters, e.g. {missing[:3]}")
        ema = [blob[n].detach().clone() for n in names]
    step.load_state_dict({"iter": resume_state["iter"], "optimizers": opts, "ema": ema})
    return {"epoch": int(resume_state["epoch"]), "iter": int(resume_state["iter"])}


def train_loop(step, batches, schedule, start_iter: int = 0, total_iters: Optional[int] = None, save_every: int = 0,
               states_dir: Optional[str] = None, epoch: int = 0, on_iter=None) -> int:
    """The reference's inner training loop (Deraining/basicsr/train.py:226-271, SRGAN/VmambaIR/train.py) around a
    ``GraphedTrainStep``: for every batch, ``current_iter += 1``; the schedule is evaluated at that iteration and handed to
    ``step.set_lr`` BEFORE the step (``update_learning_rate(current_iter)``, base_model.py:183-205) -- this is the caller of
    ``lr_schedule.py`` and ``set_lr`` -- then the optimisation step; every ``save_every`` iterations the training state is
    written.  ``batches``: an iterable of ``(lq, gt)``; ``schedule``: ``iteration -> learning rate`` (e.g.
    ``lambda it: lr_schedule.multistep(it, 2e-4, [50000, 70000], 0.5)``).  -> the last iteration run."""
    it = int(start_iter)
    for lq, gt in batches:
        if total_iters is not None and it >= total_iters:
            break
        it += 1
        step.set_lr(float(schedule(it)))
        loss = step(lq, gt)
        if on_iter is not None:
            on_iter(it, loss)
        if save_every and states_dir and it % save_every == 0:
            save_training_state(step, states_dir, epoch, it)
    return it
