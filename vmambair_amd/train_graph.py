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

16-bit shadow weights: under autocast every conv / projection weight would be cast fp32 -> bf16 by its
own kernel in the forward and its gradient cast back by another one in the backward (~40 launches per
block).  Instead all such parameters get a 16-bit shadow leaf that is refreshed by ONE multi-tensor
copy per step; the net runs on the shadows (``torch.func.functional_call``) and their gradients are
converted to the fp32 master gradients by one more multi-tensor copy.  Numerically identical to
autocast (the same rounding of the same master weights every step).

Multi-GPU: the first graph holds forward + backward and ends by packing every gradient into ONE
persistent flat buffer (``ddp.FlatGrads``; ``p.grad`` then points at views of it); between the graphs
that buffer is all-reduced with a single RCCL call (48.1 MB for MambaSISR6 -- xGMI-link-bound
~0.1-0.6 ms) and divided in place; a second graph applies Adam + EMA reading the views.
No DDP hooks inside a capture, no per-bucket calls, no per-step host loop over the gradient tensors.

(round 6) ``grad_buckets = K > 1`` overlaps that exchange with the end of the backward, the way the reference's
DistributedDataParallel reducer does with its reverse-order buckets (Deraining/basicsr/models/base_model.py:76-82): the weight
gradients of the step are only RECORDED during the backward and run as grouped launches at its end (ops/_common.py), so the
overlap window is that end -- the first graph stops after the backward, K small graphs each flush the recorded products and
finishing sums of ONE bucket (backward order: last layers first; the cut points are the library's record counts at the moment
the bucket's last gradient was handed to autograd, found on the first warm-up pass) and pack it into its slice of the flat
buffer, and between their replays bucket k's all-reduce is issued on a side stream while the main stream flushes bucket k + 1.
The collectives stay OUTSIDE the graphs.  Same kernels on the same operands as K = 1, so the gradients are bit-identical.

Deferred finishing (``defer_finishes``), the adoption contract that goes with it and the optional
micro-batch branches (``micro_streams``) are described at their code below and in ops.py.
"""
from __future__ import annotations

import os

from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import ops as _ops


class GraphedTrainStep:
    def __init__(self, net: torch.nn.Module, lr: float = 2e-4, betas=(0.9, 0.99), ema_decay: float = 0.999,
                 autocast_dtype: Optional[torch.dtype] = torch.bfloat16,
                 loss_fn: Callable = torch.nn.functional.l1_loss, warmup: int = 3, shadow_weights: bool = True,
                 fused_optimizer: bool = True, split_graphs: bool = False, overlap_wgrads: bool = False,
                 defer_finishes: bool = True, micro_streams: int = 1, weight_decay: float = 0.0,
                 clip_grad_norm: Optional[float] = None, multi_shape: bool = False, grad_buckets: int = 1,
                 allreduce_via_host: bool = False):
        """``weight_decay`` / ``clip_grad_norm`` / ``ema_decay=0``: the Deraining step (AdamW + clip_grad_norm_(0.01), no
        EMA: Deraining/basicsr/models/image_restoration_model.py:121-167).  ``multi_shape``: keep one forward+backward
        graph per (lq, gt) shape -- the progressive patch schedule of that tree changes the patch size and the batch
        six times (Deraining/basicsr/train.py:213-271) -- all sharing ONE optimizer graph (implies ``split_graphs``: the
        gradients then always sit in the same flat buffer, so the optimizer's pointer table never changes)."""
        self.net = net
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.device = self.params[0].device
        assert self.device.type == "cuda", "graph capture needs a GPU"
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # two graphs (forward+backward | optimizer) with the gradient all-reduce between them: always for world > 1;
        # ``split_graphs`` forces the same structure on one GPU (tests)
        self.multi_shape = multi_shape
        # grad_buckets > 1: the gradient exchange in buckets overlapped with the flushes at the end of the backward (module docstring)
        self.nbuckets = max(1, int(grad_buckets))
        self.allreduce_via_host = bool(allreduce_via_host)
        self.split = self.world > 1 or split_graphs or multi_shape or self.nbuckets > 1
        # split mode: the first graph ends by packing every gradient into ONE persistent fp32 buffer (a multi-tensor copy,
        # captured) and re-points ``p.grad`` at views of it, so that the only work between the two graphs is a single
        # all-reduce of that buffer -- no per-step host loop over the ~1450 gradient tensors, no unpack
        self._flat = None   # ddp.FlatGrads
        self.autocast_dtype = autocast_dtype
        self.loss_fn = loss_fn
        self.ema_decay = ema_decay
        self.warmup = warmup
        # 16-bit shadows of the parameters autocast would cast per use (conv weights / biases and the
        # projection matrices of SS2D_1); norms, A_logs, D, dt biases and the fp32 channel branch stay fp32
        self.shadow = {}
        if shadow_weights and autocast_dtype is not None:
            mods = dict(net.named_modules())
            for name, p in net.named_parameters():
                owner, leaf = name.rsplit(".", 1) if "." in name else ("", name)
                m = mods.get(owner)
                if ".conv_cin." in name or ".conv_cout." in name:
                    continue  # consumed in fp32 by the channel branch
                dense_conv = isinstance(m, torch.nn.Conv2d) and m.groups == 1  # depth-wise convs run on our fp32-weight kernel
                if dense_conv and m.kernel_size == (1, 1) and _ops.pointwise.CONV1X1_IMPL == "mfma" and owner.rsplit(".", 1)[-1] in (
                        "in_conv", "out_conv", "project_in", "project_out", "reduce_chan_level2", "reduce_chan_level3"):
                    continue  # the MFMA 1x1 kernels read the fp32 masters and narrow them in their loader
                if dense_conv and m.kernel_size == (3, 3) and min(m.in_channels, m.out_channels) <= 4 and _ops.conv3x3.THIN_IMPL:
                    continue  # patch_embed / the tail's last layer: the thin-convolution kernels read the fp32 masters (ops/conv3x3.py)
                # x_proj_weight / dt_projs_weight: the fused spatial core (SS2DCoreFn) reads the fp32 masters
                fused_proj = getattr(m, "fused_core", False) and getattr(m, "omni", False)
                if dense_conv or (leaf in ("x_proj_weight", "dt_projs_weight") and not fused_proj):
                    self.shadow[name] = (p, torch.empty_like(p, dtype=autocast_dtype).requires_grad_())
        self._masters = [m for m, _ in self.shadow.values()]
        self._shadows = [s for _, s in self.shadow.values()]
        self._master_grads = [torch.zeros_like(m) for m in self._masters]
        self._shadow_map = {n: s for n, (_, s) in self.shadow.items()}
        self.ema = [p.detach().clone() for p in self.params] if ema_decay > 0 else None
        # Adam + EMA: one HIP launch over a chunk table (vmambair_amd/optim.py) or torch's fused multi-tensor Adam + foreach EMA
        self.fopt = None
        self.opt = None
        if fused_optimizer and all(p.dtype == torch.float32 and p.is_contiguous() for p in self.params):
            from .optim import FusedAdamEMA
            self.fopt = FusedAdamEMA(self.params, self.ema, lr=lr, betas=betas, ema_decay=ema_decay,
                                     weight_decay=weight_decay, clip_grad_norm=clip_grad_norm)
        else:
            assert clip_grad_norm is None, "gradient clipping is implemented in the fused optimizer launch only"
            cls = torch.optim.AdamW if weight_decay > 0 else torch.optim.Adam
            kw = dict(weight_decay=weight_decay) if weight_decay > 0 else {}
            # the rate as a device tensor: a captured torch step then follows ``set_lr`` too
            self.opt = cls(self.params, lr=torch.tensor(float(lr), dtype=torch.float32, device=self.device), betas=betas,
                           fused=True, capturable=True, **kw)
        self.lr = float(lr)
        self.iteration = 0   # optimizer steps taken through __call__ (``current_iter`` of the reference's training loop)
        # opt-in: weight gradients (needed only by the optimizer) on a second stream next to the input-gradient chain.
        # Measured SLOWER inside the hipGraph on this stack (138.5 vs 141.5 images/s: the cross-stream edges cost more
        # than the overlap of these 5-20 us kernels buys), hence off by default
        self.wside = torch.cuda.Stream(device=self.device) if overlap_wgrads else None
        # the ~12 small partial-sum finishing launches per block (weight-gradient slabs, LayerNorm / depth-wise conv /
        # channel partials) run as ONE launch at the end of the backward (oss_flush_finishes)
        self.defer = defer_finishes
        # micro_streams = M > 1: the batch is cut into M micro-batches whose forward + backward run as M parallel branches
        # of the graph (one HIP stream each, ONE fork and ONE join per step); their gradients are added before the
        # optimizer, so the step equals the full-batch step (mean loss; no cross-image statistics anywhere in the nets).
        # At batch 8 most kernels of the step are too small to fill 256 CUs: two half-size launches side by side do.
        # Every branch runs the net on its own leaf tensors (aliases of the parameters / shadows) so that each backward
        # ASSIGNS its gradients -- no per-tensor accumulation kernels.
        self.nmicro = max(1, int(micro_streams))
        assert not (self.nmicro > 1 and overlap_wgrads), "micro_streams and overlap_wgrads are alternatives"
        self.mstreams = [torch.cuda.Stream(device=self.device) for _ in range(self.nmicro)] if self.nmicro > 1 else []
        self._names = [n for n, p in net.named_parameters() if p.requires_grad]
        self._leaf_sets = []
        if self.nmicro > 1:
            named = dict(net.named_parameters())
            for _ in range(self.nmicro):
                self._leaf_sets.append({n: self._shadow_map.get(n, named[n]).detach().requires_grad_() for n in self._names})
            self._shadow_tmp = [torch.zeros_like(m) for m in self._masters]
        self.ftables = [None] * self.nmicro
        self.wtables = [None] * self.nmicro
        self.wgrad_stats = None   # of the last deferred backward: bytes held for recorded products, grouped launches (ops/_common.py)
        if self.nbuckets > 1:
            assert self.nmicro == 1 and not overlap_wgrads and defer_finishes, "grad_buckets needs the deferred single-branch step"
            assert not multi_shape, "grad_buckets: the bucket cuts are record counts of ONE input shape (use grad_buckets=1 with multi_shape)"
            self.warmup = max(self.warmup, 2)   # pass 1 finds the bucket cuts, pass 2 sizes the per-bucket tables
        self._cuts = None            # per bucket: (recorded products, registered chunks) when its last gradient was handed over
        self._defer_ctx = None       # the deferred_finishes() context kept open between the backward graph and the flush graphs
        self._flushed = (0, 0)
        self.btables = None          # per bucket (WgradTable, FinishTable)
        self.graph_flush = []
        self.ar_stream = torch.cuda.Stream(device=self.device) if self.nbuckets > 1 else None
        self.graph_fb: Optional[torch.cuda.CUDAGraph] = None
        self.graph_opt: Optional[torch.cuda.CUDAGraph] = None
        self.static_lq = self.static_gt = self.static_loss = None
        # shape-keyed cache: (lq shape, gt shape) -> the per-shape part of the state above
        self._shapes = {}
        self._active = None
        self.allreduce_ms = None   # set by time_allreduce(): events around the flat all-reduce (bench.py)

    # ---- pieces ------------------------------------------------------------------------------
    def _backward_open(self, loss, leaves):
        """bucketed mode: the backward with every weight-gradient product and finishing sum only RECORDED; the deferred context
        stays open (``_flush_bucket`` closes it after the last bucket).  On the very first pass it also finds the bucket cuts: a
        post-accumulate hook on every leaf notes the order in which autograd hands the gradients over and the library's record
        counts at that moment."""
        assert self._defer_ctx is None, "the previous step's buckets were not all flushed"
        marks, handles = [], []
        if self._cuts is None:
            from . import _capi
            lib = _capi.load()
            owner = {id(p): i for i, p in enumerate(self.params)}
            for name, (m, sh) in self.shadow.items():
                owner[id(sh)] = owner[id(m)]
            for t in leaves:
                handles.append(t.register_post_accumulate_grad_hook(
                    lambda t_, i=owner[id(t)]: marks.append((i, int(lib.oss_deferred_wgrads()), int(lib.oss_deferred_chunks())))))
        self._defer_ctx = _ops.deferred_finishes(wgrad_flusher=None)   # no budget flushes: they would shift the cut counts
        self._defer_ctx.__enter__()
        self._flushed = (0, 0)
        try:
            loss.backward()
        except BaseException:
            self._defer_ctx.__exit__(None, None, None)
            self._defer_ctx = None
            raise
        finally:
            for h in handles:
                h.remove()
        if self._cuts is None:
            self._probe = marks

    def _make_buckets(self):
        """from the first pass's marks: parameter order = order of the hand-over, K buckets of about equal size, the cut of a bucket
        = the record counts at its last member's hand-over (monotone: later buckets include everything recorded before)"""
        from .ddp import FlatGrads
        seen, order, at = set(), [], {}
        for i, nw, nc in self._probe:
            if i not in seen:
                seen.add(i)
                order.append(i)
            at[i] = (nw, nc)
        order += [i for i in range(len(self.params)) if i not in seen]   # (a parameter that never got a gradient: none in these nets)
        self._flat = FlatGrads(self.params, order=order, n_buckets=self.nbuckets, via_host=self.allreduce_via_host)
        self.nbuckets = self._flat.n_buckets
        cuts, hi = [], (0, 0)
        for members in self._flat.bucket_members:
            for i in members:
                if i in at:
                    hi = (max(hi[0], at[i][0]), max(hi[1], at[i][1]))
            cuts.append(hi)
        self._cuts = cuts
        self._probe = None
        shadow_of = {id(m): sh for m, sh in self.shadow.values()}
        mg_of = {id(m): g for m, g in zip(self._masters, self._master_grads)}
        self._bucket_shadows = [[(self.params[i], shadow_of[id(self.params[i])], mg_of[id(self.params[i])]) for i in members
                                 if id(self.params[i]) in shadow_of] for members in self._flat.bucket_members]

    def _flush_bucket(self, k: int):
        """finish the gradients of bucket ``k`` (grouped weight-gradient launch + finishing sums of everything recorded up to its
        cut), convert its shadow gradients, pack it into its slice of the flat buffer.  The last bucket takes whatever is left
        and closes the deferred context."""
        last = k == self.nbuckets - 1
        nw = 0 if last else max(0, self._cuts[k][0] - self._flushed[0])
        nc = 0 if last else max(0, self._cuts[k][1] - self._flushed[1])
        if self.btables is None:
            self.btables = [[None, None] for _ in range(self.nbuckets)]
        wt, ft = self.btables[k]
        if last or nw > 0:
            if _ops.pending_wgrads():
                need = _ops.pending_wgrad_table_bytes()
                if wt is None or wt.capacity < need:
                    assert not torch.cuda.is_current_stream_capturing(), "the weight-gradient tables must exist before the capture"
                    wt = self.btables[k][0] = _ops.WgradTable(self.device, need)
                _ops.flush_wgrads(wt, count=nw)
        if last or nc > 0:
            n = _ops.pending_finish_chunks()
            if n:
                if ft is None or ft.capacity < n:
                    assert not torch.cuda.is_current_stream_capturing(), "the finish tables must exist before the capture"
                    ft = self.btables[k][1] = _ops.FinishTable(self.device, n)
                _ops.flush_finishes(ft, count=nc)
        if not last:
            self._flushed = (self._cuts[k][0], self._cuts[k][1])
        trip = self._bucket_shadows[k]
        if trip:
            with torch.no_grad():
                torch._foreach_copy_([g for _, _, g in trip], [sh.grad for _, sh, _ in trip])
            for m, _, g in trip:
                m.grad = g
        self._flat.pack_bucket(k)
        if last:
            self._defer_ctx.__exit__(None, None, None)
            self._defer_ctx = None

    def _allreduce_bucket(self, k: int):
        """bucket k's exchange on the side stream, behind everything the main stream has queued so far (its flush graph)"""
        if self.world == 1:
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.ar_stream):
            self.ar_stream.wait_event(ev)
            if self.allreduce_ms is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._flat.allreduce_bucket(k)
                e1.record()
                self._ar_events.append((e0, e1))
            else:
                self._flat.allreduce_bucket(k)

    def _backward(self, loss, leaves, slot: int = 0):
        """backward of one (micro-)batch on the current stream; ``leaves``: the tensors whose .grad it fills"""
        if self.nbuckets > 1:
            return self._backward_open(loss, leaves)
        if not self.defer:
            with _ops.wgrad_side_stream(self.wside):
                loss.backward()
            return
        tables = self.wtables[slot] if self.wtables[slot] is not None else []
        self.wtables[slot] = tables
        used = [0]

        def flush_wgrads_now():
            # one pinned table per grouped launch of a backward: a captured copy node reads its host buffer at REPLAY time,
            # so two launches of one graph must not share one
            nb = _ops.pending_wgrad_table_bytes()
            k = used[0]
            if k == len(tables):
                tables.append(None)
            if tables[k] is None or tables[k].capacity < nb:
                assert not torch.cuda.is_current_stream_capturing(), "the weight-gradient tables must exist before the capture"
                tables[k] = _ops.WgradTable(self.device, nb)
            _ops.flush_wgrads(tables[k])
            used[0] = k + 1

        with _ops.deferred_finishes(wgrad_flusher=flush_wgrads_now if self.wside is None else None):
            with _ops.wgrad_side_stream(self.wside):
                loss.backward()
            if self.wside is not None:
                torch.cuda.current_stream().wait_stream(self.wside)
            if not torch.cuda.is_current_stream_capturing():   # eager warm-up steps: every deferred gradient was adopted
                lost = _ops.orphaned_deferred_outputs(leaves)
                if lost:
                    raise RuntimeError(f"{lost} deferred weight gradients were copied before the flush (see ops.py CONTRACT)")
            if _ops.pending_wgrads():   # the recorded weight-gradient products, as ONE grouped launch (ops/_common.py) --
                flush_wgrads_now()      # or the last of a few, when the operands held passed WGRAD_KEEP_BUDGET on the way
            self.wgrad_stats = dict(_ops.WGRAD_STATS, grouped_launches=used[0])
            n = _ops.pending_finish_chunks()
            if self.ftables[slot] is None or self.ftables[slot].capacity < n:   # first (eager, warm-up) step: sizes the table
                assert not torch.cuda.is_current_stream_capturing(), "the finish table must exist before the capture"
                self.ftables[slot] = _ops.FinishTable(self.device, n)
            _ops.flush_finishes(self.ftables[slot])

    def _fwd_bwd(self):
        if self.nmicro > 1:
            return self._fwd_bwd_micro()
        for p in self.params:  # host-side only: the backward then assigns instead of accumulating
            p.grad = None
        for s_ in self._shadows:
            s_.grad = None
        if self._shadows:
            with torch.no_grad():
                torch._foreach_copy_(self._shadows, self._masters)  # fp32 -> 16-bit, one multi-tensor kernel
        with torch.autocast("cuda", dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
            if self._shadows:
                out = torch.func.functional_call(self.net, self._shadow_map, (self.static_lq,))
            else:
                out = self.net(self.static_lq)
        loss = self.loss_fn(out.float(), self.static_gt)
        self._backward(loss, list(self.params) + self._shadows)
        if self.nbuckets > 1:
            return loss.detach()   # gradients are finished, converted and packed bucket by bucket (_flush_bucket)
        if self.wside is not None:
            torch.cuda.current_stream().wait_stream(self.wside)   # join before anything reads a weight gradient
        if self._shadows:
            with torch.no_grad():
                torch._foreach_copy_(self._master_grads, [s_.grad for s_ in self._shadows])
            for m, g in zip(self._masters, self._master_grads):
                m.grad = g
        if self.split:
            self._pack_grads()
        return loss.detach()

    def _fwd_bwd_micro(self):
        cur = torch.cuda.current_stream()
        M = self.nmicro
        B = self.static_lq.shape[0]
        assert B % M == 0, f"batch {B} does not split into {M} micro-batches"
        mb = B // M
        if self._shadows:
            with torch.no_grad():
                torch._foreach_copy_(self._shadows, self._masters)
        losses = []
        for m, st in enumerate(self.mstreams):
            leaves = self._leaf_sets[m]
            for t in leaves.values():
                t.grad = None
            st.wait_stream(cur)                      # fork
            with torch.cuda.stream(st):
                lq, gt = self.static_lq[m * mb:(m + 1) * mb], self.static_gt[m * mb:(m + 1) * mb]
                with torch.autocast("cuda", dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
                    out = torch.func.functional_call(self.net, leaves, (lq,))
                loss = self.loss_fn(out.float(), gt) / M     # mean over the whole batch = mean of the micro-batch means
                self._backward(loss, list(leaves.values()), m)
                losses.append(loss.detach())
        for st in self.mstreams:
            cur.wait_stream(st)                      # join
        shadowed = set(self._shadow_map)
        with torch.no_grad():
            plain = [n for n in self._names if n not in shadowed]
            g0 = [self._leaf_sets[0][n].grad for n in plain]
            for m in range(1, M):
                torch._foreach_add_(g0, [self._leaf_sets[m][n].grad for n in plain])
            if self._shadows:
                sn = list(self._shadow_map)
                torch._foreach_copy_(self._master_grads, [self._leaf_sets[0][n].grad for n in sn])   # 16-bit -> fp32
                for m in range(1, M):
                    torch._foreach_copy_(self._shadow_tmp, [self._leaf_sets[m][n].grad for n in sn])
                    torch._foreach_add_(self._master_grads, self._shadow_tmp)
            total = losses[0]
            for l_ in losses[1:]:
                total = total + l_
        named = dict(zip(plain, g0))
        mg = dict(zip(self._shadow_map, self._master_grads))
        for n, p in zip(self._names, self.params):
            p.grad = mg[n] if n in mg else named[n]
        if self.split:
            self._pack_grads()
        return total

    def _first_bucket_pass(self):
        """the very first warm-up pass of the bucketed mode: everything flushed at once (as the unbucketed step does), then the
        flat buffer is built in hand-over order and cut into buckets"""
        wt = _ops.WgradTable(self.device, _ops.pending_wgrad_table_bytes()) if _ops.pending_wgrads() else None
        if wt is not None:
            _ops.flush_wgrads(wt)
        n = _ops.pending_finish_chunks()
        if n:
            _ops.flush_finishes(_ops.FinishTable(self.device, n))
        self._defer_ctx.__exit__(None, None, None)
        self._defer_ctx = None
        if self._shadows:
            with torch.no_grad():
                torch._foreach_copy_(self._master_grads, [s_.grad for s_ in self._shadows])
            for m, g in zip(self._masters, self._master_grads):
                m.grad = g
        self._make_buckets()
        self._flat.pack()
        self._allreduce()

    def _pack_grads(self):
        if self._flat is None:
            assert not torch.cuda.is_current_stream_capturing(), "the flat gradient buffer must exist before the capture"
            from .ddp import FlatGrads
            self._flat = FlatGrads(self.params)
        self._flat.pack()

    def _opt_ema(self):
        if self.fopt is not None:
            self.fopt.step()
            return
        self.opt.step()
        if self.ema is not None:
            with torch.no_grad():  # model_ema(decay) of the reference
                torch._foreach_mul_(self.ema, self.ema_decay)
                torch._foreach_add_(self.ema, [p.detach() for p in self.params], alpha=1.0 - self.ema_decay)

    def _allreduce(self):
        if self.world > 1:   # gradients already sit in the flat buffer (see _pack_grads): one collective, mean over ranks
            self._flat.allreduce_mean()

    # ---- capture -----------------------------------------------------------------------------
    def _snapshot(self):
        """everything a training step changes: parameters, EMA, optimizer moments and step count"""
        st = [p.detach().clone() for p in self.params]
        if self.ema is not None:
            st += [e.clone() for e in self.ema]
        if self.fopt is not None:
            st += [t.clone() for t in self.fopt.exp_avg + self.fopt.exp_avg_sq + [self.fopt.state]]
        osd = None
        if self.opt is not None:
            # torch optimizer: clone every state tensor that exists NOW.  A second shape captured in the middle of a run
            # (multi_shape: the progressive patch schedule does that six times) must get the trained moments and step
            # count back after its warm-up steps, not zeros
            osd = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in stt.items()}
                   for p, stt in self.opt.state.items()}
        return st, osd

    def _restore(self, snap):
        st, osd = snap
        live = [p.detach() for p in self.params] + (list(self.ema) if self.ema is not None else [])
        if self.fopt is not None:
            live += self.fopt.exp_avg + self.fopt.exp_avg_sq + [self.fopt.state]
        with torch.no_grad():
            torch._foreach_copy_(live, st)
        if self.opt is not None:
            with torch.no_grad():
                for p, stt in self.opt.state.items():
                    saved = (osd or {}).get(id(p))
                    for k, v in stt.items():
                        if not torch.is_tensor(v):
                            continue
                        if saved is not None and k in saved:
                            v.copy_(saved[k])   # in place: a graph captured earlier keeps reading these tensors
                        else:
                            v.zero_()           # created by this (the first) warm-up: back to "no step yet"

    # ---- learning rate, training state ---------------------------------------------------------
    def set_lr(self, lr: float) -> None:
        """learning rate of the following steps (graph replays included): what the reference's
        ``update_learning_rate(current_iter)`` does through its scheduler before every ``optimize_parameters``
        (Deraining/basicsr/train.py:238, basicsr/models/base_model.py:183-205)."""
        if self.fopt is not None:
            self.fopt.set_lr(lr)
        else:
            with torch.no_grad():
                for g in self.opt.param_groups:
                    g["lr"].fill_(float(lr))
        self.lr = float(lr)

    def state_dict(self) -> dict:
        """what ``save_training_state`` keeps (base_model.py:312-334: iteration, optimizer state) plus the EMA weights
        (a second network in the reference, saved by ``save_network`` under ``params_ema``)"""
        osd = self.fopt.state_dict() if self.fopt is not None else self.opt.state_dict()
        return {"iter": self.iteration, "optimizers": [osd],
                "ema": [e.detach().clone() for e in self.ema] if self.ema is not None else None}

    def load_state_dict(self, sd: dict) -> None:
        """resume (``resume_training``, base_model.py:336-351).  Tensors are written IN PLACE, so graphs captured before
        the call keep working; hyper-parameters other than the learning rate are baked into captured launches and must
        match the ones this object was built with."""
        osd = sd["optimizers"][0]
        if self.fopt is not None:
            self.fopt.load_state_dict(osd)
            self.lr = self.fopt.lr
        else:
            lr_t = self.opt.param_groups[0]["lr"]
            cur = self.opt.state_dict()
            for k, ent in osd["state"].items():
                if k in cur["state"]:        # existing tensors (possibly captured): copy in place
                    for name, v in ent.items():
                        if torch.is_tensor(v):
                            cur["state"][k][name].copy_(v)
                else:
                    self.opt.state[self.params[k]] = {n: (v.to(self.device).clone() if torch.is_tensor(v) else v) for n, v in ent.items()}
            with torch.no_grad():
                lr_t.fill_(float(osd["param_groups"][0]["lr"]))
            self.lr = float(osd["param_groups"][0]["lr"])
        if self.ema is not None and sd.get("ema") is not None:
            with torch.no_grad():
                torch._foreach_copy_(self.ema, [e.to(self.device) for e in sd["ema"]])
        self.iteration = int(sd.get("iter", 0))

    def _key(self, lq, gt):
        return (tuple(lq.shape), tuple(gt.shape))

    def _stash(self):
        if self._active is not None:
            self._shapes[self._active] = (self.static_lq, self.static_gt, self.static_loss, self.graph_fb, self.ftables, self.wtables,
                                          self.graph_flush, self.btables)

    def _activate(self, key):
        if key == self._active:
            return
        self._stash()
        (self.static_lq, self.static_gt, self.static_loss, self.graph_fb, self.ftables, self.wtables, self.graph_flush,
         self.btables) = self._shapes[key]
        self._active = key

    @property
    def n_graphs(self) -> int:
        return len(self._shapes) + (1 if self._active is not None and self._active not in self._shapes else 0)

    def capture(self, lq: torch.Tensor, gt: torch.Tensor) -> None:
        """Capture the step for this (lq, gt) shape.  The eager warm-up steps (lazy initialisation, vendor solver search,
        LDS attributes, table sizing) run real updates, so parameters, EMA and optimizer state are put back afterwards:
        the first replay is the first update, as one ``optimize_parameters`` call per batch does in the reference."""
        key = self._key(lq, gt)
        if self._active is not None and key != self._active:
            if not self.multi_shape:
                raise RuntimeError("GraphedTrainStep was captured for another input shape; construct it with multi_shape=True "
                                   "to keep one graph per shape (progressive patch schedule)")
            self._stash()
            self.graph_fb, self.ftables, self.wtables = None, [None] * self.nmicro, [None] * self.nmicro
            self.graph_flush, self.btables = [], None
        self._active = key
        self.static_lq = lq.clone()
        self.static_gt = gt.clone()
        snap = self._snapshot()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture: lazy inits, MIOpen find, LDS attributes
            for _ in range(self.warmup):
                self._fwd_bwd()
                if self.nbuckets > 1:
                    if self._cuts is None:       # first pass of all: one flush of everything, then the buckets are defined
                        self._first_bucket_pass()
                    else:
                        for k in range(self.nbuckets):
                            self._flush_bucket(k)
                            self._allreduce_bucket(k)
                        torch.cuda.current_stream().wait_stream(self.ar_stream)
                else:
                    self._allreduce()
                self._opt_ema()
                torch.cuda.synchronize()   # the pointer tables of step k must not be rewritten while step k - 1 runs
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        torch.cuda.synchronize()
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self.static_loss = self._fwd_bwd()
            if not self.split:
                self._opt_ema()
        self.graph_flush = []
        if self.nbuckets > 1:   # one small graph per bucket, same memory pool (they read what the backward graph left recorded)
            for k in range(self.nbuckets):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, pool=self.graph_fb.pool()):
                    self._flush_bucket(k)
                self.graph_flush.append(gk)
        if self.split and self.graph_opt is None:   # one optimizer graph for every shape: it reads the flat gradient views
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt):
                self._opt_ema()
        self._stash()

    def time_allreduce(self, on: bool = True) -> None:
        """bench.py: bracket the flat all-reduce of every following step with events (``allreduce_ms`` = list of ms)"""
        self.allreduce_ms = [] if on else None
        self._ar_events = []

    # ---- replay ------------------------------------------------------------------------------
    def __call__(self, lq: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        key = self._key(lq, gt)
        if key not in self._shapes:
            self.capture(lq, gt)
        self._activate(key)
        self.static_lq.copy_(lq, non_blocking=True)
        self.static_gt.copy_(gt, non_blocking=True)
        if os.environ.get("VMAMBAIR_STEP_TRACE", "0") == "1" and self.split:   # diagnosis only: host seconds per phase, fenced
            import time
            t = [time.perf_counter()]
            self.graph_fb.replay(); torch.cuda.synchronize(); t.append(time.perf_counter())
            self._allreduce(); torch.cuda.synchronize(); t.append(time.perf_counter())
            self.graph_opt.replay(); torch.cuda.synchronize(); t.append(time.perf_counter())
            print(f"[step trace rank {dist.get_rank() if self.world > 1 else 0}] fwd+bwd graph {t[1] - t[0]:.3f} s, all-reduce "
                  f"{t[2] - t[1]:.3f} s, optimizer graph {t[3] - t[2]:.3f} s", flush=True)
            self.iteration += 1
            return self.static_loss
        self.graph_fb.replay()
        if self.nbuckets > 1:
            for k, gk in enumerate(self.graph_flush):
                gk.replay()
                self._allreduce_bucket(k)     # side stream: runs while the main stream flushes the next bucket
            if self.world > 1:
                torch.cuda.current_stream().wait_stream(self.ar_stream)
            self.graph_opt.replay()
        elif self.split:
            if self.allreduce_ms is not None and self.world > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._allreduce()
                e1.record()
                self._ar_events.append((e0, e1))
            else:
                self._allreduce()
            self.graph_opt.replay()
        self.iteration += 1
        return self.static_loss

    def collect_allreduce_ms(self):
        """-> mean milliseconds of the timed all-reduces (synchronises), or None"""
        if not getattr(self, "_ar_events", None):
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._ar_events]
        self._ar_events = []
        return sum(ms) / (len(ms) / self.nbuckets)   # per step: the buckets of a step are separate collectives
