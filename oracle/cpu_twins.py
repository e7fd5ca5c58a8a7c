"""CPU twins of every ``torch.ops.vmambair`` op (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

The product registers GPU kernels only.  ``install()`` registers, for the CPU dispatch key, the
oracle (selective scan: oracle/oss_scan_oracle.c; omni form: the same oracle on materialised
direction tensors) and plain PyTorch fp32 references (depth-wise conv, NCHW LayerNorm) so that the
host-side mirrors of the reference modules can be exercised without a GPU.  Importers: tests/ and
bench.py's cpu_baseline leg only.
"""
import torch
import torch.nn.functional as F

_CPU_LIB = None


def install():
    global _CPU_LIB
    if _CPU_LIB is not None:
        return
    import vmambair_amd.ops  # noqa: F401  defines the ops
    from oracle import oss_oracle

    chunk = vmambair_amd.ops.scan_chunk()

    def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows):
        return oss_oracle.scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, chunk=chunk)

    def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
        res = oss_oracle.scan_bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows)
        return [t if t is not None else torch.empty(0) for t in res]

    import torch.nn.functional as F

    def dw_fwd(x, weight, bias, act):  # plain PyTorch fp32 reference of the depth-wise conv (+ silu)
        pre = F.conv2d(x.float(), weight.float(), None if bias is None else bias.float(), padding=1, groups=x.shape[1])
        return [(F.silu(pre) if act else pre).to(x.dtype), pre.to(x.dtype) if act else torch.empty(0)]

    def dw_bwd(x, weight, dy, has_bias, pre=None, dx_into=None):
        xx = x.detach().float().requires_grad_()
        ww = weight.detach().float().requires_grad_()
        g = dy.float()
        if pre is not None and pre.numel():
            pp = pre.detach().float().requires_grad_()
            with torch.enable_grad():
                a = F.silu(pp)
            g = torch.autograd.grad(a, pp, g)[0]
        with torch.enable_grad():
            y = F.conv2d(xx, ww, None, padding=1, groups=x.shape[1])
        dx, dw = torch.autograd.grad(y, (xx, ww), g)
        db = g.sum(dim=(0, 2, 3)) if has_bias else torch.empty(0)
        dx = dx.to(x.dtype)
        if dx_into is not None and dx_into.dtype == dx.dtype:   # one half of the gradient of a split tensor (ops.PairGrad)
            dx_into.copy_(dx)        # mutated argument, empty return (the operator's contract: ops/dwconv.py)
            dx = x.new_empty(0)
        return [dx, dw, db]

    def _mirror(t, G_or_rows, start, per):  # flip time of groups >= start (per rows each)
        t = t.clone()
        t[:, start * per:] = t[:, start * per:].flip(-1)
        return t

    def omni_fwd(u, delta, A_log, B, C, D, delta_bias, delta_softplus, rev_group_start, u_row_mod):
        """oracle twin of the omni form: materialise what the kernels read implicitly"""
        A = -torch.exp(A_log.float())
        dim, G = A.shape[0], B.shape[1]
        per = dim // G
        uu = u.repeat(1, dim // u_row_mod, 1) if u_row_mod else u
        uu, dd = _mirror(uu, G, rev_group_start, per), _mirror(delta, G, rev_group_start, per)
        Bm, Cm = _mirror(B, G, rev_group_start, 1), _mirror(C, G, rev_group_start, 1)
        out, x = oss_oracle.scan_fwd(uu, dd, A, Bm, Cm, D, delta_bias, delta_softplus, 1, chunk=chunk)
        return [_mirror(out, G, rev_group_start, per), x]

    def merge4(out, H, W):
        Bsz, _, Dn, L = out.shape
        o = out.float()
        y = o[:, 0] + o[:, 2]
        y = y + o[:, 1].reshape(Bsz, Dn, W, H).transpose(2, 3).reshape(Bsz, Dn, L)
        y = y + o[:, 3].reshape(Bsz, Dn, W, H).transpose(2, 3).reshape(Bsz, Dn, L)
        return y.view(Bsz, Dn, H, W)

    def omni_bwd(u, delta, A_log, B, C, D, delta_bias, dout, x, delta_softplus, rev_group_start, u_row_mod, dout_row_mod):
        A = -torch.exp(A_log.float())
        dim, G = A.shape[0], B.shape[1]
        per = dim // G
        if dout_row_mod:
            dout = dout.repeat(1, dim // dout_row_mod, 1)
        uu = u.repeat(1, dim // u_row_mod, 1) if u_row_mod else u
        uu, dd = _mirror(uu, G, rev_group_start, per), _mirror(delta, G, rev_group_start, per)
        Bm, Cm = _mirror(B, G, rev_group_start, 1), _mirror(C, G, rev_group_start, 1)
        gg = _mirror(dout, G, rev_group_start, per)
        du, ddl, dA, dB, dC, dD, db = oss_oracle.scan_bwd(uu, dd, A, Bm, Cm, D, delta_bias, gg, None, delta_softplus)
        res = [_mirror(du, G, rev_group_start, per), _mirror(ddl, G, rev_group_start, per), dA * A,
               _mirror(dB, G, rev_group_start, 1), _mirror(dC, G, rev_group_start, 1), dD, db]
        return [t if t is not None else torch.empty(0) for t in res]

    def _ln_ref(x, weight, bias, gate):
        """plain PyTorch fp32 reference of the NCHW LayerNorm (+ silu gate), differentiable"""
        xf = x.float()
        mu = xf.mean(1, keepdim=True)
        var = xf.var(1, keepdim=True, unbiased=False)
        rstd = (var + 1e-5).rsqrt()
        w = weight.float().view(1, -1, 1, 1)
        y = (xf - mu) * rstd * w + bias.float().view(1, -1, 1, 1) if bias is not None else xf * rstd * w
        if gate is not None:
            y = y * F.silu(gate.float())
        return y, mu, rstd

    codes = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}

    def ln_fwd(x, weight, bias, gate, out_code, want_pool=False):
        y, mu, rstd = _ln_ref(x, weight, bias, gate)
        B, C, H, W = x.shape
        return [y.to(codes[out_code]), mu.reshape(B, H * W), rstd.reshape(B, H * W)] + ([torch.empty(0)] if want_pool else [])

    def ln_bwd(x, weight, bias, gate, dy, mean, rstd, skip_grad=None, dgate_into=None, dy_mul=None, dy_add=None, add_scale=1.0):
        if dy_add is not None:   # the channel gate's backward folded into the load (ops/layernorm.py)
            dyf = dy.float()
            if dy_mul is not None:
                dyf = dyf * (1.0 + dy_mul.float())[:, :, None, None]
            dy = (dyf + add_scale * dy_add.float()[:, :, None, None])
        out_dt = dy.dtype if dy_add is None else (gate.dtype if gate is not None else x.dtype)
        leaves = [x.detach().float().requires_grad_(), weight.detach().float().requires_grad_()]
        bb = bias.detach().float().requires_grad_() if bias is not None else None
        gg = gate.detach().float().requires_grad_() if gate is not None else None
        with torch.enable_grad():
            y, _, _ = _ln_ref(leaves[0], leaves[1], bb, gg)
        ins = leaves + ([bb] if bb is not None else []) + ([gg] if gg is not None else [])
        gr = list(torch.autograd.grad(y, ins, dy.float()))
        dx, dw = gr.pop(0), gr.pop(0)
        db = gr.pop(0) if bb is not None else torch.empty(0)
        dg = gr.pop(0).to(out_dt) if gg is not None else torch.empty(0)
        if skip_grad is not None:
            dx = dx + skip_grad.float()
        if dgate_into is not None and gg is not None and dgate_into.dtype == dg.dtype:
            dgate_into.copy_(dg)      # mutated argument, empty return (the operator's contract: ops/layernorm.py)
            dg = torch.empty(0)
        return [dx.to(x.dtype), dg, dw, db]

    def _core_ref(x, wx, wdt, A_logs, Ds, dt_bias):
        """literal reference data flow of SS2D_1.forward_core up to out_norm (MambaSISR6_arch.py:395-431) on
        the oracle scan, differentiable: four materialised directions, flips and transposes"""
        B, D, H, W = x.shape
        L = H * W
        R, N = wdt.shape[2], A_logs.shape[1]
        hw = x.flatten(2, 3)
        wh = x.transpose(2, 3).contiguous().flatten(2, 3)
        fwd2 = torch.stack([hw, wh], dim=1)
        xs = torch.cat([fwd2, fwd2.flip(-1)], dim=1)
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, wx)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, wdt)
        out = oss_oracle.OracleScanFn.apply(xs.reshape(B, -1, L), dts.reshape(B, -1, L), -torch.exp(A_logs), Bs.contiguous(),
                                            Cs.contiguous(), Ds, dt_bias.reshape(-1), True).view(B, 4, -1, L)
        inv = out[:, 2:4].flip(-1)
        wh_y = out[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        invwh_y = inv[:, 1].reshape(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        return (out[:, 0] + inv[:, 0] + wh_y + invwh_y).view(B, D, H, W)

    def core_fwd(x, wx, wdt, A_logs, Ds, dt_bias, want_hs=False):
        B, D, H, W = x.shape
        R = wdt.shape[2]
        xf = x.float()
        with torch.no_grad():
            y = _core_ref(xf, wx.float(), wdt.float(), A_logs.float(), Ds.float(), dt_bias.float())
            x2 = torch.stack([xf.flatten(2, 3), xf.transpose(2, 3).contiguous().flatten(2, 3)], dim=1)
            xdbl = torch.cat([torch.einsum("bjdl,jcd->bjcl", x2, wx[0:2].float()),
                              torch.einsum("bjdl,jcd->bjcl", x2, wx[2:4].float())], dim=1)
            dts = torch.einsum("bkrl,kdr->bkdl", xdbl[:, :, :R], wdt.float()).reshape(B, 4 * D, H * W)
        return [y, x2.to(x.dtype), xdbl.to(x.dtype), dts.to(x.dtype), torch.empty(0), torch.empty(0)]

    def core_bwd(dy, x2, xdbl, dts, states, wx, wdt, A_logs, Ds, dt_bias, lane_states=None):
        B, _, D, L = x2.shape
        H, W = dy.shape[2], dy.shape[3]
        leaves = [t.detach().float().requires_grad_() for t in (x2[:, 0].reshape(B, D, H, W), wx, wdt, A_logs, Ds, dt_bias)]
        with torch.enable_grad():
            y = _core_ref(*leaves)
        gr = torch.autograd.grad(y, leaves, dy.float())
        return [gr[0].to(x2.dtype), gr[1], gr[2], gr[3], gr[4], gr[5]]

    def _chan_ref(y2, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b, mul_mode):
        """literal reference data flow of the channel branch + gate (MambaSISR6_arch.py:438-496; RealSR form
        MambaRealSR11_arch.py:758-817) on the oracle scan, differentiable: stack / flip of the two directions"""
        b, d, H, W = y2.shape
        pooled = y2.mean(dim=(2, 3))
        if cin_w is not None:
            dc = cin_w.shape[0]
            seq = F.conv2d(pooled.view(b, 1, d, 1), cin_w.view(dc, 1, 1, 1), cin_b).squeeze(-1)      # (b, dc, L = d)
        else:
            dc = 1
            seq = pooled.view(b, 1, d)
        Rc, N = Wdtc.shape[2], A_logs.shape[1]
        xsc = torch.stack([seq, seq.flip(-1)], dim=1)
        z = torch.einsum("bkdl,kcd->bkcl", xsc, Wxc)
        dts, Bs, Cs = torch.split(z, [Rc, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, Wdtc).contiguous()
        out = oss_oracle.OracleScanFn.apply(xsc.reshape(b, -1, d), dts.view(b, -1, d), -torch.exp(A_logs), Bs.contiguous(),
                                            Cs.contiguous(), Dsc, dt_bias.reshape(-1), True).view(b, 2, dc, d)
        y = out[:, 0] + out[:, 1].flip(-1)
        if cout_w is not None:
            y = F.conv2d(y.unsqueeze(-1), cout_w.view(1, dc, 1, 1), cout_b).view(b, d)
        else:
            y = y.reshape(b, d)
        c = F.layer_norm(y, (d,), cn_w, cn_b, 1e-5).view(b, d, 1, 1)
        return (y2 * c + y2) if mul_mode else (y2 + c), c.view(b, d)

    def chan_fwd(y2, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b, mul_mode, pool_part=None):
        f = lambda t: None if t is None else t.float()
        with torch.no_grad():
            out, c = _chan_ref(y2.float(), f(cin_w), f(cin_b), Wxc.float(), Wdtc.float(), dt_bias.float(), A_logs.float(),
                               Dsc.float(), f(cout_w), f(cout_b), cn_w.float(), cn_b.float(), mul_mode)
        e = torch.empty(0)
        return [out.to(y2.dtype), c, e, e, e, e, e, e, e]

    def chan_bwd(g, y2, c, pooled, zt, dts, hs, y, yc, stat, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b,
                 cn_w, cn_b, mul_mode, fold=False):
        prm = [cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b]
        leaves = [y2.detach().float().requires_grad_()] + [None if t is None else t.detach().float().requires_grad_() for t in prm]
        with torch.enable_grad():
            out, _ = _chan_ref(*leaves, mul_mode)
        live = [t for t in leaves if t is not None]
        gr = dict(zip([id(t) for t in live], torch.autograd.grad(out, live, g.float(), allow_unused=True)))
        get = lambda t, n: torch.zeros(n) if t is None else (gr[id(t)] if gr[id(t)] is not None else torch.zeros_like(t)).reshape(-1)
        dc = Wdtc.shape[1]
        lv = leaves[1:]
        # flat layout of oss_chan_bwd: cn_w, cn_b, cout_w, cout_b, A_logs, Dsc, dt_bias, Wdtc, Wxc, cin_w, cin_b
        flat = torch.cat([get(lv[9], 0), get(lv[10], 0), get(lv[7], dc), get(lv[8], 1), get(lv[5], 0), get(lv[6], 0),
                          get(lv[4], 0), get(lv[3], 0), get(lv[2], 0), get(lv[0], dc), get(lv[1], dc)])
        dy2 = gr[id(leaves[0])]
        if fold:   # d pooled alone: dy2 = g * (1 + c) [or g] + d pooled / (H W)
            direct = g.float() * (1.0 + c.float())[:, :, None, None] if mul_mode else g.float()
            return [(dy2 - direct).sum(dim=(2, 3)), flat]
        return [dy2.to(y2.dtype), flat]

    def gg_fwd(h):
        x1, x2 = h.float().chunk(2, dim=1)
        return (F.gelu(x1) * x2).to(h.dtype)

    def gg_bwd(h, dout):
        hh = h.detach().float().requires_grad_()
        with torch.enable_grad():
            x1, x2 = hh.chunk(2, dim=1)
            o = F.gelu(x1) * x2
        return torch.autograd.grad(o, hh, dout.float())[0].to(h.dtype)

    _CPU_LIB = torch.library.Library("vmambair", "IMPL")
    _CPU_LIB.impl("gelu_gate_fwd", gg_fwd, "CPU")
    _CPU_LIB.impl("gelu_gate_bwd", gg_bwd, "CPU")
    _CPU_LIB.impl("chan_gate_fwd", chan_fwd, "CPU")
    _CPU_LIB.impl("chan_gate_bwd", chan_bwd, "CPU")
    _CPU_LIB.impl("ss2d_core_fwd", core_fwd, "CPU")
    _CPU_LIB.impl("ss2d_core_bwd", core_bwd, "CPU")
    _CPU_LIB.impl("ln_nchw_fwd", ln_fwd, "CPU")
    _CPU_LIB.impl("ln_nchw_bwd", ln_bwd, "CPU")
    _CPU_LIB.impl("omni_scan_fwd", omni_fwd, "CPU")
    _CPU_LIB.impl("omni_scan_bwd", omni_bwd, "CPU")
    _CPU_LIB.impl("merge4", merge4, "CPU")
    _CPU_LIB.impl("selective_scan_fwd", fwd, "CPU")
    _CPU_LIB.impl("selective_scan_bwd", bwd, "CPU")
    _CPU_LIB.impl("dwconv3x3_fwd", dw_fwd, "CPU")
    _CPU_LIB.impl("dwconv3x3_bwd", dw_bwd, "CPU")


