#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

Needs /root/reference (read-only); never runs on the GPU box.  Nothing of the reference is
copied: its Python is imported from where it lies, executed on the CPU, and only input/output
tensors are stored (small .npz files).  Re-run with:  python tests/golden/make_golden.py

What is imported from the reference
  * ``selective_scan_ref`` -- the reference's pure-PyTorch sequential scan
    (Mamba/kernels/selective_scan/test_selective_scan.py:168-234).  The test module cannot be
    imported as a whole (it imports CUDA extensions at module level, :319-359, and rebinds the
    name), so only that FunctionDef is compiled out of the parsed module.
  * the arch files (SRGAN/VmambaIR/archs/MambaSISR6_arch.py, Deraining/basicsr/models/archs/
    mamber32_arch.py, mamber33_arch.py, RealSR/VmambaIR/archs/MambaRealSR11_arch.py) with their
    missing third-party imports stubbed and ``selective_scan_cuda_core`` bound to a module whose
    ``fwd`` runs ``selective_scan_ref`` and whose ``bwd`` differentiates it with autograd.

Fixture families:
  g1_scan_*.npz     scan-level: inputs, out, last_state and all grads
  g2_perm.npz       the four spatial direction maps and the merge, on integer data (bit-exact)
  g3_block_*.npz    OSS block (SS2D_1 / MamberBlock variants): state_dict, input, output, grads
  g4_net_*.npz      whole small UNet forward
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from einops import rearrange, repeat

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------------
# reference import machinery
# --------------------------------------------------------------------------------------------
def load_selective_scan_ref():
    path = f"{REF}/Mamba/kernels/selective_scan/test_selective_scan.py"
    tree = ast.parse(open(path).read(), filename=path)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "selective_scan_ref"]
    assert len(wanted) == 1
    mod = ast.Module(body=wanted, type_ignores=[])
    ns = {"torch": torch, "F": F, "rearrange": rearrange, "repeat": repeat}
    exec(compile(mod, path, "exec"), ns)
    return ns["selective_scan_ref"]


selective_scan_ref = load_selective_scan_ref()


def ref_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows):
    out, last = selective_scan_ref(u, delta, A, B, C, D, None, delta_bias, delta_softplus, True)
    n = A.shape[1]
    x = torch.zeros(u.shape[0], u.shape[1], 1, 2 * n)
    x[:, :, 0, 1::2] = last
    return [out, x]


def ref_bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
    leaves = [t.detach().clone().requires_grad_(True) if t is not None else None
              for t in (u, delta, A, B, C, D, delta_bias)]
    with torch.enable_grad():
        out = selective_scan_ref(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5], None,
                                 leaves[6], delta_softplus, False)
        present = [t for t in leaves if t is not None]
        grads = list(torch.autograd.grad(out, present, dout))
    res = []
    for t in leaves:
        res.append(grads.pop(0) if t is not None else None)
    return res


def install_stubs():
    core = types.ModuleType("selective_scan_cuda_core")
    core.fwd, core.bwd = ref_fwd, ref_bwd
    sys.modules["selective_scan_cuda_core"] = core
    fv = types.ModuleType("fvcore")
    fvnn = types.ModuleType("fvcore.nn")
    fvnn.flop_count = lambda *a, **k: ({}, {})
    fvnn.parameter_count = lambda *a, **k: {"": 0}
    fv.nn = fvnn
    sys.modules["fvcore"], sys.modules["fvcore.nn"] = fv, fvnn

    class _Reg:
        def register(self, *a, **k):
            return lambda cls: cls

    for name in ("basicsr", "basicsr.utils", "basicsr.utils.registry"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["basicsr.utils.registry"].ARCH_REGISTRY = _Reg()


def load_by_path(name, path, pkg_alias=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_arch(tree):
    """tree in {'SRGAN','RealSR','mamber32','mamber33'}"""
    install_stubs()
    if tree in ("SRGAN", "RealSR"):
        for name in ("VmambaIR", "VmambaIR.archs"):
            sys.modules[name] = types.ModuleType(name)
        common = load_by_path("VmambaIR.archs.common", f"{REF}/{tree}/VmambaIR/archs/common.py")
        sys.modules["VmambaIR.archs"].common = common
        fname = "MambaSISR6_arch.py" if tree == "SRGAN" else "MambaRealSR11_arch.py"
        return load_by_path(f"ref_{tree}_arch", f"{REF}/{tree}/VmambaIR/archs/{fname}")
    return load_by_path(f"ref_{tree}_arch", f"{REF}/Deraining/basicsr/models/archs/{tree}_arch.py")


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items() if v is not None})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------------------------
# G1 -- scan level.  Input distributions of test_selective_scan.py:406-441.
# --------------------------------------------------------------------------------------------
def g1_case(tag, batch, dim, N, G, L, softplus, has_D, has_bias, itype=torch.float32, seed=0):
    torch.random.manual_seed(seed)
    A = (-0.5 * torch.rand(dim, N)).requires_grad_()
    Bm = torch.randn(batch, G, N, L).to(itype).float().requires_grad_()
    Cm = torch.randn(batch, G, N, L).to(itype).float().requires_grad_()
    D = torch.randn(dim).requires_grad_() if has_D else None
    bias = (0.5 * torch.rand(dim)).requires_grad_() if has_bias else None
    u = torch.randn(batch, dim, L).to(itype).float().requires_grad_()
    delta = (0.5 * torch.rand(batch, dim, L)).to(itype).float().requires_grad_()
    # inputs are stored as fp32 values exactly representable in `itype`; the reference upcasts
    # 16-bit inputs to fp32 before computing (test_selective_scan.py:183-200)
    out, last = selective_scan_ref(u, delta, A, Bm, Cm, D, None, bias, softplus, True)
    g = torch.randn_like(out).to(itype).float()
    out.backward(g)
    save(f"g1_scan_{tag}.npz", u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=bias,
         delta_softplus=np.array(softplus), itype=np.array(str(itype).split(".")[-1]),
         out=out, last_state=last, dout=g,
         du=u.grad, ddelta=delta.grad, dA=A.grad, dB=Bm.grad, dC=Cm.grad,
         dD=None if D is None else D.grad, ddelta_bias=None if bias is None else bias.grad)


def make_g1():
    shapes = {"s64": (2, 8, 16, 4, 64), "odd100": (2, 8, 16, 4, 100), "chan48": (2, 8, 16, 2, 48)}
    for tag, (b, d, n, g, L) in shapes.items():
        for sp in (False, True):
            for extras in (False, True):
                g1_case(f"{tag}_sp{int(sp)}_db{int(extras)}", b, d, n, g, L, sp, extras, extras)
    # two reference chunks (2048+37); one variant only, smaller N to keep the file small
    g1_case("twochunk2085_sp1_db1", 1, 4, 8, 2, 2085, True, True, True)
    # dstate=1 as in the reference's own grid (test_selective_scan.py:365), one group
    g1_case("n1_g1_sp1_db1", 2, 8, 1, 1, 96, True, True, True)
    # 16-bit inputs
    g1_case("bf16_s64_sp1_db1", 2, 8, 16, 4, 64, True, True, True, itype=torch.bfloat16)
    g1_case("fp16_s64_sp1_db1", 2, 8, 16, 4, 64, True, True, True, itype=torch.float16)
    # values straddling the softplus threshold (delta + bias around 20)
    torch.random.manual_seed(1)
    g1_threshold()


def g1_threshold():
    batch, dim, N, G, L = 1, 4, 4, 2, 40
    A = (-0.05 * torch.rand(dim, N)).requires_grad_()
    Bm = torch.randn(batch, G, N, L).requires_grad_()
    Cm = torch.randn(batch, G, N, L).requires_grad_()
    D = torch.randn(dim).requires_grad_()
    bias = (0.5 * torch.rand(dim)).requires_grad_()
    u = torch.randn(batch, dim, L).requires_grad_()
    delta = (19.0 + 2.0 * torch.rand(batch, dim, L)).requires_grad_()
    out, last = selective_scan_ref(u, delta, A, Bm, Cm, D, None, bias, True, True)
    g = torch.randn_like(out)
    out.backward(g)
    save("g1_scan_threshold_sp1_db1.npz", u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=bias,
         delta_softplus=np.array(True), itype=np.array("float32"), out=out, last_state=last, dout=g,
         du=u.grad, ddelta=delta.grad, dA=A.grad, dB=Bm.grad, dC=Cm.grad, dD=D.grad, ddelta_bias=bias.grad)


# --------------------------------------------------------------------------------------------
# G2 -- direction maps + merge on integers (MambaSISR6_arch.py:401-404,427-430)
# --------------------------------------------------------------------------------------------
def make_g2():
    arch = load_arch("SRGAN")
    torch.manual_seed(0)
    m = arch.SS2D_1(d_model=2, ssm_ratio=1)
    m.out_norm = torch.nn.Identity()
    Bsz, D, H, W = 1, 2, 3, 5
    x = torch.arange(Bsz * D * H * W, dtype=torch.float32).view(Bsz, D, H, W) + 1.0
    captured = {}
    out_y = (torch.arange(Bsz * 4 * D * H * W, dtype=torch.float32) * 3.0 + 7.0).view(Bsz, 4 * D, H * W)

    def fake_scan(u, delta, A, Bm, Cm, Dm=None, delta_bias=None, delta_softplus=False, nrows=1):
        captured["xs"] = u.detach().clone()
        return out_y.clone()

    arch.selective_scan_fn_v1 = fake_scan
    y = m.forward_corev1(x)
    save("g2_perm.npz", x=x, xs=captured["xs"], out_y=out_y, y=y)
    # channel directions (MambaSISR6_arch.py:453,473): integers through flip/merge
    p = torch.arange(2 * 7, dtype=torch.float32).view(2, 1, 7) * 2.0 + 1.0  # (B, rows, L=D)
    xsc = torch.stack([p, torch.flip(p, dims=[-1])], dim=1).view(2, -1, 7)
    oy = (torch.arange(2 * 2 * 7, dtype=torch.float32) * 5.0 - 11.0).view(2, 2, 1, 7)
    yc = oy[:, 0] + torch.flip(oy[:, 1], dims=[-1])
    save("g2_perm_channel.npz", p=p, xsc=xsc, out_y=oy, y=yc)


# --------------------------------------------------------------------------------------------
# G3 -- block level
# --------------------------------------------------------------------------------------------
def run_block(arch, tag, cls_name, dim, shape, seed=0, **kw):
    torch.manual_seed(seed)
    cls = getattr(arch, cls_name)
    if cls_name == "MamberBlock":
        m = cls(dim=dim, num_heads=1, ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias")
    else:
        m = cls(d_model=dim, ssm_ratio=1)
    # make the learned vectors non-trivial so that parity is not vacuous (fresh init has
    # LayerNorm weight=1/bias=0, Ds=1)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith(("body.weight", "body.bias", "Ds", "Dsc")):
                p_.data = p_.data.clone() + 0.1 * torch.randn(p_.shape)
    x = torch.randn(*shape).requires_grad_()
    y = m(x)
    g = torch.randn_like(y)
    y.backward(g)
    arrays = {"x": x, "y": y, "dy": g, "dx": x.grad}
    for k, v in m.state_dict().items():
        arrays["sd." + k] = v
    for k, p_ in m.named_parameters():
        arrays["grad." + k] = p_.grad if p_.grad is not None else torch.zeros_like(p_)
    save(f"g3_block_{tag}.npz", **arrays)


def make_g3():
    sr = load_arch("SRGAN")
    run_block(sr, "srgan_ss2d_d48", "SS2D_1", 48, (1, 48, 12, 10))
    run_block(sr, "srgan_mamber_d48", "MamberBlock", 48, (2, 48, 12, 10))
    m32 = load_arch("mamber32")
    run_block(m32, "mamber32_d48", "MamberBlock", 48, (1, 48, 10, 12))
    m33 = load_arch("mamber33")
    run_block(m33, "mamber33_d48", "MamberBlock", 48, (1, 48, 10, 12))
    rs = load_arch("RealSR")
    run_block(rs, "realsr_mamber_d48", "MamberBlock", 48, (1, 48, 12, 10))


# --------------------------------------------------------------------------------------------
# G4 -- net level
# --------------------------------------------------------------------------------------------
def make_g4():
    sr = load_arch("SRGAN")
    torch.manual_seed(0)
    net = sr.MambaSISR6(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    x = torch.rand(1, 3, 16, 24)
    with torch.no_grad():
        y = net(x)
    arrays = {"x": x, "y": y}
    for k, v in net.state_dict().items():
        arrays["sd." + k] = v
    save("g4_net_mambasisr6_d8.npz", **arrays)
    m32 = load_arch("mamber32")
    torch.manual_seed(0)
    net = m32.Mamber32(dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    x = torch.rand(1, 3, 16, 24)
    with torch.no_grad():
        y = net(x)
    arrays = {"x": x, "y": y}
    for k, v in net.state_dict().items():
        arrays["sd." + k] = v
    save("g4_net_mamber32_d8.npz", **arrays)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4"]
    if "g1" in which:
        make_g1()
    if "g2" in which:
        make_g2()
    if "g3" in which:
        make_g3()
    if "g4" in which:
        make_g4()
