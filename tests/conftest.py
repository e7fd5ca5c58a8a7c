"""Shared test plumbing.

* registers the ``gpu`` marker (tests that need a real MI355X);
* golden-vector loader (tests/golden/*.npz, produced by tests/golden/make_golden.py from the
  reference);
* ``oracle_cpu_kernel``: registers the CPU oracle (oracle/) as the *CPU* kernel of
  ``torch.ops.vmambair.selective_scan_fwd/_bwd`` for the duration of the test session, so that the
  host-side mirrors (SS2D_1, MamberBlock, the UNets, DDP wiring) can be exercised without a GPU.
  This lives in tests/ only: the product registers a GPU kernel and nothing else.
"""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k]) for k in z.files}


def golden_files(prefix):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


_CPU_LIB = None


def install_oracle_cpu_kernel():
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

    def dw_fwd(x, weight, bias):  # plain PyTorch fp32 reference of the depth-wise conv
        return F.conv2d(x.float(), weight.float(), None if bias is None else bias.float(), padding=1,
                        groups=x.shape[1]).to(x.dtype)

    def dw_bwd(x, weight, dy, has_bias):
        xx = x.detach().float().requires_grad_()
        ww = weight.detach().float().requires_grad_()
        with torch.enable_grad():
            y = F.conv2d(xx, ww, None, padding=1, groups=x.shape[1])
        dx, dw = torch.autograd.grad(y, (xx, ww), dy.float())
        db = dy.float().sum(dim=(0, 2, 3)) if has_bias else torch.empty(0)
        return [dx.to(x.dtype), dw, db]

    def _mirror(t, G_or_rows, start, per):  # flip time of groups >= start (per rows each)
        t = t.clone()
        t[:, start * per:] = t[:, start * per:].flip(-1)
        return t

    def omni_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, rev_group_start, u_row_mod):
        """oracle twin of the omni form: materialise what the kernels read implicitly"""
        dim, G = A.shape[0], B.shape[1]
        per = dim // G
        uu = u.repeat(1, dim // u_row_mod, 1) if u_row_mod else u
        uu, dd = _mirror(uu, G, rev_group_start, per), _mirror(delta, G, rev_group_start, per)
        Bm, Cm = _mirror(B, G, rev_group_start, 1), _mirror(C, G, rev_group_start, 1)
        out, x = oss_oracle.scan_fwd(uu, dd, A, Bm, Cm, D, delta_bias, delta_softplus, 1, chunk=chunk)
        return [_mirror(out, G, rev_group_start, per), x]

    def omni_bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, rev_group_start, u_row_mod):
        dim, G = A.shape[0], B.shape[1]
        per = dim // G
        uu = u.repeat(1, dim // u_row_mod, 1) if u_row_mod else u
        uu, dd = _mirror(uu, G, rev_group_start, per), _mirror(delta, G, rev_group_start, per)
        Bm, Cm = _mirror(B, G, rev_group_start, 1), _mirror(C, G, rev_group_start, 1)
        gg = _mirror(dout, G, rev_group_start, per)
        du, ddl, dA, dB, dC, dD, db = oss_oracle.scan_bwd(uu, dd, A, Bm, Cm, D, delta_bias, gg, None, delta_softplus)
        res = [_mirror(du, G, rev_group_start, per), _mirror(ddl, G, rev_group_start, per), dA,
               _mirror(dB, G, rev_group_start, 1), _mirror(dC, G, rev_group_start, 1), dD, db]
        return [t if t is not None else torch.empty(0) for t in res]

    _CPU_LIB = torch.library.Library("vmambair", "IMPL")
    _CPU_LIB.impl("omni_scan_fwd", omni_fwd, "CPU")
    _CPU_LIB.impl("omni_scan_bwd", omni_bwd, "CPU")
    _CPU_LIB.impl("selective_scan_fwd", fwd, "CPU")
    _CPU_LIB.impl("selective_scan_bwd", bwd, "CPU")
    _CPU_LIB.impl("dwconv3x3_fwd", dw_fwd, "CPU")
    _CPU_LIB.impl("dwconv3x3_bwd", dw_bwd, "CPU")


@pytest.fixture(scope="session")
def oracle_cpu_kernel():
    install_oracle_cpu_kernel()
    yield


def assert_close(a, b, rtol, atol, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    diff = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = diff > tol
    if bad.any():
        i = int(torch.argmax(diff - tol))
        raise AssertionError(f"{what}: {int(bad.sum())}/{a.numel()} elements off; worst |diff|="
                             f"{diff.flatten()[i].item():.3e} at flat index {i} (got {a.flatten()[i].item():.6e}, "
                             f"want {b.flatten()[i].item():.6e}), max|ref|={b.abs().max().item():.3e}")
