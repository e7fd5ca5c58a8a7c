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


def install_oracle_cpu_kernel():
    from oracle import cpu_twins
    cpu_twins.install()


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


def reseed_parameters(net, seed=0):
    """Overwrite every parameter with values that depend only on (seed, parameter NAME, shape): the reference's arch (inside
    tests/golden/make_golden.py, build container) and this repo's mirror (here and on the GPU box) then hold identical weights
    without a 48 MB state dict in the fixture.  CPU generator, so the values do not depend on the device either.
    Scales follow what the reference's initialisers produce: fan-in uniform weights, A_log in [0, log 16], dt biases whose
    softplus is a small step, norm / skip scales around 1."""
    import math
    import zlib
    with torch.no_grad():
        for name, p in net.named_parameters():
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
            r = torch.rand(p.shape, generator=g, dtype=torch.float32)
            leaf = name.rsplit(".", 1)[-1]
            if "A_log" in leaf:
                v = r * math.log(16.0)
            elif leaf in ("dt_projs_bias", "dt_projs_biasc") or leaf.startswith("dt_") and leaf.endswith("bias"):
                v = -5.0 + 3.0 * r
            elif leaf.startswith("Ds"):
                v = 0.5 + r
            elif p.dim() == 1:
                v = (0.5 + r) if leaf == "weight" else 0.2 * (r - 0.5)
            else:
                fan_in = p.shape[-1] if ("projs_weight" in leaf or "proj_weight" in leaf) else p[0].numel()
                v = (2.0 * r - 1.0) * math.sqrt(3.0 / max(1, fan_in))
            p.copy_(v.to(p.dtype))
    return net


def seeded_tensor(tag, shape, seed=0):
    import zlib
    g = torch.Generator().manual_seed((zlib.crc32(tag.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
    return torch.rand(shape, generator=g, dtype=torch.float32)
