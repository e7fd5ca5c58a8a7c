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
