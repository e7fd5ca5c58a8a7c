"""The C-ABI library: loads without a GPU, exports every symbol include/vmambair_oss.h declares,
and the ctypes structures have the layout the header gives them.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

import vmambair_amd
from vmambair_amd import _build, _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vmambair_oss.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oss_[a-z_0-9]+)\s*\(", src)))


def test_library_is_built_and_loads():
    assert os.path.exists(_build.LIB_PATH), "run __graft_entry__.build() first"
    lib = _capi.load()
    assert lib.oss_version().startswith(b"vmambair_oss")
    assert lib.oss_scan_chunk() == 256
    assert lib.oss_scan_num_chunks(1) == 1 and lib.oss_scan_num_chunks(256) == 1
    assert lib.oss_scan_num_chunks(257) == 2 and lib.oss_scan_num_chunks(4096) == 16


def test_every_declared_symbol_is_exported():
    declared = _declared_functions()
    assert declared == sorted(_capi.SYMBOLS)
    lib = ctypes.CDLL(_build.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_struct_layout_matches_header():
    """Compile a tiny C probe against the header and compare sizeof/offsetof with ctypes."""
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "vmambair_oss.h"
int main(void) {
  printf("%zu %zu ", offsetof(oss_scan_fwd_params, workspace), offsetof(oss_scan_fwd_params, workspace_bytes));
  printf("%zu %zu %zu %zu %zu %zu ", offsetof(oss_scan_fwd_params, dt_weight), offsetof(oss_scan_fwd_params, dt_rank),
         offsetof(oss_scan_fwd_params, dt_rank_stride), offsetof(oss_scan_bwd_params, ddt), offsetof(oss_scan_bwd_params, ddt_weight),
         offsetof(oss_scan_bwd_params, ddt_rank_stride));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(oss_scan_fwd_params), offsetof(oss_scan_fwd_params, u_batch_stride),
         offsetof(oss_scan_fwd_params, u), offsetof(oss_scan_fwd_params, x), sizeof(oss_scan_bwd_params),
         offsetof(oss_scan_bwd_params, dout_batch_stride), offsetof(oss_scan_bwd_params, dout),
         offsetof(oss_scan_bwd_params, workspace_bytes), offsetof(oss_scan_bwd_params, dBC_group_stride),
         sizeof(oss_chan_params), offsetof(oss_chan_params, pooled), offsetof(oss_chan_params, zt), offsetof(oss_chan_params, c));
  return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "probe.c")
        open(c, "w").write(probe)
        exe = os.path.join(td, "probe")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    F, B, Ch = _capi.ScanFwdParams, _capi.ScanBwdParams, _capi.ChanParams
    want = [F.workspace.offset, F.workspace_bytes.offset, F.dt_weight.offset, F.dt_rank.offset, F.dt_rank_stride.offset, B.ddt.offset, B.ddt_weight.offset, B.ddt_rank_stride.offset,
            ctypes.sizeof(F), F.u_batch_stride.offset, F.u.offset, F.x.offset, ctypes.sizeof(B),
            B.dout_batch_stride.offset, B.dout.offset, B.workspace_bytes.offset, B.dBC_group_stride.offset,
            ctypes.sizeof(Ch), Ch.pooled.offset, Ch.zt.offset, Ch.c.offset]
    assert got == want


def test_workspace_query_is_pure():
    lib = _capi.load()
    n = lib.oss_scan_bwd_workspace_bytes(2, 8, 100, 16, 4)
    tiles = (2 + 3) // 4  # 2 rows per group, 4 rows per workgroup in the smallest variant (bwd variant 1)
    # per row tile: dB / dC partial rows (2 * dstate) + 8 rows for the fused-delta form; per (batch, row): dA, dD, dbias, 8 dt weights
    assert n == 4 * (2 * 4 * tiles * (2 * 16 + 8) * 100 + 2 * 8 * (18 + 8))
    assert lib.oss_scan_bwd_workspace_bytes(2, 7, 100, 16, 4) == 0  # dim % n_groups != 0
    # long sequences: room for time-segmented launches -- one weight-gradient partial per (batch, segment, row) and the
    # reverse-carry pairs; segments are counted in 512-step chunks, at most 64
    n = lib.oss_scan_bwd_workspace_bytes(1, 8, 2048, 16, 4)
    assert n == 4 * (1 * 4 * 1 * (2 * 16 + 8) * 2048 + 1 * 4 * 8 * (18 + 8) + 2 * 1 * 8 * 16 * 4)
    # forward: (prod a, h) per (batch, row, segment, state); segments counted in 256-step chunks, at most 64; none needed below
    assert lib.oss_scan_fwd_workspace_bytes(2, 8, 256, 16, 4) == 0
    assert lib.oss_scan_fwd_workspace_bytes(2, 8, 1000, 16, 4) == 4 * 2 * 2 * 8 * 16 * 4
    assert lib.oss_scan_fwd_workspace_bytes(1, 384, 160 * 160, 16, 4) == 4 * 2 * 384 * 16 * 64


def test_segment_override_round_trips_without_a_gpu():
    lib = _capi.load()
    lib.oss_scan_set_segments(3, 5)
    lib.oss_scan_set_segments(-1, -1)
    assert lib.oss_scan_last_segments(0) >= 1 and lib.oss_scan_last_segments(1) >= 1


def test_cpu_tensors_are_rejected_not_silently_computed():
    """The product has no CPU path: the op only has a GPU kernel (cf. TORCH_CHECK(u.is_cuda()),
    cus/selective_scan.cpp:174)."""
    u = torch.zeros(1, 4, 8)
    with pytest.raises(RuntimeError, match="CUDA/HIP tensor"):
        vmambair_amd.selective_scan_fwd(u, u, torch.zeros(4, 2), torch.zeros(1, 1, 2, 8), torch.zeros(1, 1, 2, 8),
                                        None, None, True, 1)
    with pytest.raises(RuntimeError):
        vmambair_amd.selective_scan_fwd(u.double(), u.double(), torch.zeros(4, 2), torch.zeros(1, 1, 2, 8),
                                        torch.zeros(1, 1, 2, 8), None, None, True, 1)


def test_drop_in_module_surface():
    import selective_scan_cuda_core as m
    assert callable(m.fwd) and callable(m.bwd)


def test_compiled_torch_boundary_is_built_and_registers_its_ops():
    """lib/libvmambair_torch.so (csrc_host/oss_torch_host.cpp) loads without a GPU and defines the two operators with the
    mutated-argument annotation; no compute call here"""
    from vmambair_amd import _host
    assert os.path.exists(_build.HOST_LIB), "run __graft_entry__.build() first"
    assert _host.mode() == "c++"
    ops = _host.ops()
    s = str(ops.scan_bwd.default._schema)
    assert "Tensor(a!)? dbc_into" in s and str(ops.scan_fwd.default._schema).endswith("-> Tensor[]")
    _host.use("ctypes")
    try:
        assert _host.mode() == "ctypes" and _host.ops() is None
    finally:
        _host.use(None)


def test_shape_rules_of_the_fused_kernels_are_pure_host_queries():
    """which shapes the round-3 fused forms take (no GPU needed: the rules live in the library, the Python layer only asks)"""
    lib = _capi.load()
    BF16, F16, F32 = 2, 1, 0
    # depth-wise conv + silu (1 plane per workgroup) / + gelu gate (2 planes): 16-bit, W % 8 == 0, planes in LDS
    assert lib.oss_dwconv3x3_fused_ok(BF16, 64, 64, 1) == 1 and lib.oss_dwconv3x3_fused_ok(F16, 128, 128, 2) == 1
    # (round 4) float I/O too -- planes of twice the bytes: 64 x 64 gate fits, 160 x 160 gate (207 KiB) does not
    assert lib.oss_dwconv3x3_fused_ok(F32, 64, 64, 1) == 1 and lib.oss_dwconv3x3_fused_ok(F32, 64, 64, 2) == 1
    assert lib.oss_dwconv3x3_fused_ok(F32, 160, 160, 1) == 1 and lib.oss_dwconv3x3_fused_ok(F32, 160, 160, 2) == 0
    assert lib.oss_dwconv3x3_flat2_ok(F32, 64, 64) == 1 and lib.oss_dwconv3x3_flat2_ok(BF16, 12, 64) == 0 and lib.oss_dwconv3x3_flat2_ok(BF16, 16, 160) == 0
    # (round 4) rows whose W / 8 lane groups straddle waves are taken too (EDGE instantiations): RealSR's 160-wide tiles, W = 24
    assert lib.oss_dwconv3x3_fused_ok(F16, 160, 160, 1) == 1 and lib.oss_dwconv3x3_fused_ok(F16, 160, 160, 2) == 1
    assert lib.oss_dwconv3x3_fused_ok(BF16, 16, 24, 2) == 1 and lib.oss_dwconv3x3_fused_ok(BF16, 16, 20, 2) == 0
    assert lib.oss_dwconv3x3_fused_ok(BF16, 256, 256, 1) == 1 and lib.oss_dwconv3x3_fused_ok(BF16, 256, 256, 2) == 0   # 129 / 258 KiB
    assert lib.oss_dwconv3x3_fused_ok(BF16, 64, 64, 3) == 0
    # LayerNorm inside the 1x1 convolution: cin % 16 == 0, cin <= 192, pixels % 128 == 0
    assert lib.oss_ln_conv1x1_ok(BF16, 192, 96, 4096) == 1 and lib.oss_ln_conv1x1_ok(F16, 510, 96, 25600) == 1
    assert lib.oss_ln_conv1x1_ok(BF16, 768, 384, 1024) == 0         # level 4 of the UNet: K = 384
    assert lib.oss_ln_conv1x1_ok(BF16, 192, 96, 4000) == 0 and lib.oss_ln_conv1x1_ok(F32, 192, 96, 4096) == 0
    assert lib.oss_ln_conv1x1_ok(BF16, 254, 127, 4096) == 0
    # input gradient + LayerNorm backward: cin <= 128, 2 cin <= cout <= 192
    assert lib.oss_conv1x1_dgrad_ln_bwd_ok(BF16, 192, 96, 4096, 8) == 1 and lib.oss_conv1x1_dgrad_ln_bwd_ok(F16, 96, 48, 1024, 1) == 1
    assert lib.oss_conv1x1_dgrad_ln_bwd_ok(BF16, 510, 96, 4096, 8) == 0    # project_in: the wave-level kernel + its own LayerNorm launch
    assert lib.oss_conv1x1_dgrad_ln_bwd_ok(BF16, 96, 96, 4096, 8) == 0     # fewer than 2 cin rows to park x and the skip gradient in
    assert lib.oss_conv1x1_dgrad_ln_bwd_ok(BF16, 384, 192, 1024, 8) == 0
    n = lib.oss_conv1x1_dgrad_ln_bwd_partial_floats(8, 96, 4096)
    assert n == 8 * (4096 // 64) * 2 * 96
    assert lib.oss_conv1x1_dgrad_ln_bwd_partial_floats(0, 96, 4096) == 0
