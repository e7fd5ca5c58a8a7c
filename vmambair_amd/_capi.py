"""ctypes binding of include/vmambair_oss.h (the C ABI of the HIP library).

The structures mirror ``oss_scan_fwd_params`` / ``oss_scan_bwd_params`` field for field.  The
library is loaded on first use; a missing library is a hard error (no fallback path exists).
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

OSS_F32, OSS_F16, OSS_BF16 = 0, 1, 2
FEATURE_FUSED_DT, FEATURE_LANE_STATES = 1, 2   # oss_scan_features(): runtime-selected scan forms (in every library since round 6)

ERRORS = {
    -1: "OSS_ERR_NULL: a required pointer is NULL",
    -2: "OSS_ERR_SHAPE: invalid batch/dim/seqlen/dstate/n_groups",
    -3: "OSS_ERR_DSTATE: selective_scan only supports state dimension <= 256",
    -4: "OSS_ERR_WORKSPACE: workspace missing or too small",
}


class ScanFwdParams(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("dim", C.c_int), ("seqlen", C.c_int), ("dstate", C.c_int), ("n_groups", C.c_int),
        ("delta_softplus", C.c_int), ("rev_group_start", C.c_int), ("u_row_mod", C.c_int),
        ("a_log_form", C.c_int), ("reserved0_", C.c_int),
        ("u_batch_stride", C.c_int64), ("u_d_stride", C.c_int64),
        ("delta_batch_stride", C.c_int64), ("delta_d_stride", C.c_int64),
        ("out_batch_stride", C.c_int64), ("out_d_stride", C.c_int64),
        ("A_d_stride", C.c_int64),
        ("B_batch_stride", C.c_int64), ("B_group_stride", C.c_int64), ("B_dstate_stride", C.c_int64),
        ("C_batch_stride", C.c_int64), ("C_group_stride", C.c_int64), ("C_dstate_stride", C.c_int64),
        ("u", C.c_void_p), ("delta", C.c_void_p), ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("D", C.c_void_p), ("delta_bias", C.c_void_p), ("out", C.c_void_p), ("x", C.c_void_p),
        ("dt_weight", C.c_void_p), ("dt_rank", C.c_int), ("reserved1_", C.c_int),
        ("dt_group_stride", C.c_int64), ("dt_rank_stride", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("hs", C.c_void_p),
        # per-call launch tuning (ABI 7): 0 = heuristic; variant + 1 / segments / carry pieces (include/vmambair_oss.h)
        ("tune_variant", C.c_int), ("tune_segments", C.c_int), ("tune_carry_split", C.c_int), ("reserved2_", C.c_int),
    ]


class ScanBwdParams(C.Structure):
    _fields_ = [
        ("f", ScanFwdParams),
        ("dout_batch_stride", C.c_int64), ("dout_d_stride", C.c_int64),
        ("du_batch_stride", C.c_int64), ("du_d_stride", C.c_int64),
        ("ddelta_batch_stride", C.c_int64), ("ddelta_d_stride", C.c_int64),
        ("dout", C.c_void_p), ("du", C.c_void_p), ("ddelta", C.c_void_p), ("dA", C.c_void_p),
        ("dB", C.c_void_p), ("dC", C.c_void_p), ("dD", C.c_void_p), ("ddelta_bias", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dout_row_mod", C.c_int), ("reserved_", C.c_int),
        ("dBC_group_stride", C.c_int64),
        ("ddt", C.c_void_p), ("ddt_weight", C.c_void_p),
        ("ddt_batch_stride", C.c_int64), ("ddt_group_stride", C.c_int64), ("ddt_rank_stride", C.c_int64),
        ("tune_variant", C.c_int), ("tune_segments", C.c_int), ("tune_partials", C.c_int), ("reserved3_", C.c_int),
        ("finish_dt_weight", C.c_void_p), ("finish_dt_rank", C.c_int), ("reserved4_", C.c_int),
    ]


class ChanParams(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("dc", C.c_int), ("Rc", C.c_int), ("Cc", C.c_int), ("reserved_", C.c_int)] + \
        [(n, C.c_void_p) for n in ("pooled", "cin_w", "cin_b", "Wxc", "Wdtc", "dt_bias", "A_logs", "Dsc", "cout_w", "cout_b",
                                    "cn_w", "cn_b", "zt", "dts", "hs", "y", "yc", "stat", "c", "pool_part")] + \
        [("n_part", C.c_int), ("pool_scale", C.c_float)]


class AdamChunk(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("ema", C.c_void_p), ("n", C.c_int), ("reserved_", C.c_int)]


ADAM_CHUNK = 2048
SUM_CHUNK_BYTES = 40   # sizeof(oss_sum_chunk)

#: every symbol include/vmambair_oss.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = ["oss_scan_chunk", "oss_scan_num_chunks", "oss_scan_fwd", "oss_scan_fwd_workspace_bytes", "oss_scan_lane_state_floats", "oss_scan_bwd_workspace_bytes",
           "oss_scan_bwd", "oss_scan_bwd_finish_dt_ok", "oss_scan_fused_dt_ok", "oss_scan_set_variant", "oss_scan_last_variant", "oss_scan_set_segments", "oss_scan_set_carry_split",
           "oss_scan_last_segments", "oss_scan_last_lane_states", "oss_prof_enable", "oss_prof_reset",
           "oss_prof_collect", "oss_prof_collect2", "oss_prof_family_enable", "oss_prof_family_count", "oss_prof_family", "oss_dwconv3x3_fwd", "oss_dwconv3x3_wgrad", "oss_dwconv3x3_fused_ok", "oss_dwconv3x3_silu_fwd", "oss_dwconv3x3_silu_bwd", "oss_dwconv3x3_flat2_ok", "oss_dwconv3x3_silu_flat2_fwd",
           "oss_dwconv3x3_silu_flat2_bwd",
           "oss_dwgate_fwd_ok", "oss_effn_fwd_ok", "oss_effn_round_weights", "oss_effn_fwd", "oss_dwgate_fwd", "oss_dwgate_bwd", "oss_ln_nchw_fwd", "oss_ln_nchw_fwd_pool", "oss_ln_nchw_fwd_pool_tiles", "oss_ln_nchw_bwd", "oss_ln_nchw_bwd_affine", "oss_ln_nchw_bwd_partial_floats", "oss_merge4", "oss_conv1x1_fwd", "oss_conv1x1_dgrad",
           "oss_conv1x1_wgrad_partial_floats", "oss_conv1x1_wgrad", "oss_conv1x1_wgrad_set_tile", "oss_conv1x1_wgrad_set_span", "oss_conv1x1_wg", "oss_conv1x1_set_wg", "oss_ln_conv1x1_ok", "oss_ln_conv1x1_fwd", "oss_conv1x1_dgrad_ln_bwd_ok",
           "oss_conv1x1_dgrad_ln_bwd_partial_floats", "oss_conv1x1_dgrad_ln_bwd", "oss_cross_scan2", "oss_cross_merge2", "oss_proj_fwd",
           "oss_proj_dgrad", "oss_proj_wgrad_partial_floats", "oss_proj_wgrad", "oss_proj_set_path", "oss_proj_rows_optional_ok", "oss_chan_fwd", "oss_chan_grad_floats",
           "oss_chan_bwd_scratch_floats", "oss_chan_bwd", "oss_rowsum", "oss_row_affine", "oss_gelu_gate_fwd",
           "oss_gelu_gate_bwd", "oss_adam_ema_step", "oss_adamw_ema_step", "oss_set_defer_finish", "oss_deferred_chunks",
           "oss_flush_finishes", "oss_flush_finishes_n", "oss_flush_wgrads_n", "oss_set_defer_wgrad", "oss_deferred_wgrads", "oss_deferred_wgrad_table_bytes", "oss_flush_wgrads",
           "oss_conv3x3_thin_ok", "oss_conv3x3_thin_fwd", "oss_conv3x3_thin_dgrad", "oss_conv3x3_thin_wgrad_partial_floats",
           "oss_conv3x3_thin_wgrad", "oss_hbm_copy", "oss_prof_marker", "oss_scan_build_id", "oss_version", "oss_scan_features", "oss_abi_version", "oss_abi_struct_bytes"]

#: include/vmambair_oss.h: OSS_ABI_VERSION this binding was written against
ABI_VERSION = 7

_lib = None


def lib_path() -> str:
    # VMAMBAIR_LIB: timing experiments only (tools/build_experiment.sh)
    return os.environ.get("VMAMBAIR_LIB") or _build.LIB_PATH


def load():
    """dlopen the in-tree library; raise loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = C.CDLL(path)
    lib.oss_scan_chunk.restype = C.c_int
    lib.oss_scan_num_chunks.restype = C.c_int
    lib.oss_scan_num_chunks.argtypes = [C.c_int]
    lib.oss_scan_fwd.restype = C.c_int
    lib.oss_scan_fwd.argtypes = [C.POINTER(ScanFwdParams), C.c_int, C.c_void_p]
    lib.oss_scan_bwd_workspace_bytes.restype = C.c_size_t
    lib.oss_scan_bwd_workspace_bytes.argtypes = [C.c_int] * 5
    lib.oss_scan_fwd_workspace_bytes.restype = C.c_size_t
    lib.oss_scan_fwd_workspace_bytes.argtypes = [C.c_int] * 5
    lib.oss_scan_last_lane_states.restype = C.c_int
    lib.oss_scan_lane_state_floats.restype = C.c_size_t
    lib.oss_scan_lane_state_floats.argtypes = [C.c_int] * 4
    lib.oss_scan_set_segments.restype = None
    lib.oss_scan_set_segments.argtypes = [C.c_int, C.c_int]
    lib.oss_scan_set_carry_split.restype = None
    lib.oss_scan_set_carry_split.argtypes = [C.c_int]
    lib.oss_scan_last_segments.restype = C.c_int
    lib.oss_scan_last_segments.argtypes = [C.c_int]
    lib.oss_scan_bwd.restype = C.c_int
    lib.oss_scan_bwd.argtypes = [C.POINTER(ScanBwdParams), C.c_int, C.c_void_p]
    lib.oss_proj_rows_optional_ok.restype = C.c_int
    lib.oss_proj_rows_optional_ok.argtypes = [C.c_int] * 6
    lib.oss_scan_bwd_finish_dt_ok.restype = C.c_int
    lib.oss_scan_bwd_finish_dt_ok.argtypes = [C.c_int, C.c_int]
    lib.oss_scan_fused_dt_ok.restype = C.c_int
    lib.oss_scan_fused_dt_ok.argtypes = [C.c_int] * 7
    lib.oss_scan_set_variant.restype = None
    lib.oss_scan_set_variant.argtypes = [C.c_int, C.c_int]
    lib.oss_scan_last_variant.restype = C.c_int
    lib.oss_scan_last_variant.argtypes = [C.c_int]
    lib.oss_prof_enable.restype = None
    lib.oss_prof_enable.argtypes = [C.c_int]
    lib.oss_prof_reset.restype = None
    lib.oss_prof_collect.restype = C.c_int
    lib.oss_prof_collect.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong),
                                     C.POINTER(C.c_double)]
    lib.oss_prof_family_enable.argtypes = [C.c_int]
    lib.oss_prof_family_enable.restype = None
    lib.oss_prof_family_count.restype = C.c_int
    lib.oss_prof_family.restype = C.c_int
    lib.oss_prof_family.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    lib.oss_prof_collect2.restype = C.c_int
    lib.oss_prof_collect2.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oss_dwconv3x3_fwd.restype = C.c_int
    lib.oss_dwconv3x3_fwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + \
        [C.c_int64] * 4 + [C.c_int, C.c_void_p]
    lib.oss_dwconv3x3_wgrad.restype = C.c_int
    lib.oss_dwconv3x3_wgrad.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_void_p]
    lib.oss_dwconv3x3_fused_ok.restype = C.c_int
    lib.oss_dwconv3x3_fused_ok.argtypes = [C.c_int] * 4
    lib.oss_dwgate_fwd_ok.restype = C.c_int
    lib.oss_dwgate_fwd_ok.argtypes = [C.c_int] * 3
    lib.oss_effn_fwd_ok.restype = C.c_int
    lib.oss_effn_fwd_ok.argtypes = [C.c_int] * 5
    lib.oss_effn_round_weights.restype = C.c_int
    lib.oss_effn_round_weights.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 2 + [C.c_void_p]
    lib.oss_effn_fwd.restype = C.c_int
    lib.oss_effn_fwd.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
    for fn in (lib.oss_dwconv3x3_silu_fwd, lib.oss_dwgate_fwd):
        fn.restype = C.c_int
        fn.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_void_p]
    for fn in (lib.oss_dwconv3x3_silu_bwd, lib.oss_dwgate_bwd):
        fn.restype = C.c_int
        fn.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_int64] * 6 + [C.c_void_p]
    lib.oss_dwconv3x3_flat2_ok.restype = C.c_int
    lib.oss_dwconv3x3_flat2_ok.argtypes = [C.c_int] * 3
    lib.oss_dwconv3x3_silu_flat2_fwd.restype = C.c_int
    lib.oss_dwconv3x3_silu_flat2_fwd.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_int64] * 2 + [C.c_void_p]
    lib.oss_dwconv3x3_silu_flat2_bwd.restype = C.c_int
    lib.oss_dwconv3x3_silu_flat2_bwd.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_void_p]
    lib.oss_ln_nchw_fwd.restype = C.c_int
    lib.oss_ln_nchw_fwd.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
    lib.oss_ln_nchw_fwd_pool_tiles.restype = C.c_int
    lib.oss_ln_nchw_fwd_pool_tiles.argtypes = [C.c_int, C.c_int] + [C.c_int64] * 4
    lib.oss_ln_nchw_fwd_pool.restype = C.c_int
    lib.oss_ln_nchw_fwd_pool.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
    lib.oss_ln_nchw_bwd_partial_floats.restype = C.c_size_t
    lib.oss_ln_nchw_bwd_partial_floats.argtypes = [C.c_int] * 3
    lib.oss_ln_nchw_bwd.restype = C.c_int
    lib.oss_ln_nchw_bwd.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 13 + [C.c_int] * 3 + [C.c_int64] * 5 + [C.c_void_p]
    lib.oss_ln_nchw_bwd_affine.restype = C.c_int
    lib.oss_ln_nchw_bwd_affine.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_float] + [C.c_void_p] * 8 + [C.c_int] * 3 +
                                           [C.c_int64] * 5 + [C.c_void_p])
    lib.oss_merge4.restype = C.c_int
    lib.oss_merge4.argtypes = [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    lib.oss_conv1x1_fwd.restype = C.c_int
    lib.oss_conv1x1_fwd.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_int64] * 2 + [C.c_void_p]
    lib.oss_conv1x1_dgrad.restype = C.c_int
    lib.oss_conv1x1_dgrad.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_int64] * 2 + [C.c_void_p]
    lib.oss_conv1x1_wgrad_partial_floats.restype = C.c_size_t
    lib.oss_conv1x1_wgrad_partial_floats.argtypes = [C.c_int] * 4
    lib.oss_conv1x1_wgrad.restype = C.c_int
    lib.oss_conv1x1_wgrad.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_void_p]
    lib.oss_cross_scan2.restype = C.c_int
    lib.oss_cross_scan2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_int64] * 2 + [C.c_void_p]
    lib.oss_cross_merge2.restype = C.c_int
    lib.oss_cross_merge2.argtypes = [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    lib.oss_proj_fwd.restype = C.c_int
    lib.oss_proj_fwd.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
    lib.oss_proj_dgrad.restype = C.c_int
    lib.oss_proj_dgrad.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]
    lib.oss_proj_wgrad_partial_floats.restype = C.c_size_t
    lib.oss_proj_wgrad_partial_floats.argtypes = [C.c_int] * 5
    lib.oss_conv1x1_wgrad_set_tile.restype = None
    lib.oss_conv1x1_wgrad_set_tile.argtypes = [C.c_int]
    lib.oss_conv1x1_wg.restype = C.c_int
    lib.oss_conv1x1_wg.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_int64] * 2 + [C.c_int, C.c_void_p]
    lib.oss_ln_conv1x1_ok.restype = C.c_int
    lib.oss_ln_conv1x1_ok.argtypes = [C.c_int] * 4
    lib.oss_ln_conv1x1_fwd.restype = C.c_int
    lib.oss_ln_conv1x1_fwd.argtypes = ([C.c_int] + [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_int64] * 2 +
                                       [C.c_void_p])
    lib.oss_conv1x1_dgrad_ln_bwd_ok.restype = C.c_int
    lib.oss_conv1x1_dgrad_ln_bwd_ok.argtypes = [C.c_int] * 5
    lib.oss_conv1x1_dgrad_ln_bwd_partial_floats.restype = C.c_size_t
    lib.oss_conv1x1_dgrad_ln_bwd_partial_floats.argtypes = [C.c_int] * 3
    lib.oss_conv1x1_dgrad_ln_bwd.restype = C.c_int
    lib.oss_conv1x1_dgrad_ln_bwd.argtypes = ([C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_int64] * 2 +
                                             [C.c_void_p])
    lib.oss_conv1x1_set_wg.restype = None
    lib.oss_conv1x1_set_wg.argtypes = [C.c_int, C.c_int]
    lib.oss_conv1x1_wgrad_set_span.restype = None
    lib.oss_conv1x1_wgrad_set_span.argtypes = [C.c_int]
    lib.oss_proj_set_path.restype = None
    lib.oss_proj_set_path.argtypes = [C.c_int]
    lib.oss_proj_wgrad.restype = C.c_int
    lib.oss_proj_wgrad.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]
    lib.oss_chan_fwd.restype = C.c_int
    lib.oss_chan_fwd.argtypes = [C.POINTER(ChanParams), C.c_void_p]
    lib.oss_chan_grad_floats.restype = C.c_size_t
    lib.oss_chan_grad_floats.argtypes = [C.c_int] * 4
    lib.oss_chan_bwd_scratch_floats.restype = C.c_size_t
    lib.oss_chan_bwd_scratch_floats.argtypes = [C.c_int] * 5
    lib.oss_chan_bwd.restype = C.c_int
    lib.oss_chan_bwd.argtypes = [C.POINTER(ChanParams)] + [C.c_void_p] * 5
    lib.oss_rowsum.restype = C.c_int
    lib.oss_rowsum.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
    lib.oss_row_affine.restype = C.c_int
    lib.oss_row_affine.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_int64] * 2 + [C.c_float, C.c_void_p]
    lib.oss_gelu_gate_fwd.restype = C.c_int
    lib.oss_gelu_gate_fwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int64, C.c_void_p]
    lib.oss_gelu_gate_bwd.restype = C.c_int
    lib.oss_gelu_gate_bwd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p]
    lib.oss_adam_ema_step.restype = C.c_int
    lib.oss_adam_ema_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_float] * 5 + [C.c_void_p]
    lib.oss_adamw_ema_step.restype = C.c_int
    lib.oss_adamw_ema_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]
    lib.oss_set_defer_finish.restype = None
    lib.oss_set_defer_finish.argtypes = [C.c_int]
    lib.oss_deferred_chunks.restype = C.c_size_t
    lib.oss_flush_finishes_n.restype = C.c_int
    lib.oss_flush_finishes_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.oss_flush_wgrads_n.restype = C.c_int
    lib.oss_flush_wgrads_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.oss_flush_finishes.restype = C.c_int
    lib.oss_flush_finishes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.oss_set_defer_wgrad.restype = None
    lib.oss_set_defer_wgrad.argtypes = [C.c_int]
    lib.oss_deferred_wgrads.restype = C.c_size_t
    lib.oss_deferred_wgrad_table_bytes.restype = C.c_size_t
    lib.oss_flush_wgrads.restype = C.c_int
    lib.oss_flush_wgrads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.oss_conv3x3_thin_ok.restype = C.c_int
    lib.oss_conv3x3_thin_ok.argtypes = [C.c_int] * 5
    lib.oss_conv3x3_thin_fwd.restype = C.c_int
    lib.oss_conv3x3_thin_fwd.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_int64] * 4 + [C.c_void_p]
    lib.oss_conv3x3_thin_dgrad.restype = C.c_int
    lib.oss_conv3x3_thin_dgrad.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_int64] * 4 + [C.c_void_p]
    lib.oss_conv3x3_thin_wgrad_partial_floats.restype = C.c_size_t
    lib.oss_conv3x3_thin_wgrad_partial_floats.argtypes = [C.c_int] * 3
    lib.oss_conv3x3_thin_wgrad.restype = C.c_int
    lib.oss_conv3x3_thin_wgrad.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_int64] * 4 + [C.c_void_p]
    lib.oss_hbm_copy.restype = C.c_int
    lib.oss_hbm_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.oss_prof_marker.restype = C.c_int
    lib.oss_prof_marker.argtypes = [C.c_int, C.c_void_p]
    lib.oss_scan_build_id.restype = C.c_char_p
    lib.oss_version.restype = C.c_char_p
    if not hasattr(lib, "oss_abi_version"):
        raise RuntimeError(f"{path} predates the ABI guard of include/vmambair_oss.h: rebuild it (__graft_entry__.build())")
    lib.oss_abi_version.restype = C.c_int
    lib.oss_scan_features.restype = C.c_int
    lib.oss_abi_struct_bytes.restype = C.c_size_t
    lib.oss_abi_struct_bytes.argtypes = [C.c_int]
    # the structs cross the boundary by pointer: a library built from another revision of the header would misread them
    mine = (ABI_VERSION, C.sizeof(ScanFwdParams), C.sizeof(ScanBwdParams), C.sizeof(ChanParams))
    theirs = (lib.oss_abi_version(), *(lib.oss_abi_struct_bytes(i) for i in range(3)))
    if mine != theirs:
        raise RuntimeError(f"{path}: ABI mismatch with vmambair_amd/_capi.py (version, sizeof fwd / bwd / chan params): "
                           f"library {theirs}, binding {mine}; rebuild with __graft_entry__.build()")
    _lib = lib
    return lib


def has_feature(bit: int) -> bool:
    """opt-in build features of the loaded library (include/vmambair_oss.h: oss_scan_features)"""
    return bool(load().oss_scan_features() & bit)


def require_feature(bit: int, what: str) -> None:
    if not has_feature(bit):
        name = {FEATURE_FUSED_DT: "fused_dt", FEATURE_LANE_STATES: "lane_states"}[bit]
        raise RuntimeError(f"{what}: {lib_path()} was built without the scan form '{name}' "
                           f"(-DOSS_WITHOUT_... in csrc/oss_host.h; the shipped build has both)")


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {ERRORS.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP error {rc}")
