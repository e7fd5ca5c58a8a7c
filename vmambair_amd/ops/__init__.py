"""Torch-facing boundary of the HIP library: ``torch.ops.vmambair.*`` and the autograd nodes built on them.

One module per operator family (round 1 had a single 1 000-line ``ops.py``):
  ``scan``      selective_scan_fwd / _bwd -- the drop-in for the reference's ``selective_scan_cuda_core`` (cus/selective_scan.cpp:157-349),
                its omni form and the four-direction merge
  ``core``      the spatial branch of SS2D_1 as one node (flattenings + projections + scan + merge)
  ``channel``   the channel branch + gate as one node
  ``pointwise`` in_conv / out_conv / project_in / project_out (1x1 convolutions) on the matrix cores
  ``conv3x3``   dense 3x3 convolutions with <= 4 channels on one side (patch_embed, the tail's last layer)
  ``dwconv``    depth-wise 3x3 (+ silu)          ``layernorm``  NCHW LayerNorm (+ gate)          ``ffn``  gelu gate of the EFFN
  ``_common``   dtype table, checks, deferred finishing, weight-gradient side stream
Everything is re-exported here, so ``from vmambair_amd import ops; ops.selective_scan_fwd(...)`` keeps working.  Module-level
switches live in their own module (``ops.core.FUSED_DT``, ``ops.pointwise.CONV1X1_IMPL``, ``ops._common._DEFER_KEEP``).

No CPU implementation exists: CPU tensors are rejected exactly as the reference rejects them (``TORCH_CHECK(u.is_cuda())``,
cus/selective_scan.cpp:174).
"""
from .. import _capi  # noqa: F401
from . import _common, channel, conv3x3, core, dwconv, ffn, layernorm, pointwise, scan  # noqa: F401
from ._common import (WGRAD_STATS, WgradTable, flush_wgrads, pending_wgrad_table_bytes, pending_wgrads, _keep_operands,  # noqa: F401
                      FinishTable, _DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr,  # noqa: F401
                      deferred_finishes, flush_finishes, orphaned_deferred_outputs, pending_finish_chunks, scan_chunk,
                      PairGrad, split_halves, wgrad_side_stream)
from .channel import ChannelGateFn, NormChannelGateFn, chan_gate_bwd, chan_gate_fwd, chan_supported  # noqa: F401
from .pointwise import (Conv1x1Fn, LNConv1x1Fn, conv1x1, conv1x1_bwd, conv1x1_fwd, ln_conv1x1, ln_conv1x1_fwd,  # noqa: F401
                        ln_conv1x1_ok)
from .core import (ConvCoreFn, SS2DCoreFn, conv_core_ok, core_supported, cross_merge2, cross_scan2, fused_dt_supported, proj_dgrad, proj_fwd,  # noqa: F401
                   proj_set_path, proj_wgrad, ss2d_conv_core_bwd, ss2d_conv_core_fwd, ss2d_core_bwd, ss2d_core_fwd)
from .dwconv import (DWConv3x3Fn, DWGateFn, dwconv3x3, dwconv3x3_bwd, dwconv3x3_fwd, dwconv3x3_gelu_gate,  # noqa: F401
                     dwconv3x3_silu_bwd, dwconv3x3_silu_flat2_bwd, dwconv3x3_silu_flat2_fwd, dwconv3x3_silu_fwd, dwgate_bwd, dwgate_fwd, flat2_ok)
from .conv3x3 import ThinConv3x3Fn, conv3x3_thin_bwd, conv3x3_thin_fwd  # noqa: F401
from .conv3x3 import conv3x3 as conv3x3_layer  # noqa: F401
from .ffn import GeluGateFn, effn_fwd, effn_fwd_ok, effn_round_weights, gelu_gate, gelu_gate_bwd, gelu_gate_fwd  # noqa: F401
from .layernorm import _CODE_DT, _DT_CODE, LayerNormNCHWFn, layer_norm_nchw, ln_nchw_bwd, ln_nchw_fwd  # noqa: F401
from .scan import merge4, selective_scan_bwd, selective_scan_fwd  # noqa: F401
