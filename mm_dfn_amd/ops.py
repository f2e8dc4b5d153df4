"""autograd.Functions over the C-ABI kernels (libmmdfn_hip.so) -- the FACADE: the operators live in `ops_*.py`, one module per
group (round 6 split of a 2 000-line file); every name, private ones included, is re-exported here, so `ops.X` keeps working.
Module-level state that is REBOUND (not mutated) must be patched in its own module: `ops_wgrad._prepare_wgrad_batch`,
`ops_flags.draw_flags`, `ops_flags._FLAG_SCOPE`.

Every function launches hand-written gfx950 kernels on the current HIP stream.  There is no CPU / eager fallback: CPU
tensors raise.
"""
from . import _hip  # noqa: F401
from .layout import BlockTileAdjacency, DialogueLayout  # noqa: F401
from .ops_pad import (  # noqa: F401
    _lay_args, _rows_view, _PAD_REG, pad4, FEATURE_SLOTS, is_odd_feature_tensor, register_row_padded, padded_zeros,
    pad_rows, row_padded_view, row_operand, ensure_row_padded, padded_grad_like, weight_operand,
)
from .ops_graph import (  # noqa: F401
    propagate_raw, tile_outer_raw, _Propagate, propagate, _BuildAdjacency, build_adjacency, _PropagateConcat,
    propagate_concat, _LstmPointwise, lstm_pointwise, _GcniiCombine, gcnii_combine,
)
from .ops_wgrad import (  # noqa: F401
    colsum, _strided_rows, gemm_tn_supported, gemm_tn, gemm_tn_grouped, _WGQ, EARLY_WGRAD, _WG_MAX, wgrad_batch,
    wgrad_batching, _leaf, _hooked, _queueable, queue_wgrad, SLAB_RIDE, slab_reduce_queueable, queue_slab_reduce,
    _join_side, _GRAPH_DONE_HOOK, set_graph_backward_done_hook, flush_queued_wgrads_now, flush_queued_wgrads_early,
    _GRAD_ADDENDS, _GRAD_ADDENDS_ARMED, drop_grad_addends, add_grad_addends, apply_grad_addends, flush_queued_wgrads,
    _ext_destinations, _flush_outs, _launch_wgrad_batch, _prepare_wgrad_batch, join_weight_grads,
    set_async_weight_grads, _wgrad_inline, _wgrad, stage_riders, finish_riders,
)
from .ops_linear import (  # noqa: F401
    linear_raw, linear_group_raw, linear_group_supported, dense_nk, dense_kn, linear_supported, linear_preferred,
    _Linear, _MatmulKN, _linear_forward, _linear_dx, _LinearGroup, linear_group, _GateLinear, gate_linear, _Linear2,
    PLANES_MIN_ROWS, _PLANES, _PLANE_EPOCH, _PLANE_RECORDERS, _PlaneEntry, _plane_stamp, _plane_params, _plane_fresh,
    _cut_planes, planes_supported, weight_planes, refresh_planes, invalidate_planes, planes_recording,
    linear_planes_raw, linear_planes_group_raw, linear2_group, LINEAR2_FEW_ROWS, GROUP_ROWS, linear2, matmul_kn, linear,
)
from .ops_party import (  # noqa: F401
    _PartyGather, party_gather, _HalvesGrad, _queue_bias_halves, _ProjectGather, project_gather, _PartyCombine,
    party_combine,
)
from .ops_fusion import (  # noqa: F401
    _SoftmaxScale, softmax_scale, _MfnMem, mfn_mem, _GatedPair, gated_pair,
)
from .ops_flags import (  # noqa: F401
    _FLAG_SCOPE, _FLAG_HINT, flag_pool, keep_scale, _FLAG_STATE, _FLAG_CONSUMED, flags_consumed, flag_state_snapshot,
    flag_state_restore, flag_state_sync, flags_advance_host, draw_flags, keep_flags, _MaskScale, mask_scale,
    stage_flag_draw, finish_flag_draw,
)
from .ops_head import (  # noqa: F401
    _Head, _head_width, head_supported, head,
)
