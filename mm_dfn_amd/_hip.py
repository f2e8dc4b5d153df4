"""ctypes binding of libmmdfn_hip.so (the C ABI declared in include/mmdfn_hip.h).

There is no CPU fallback: if the library is missing, stale or a kernel returns
an error, the product path raises.
"""
import ctypes
import os

import torch

from . import build as _build

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_int64

# name -> argtypes; must list every symbol of include/mmdfn_hip.h
SIGNATURES = {
    "mmdfn_abi_version": [],
    "mmdfn_propagate": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mmdfn_tile_outer": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mmdfn_adj_build": [_P] * 8 + [_P, _P, _P] + [_I] * 5 + [_F, _P],
    "mmdfn_adj_build_bwd": [_P] * 16 + [_P, _P, _P] + [_I] * 5 + [_F, _P],
    "mmdfn_gru_seq_fwd": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "mmdfn_gru_seq_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "mmdfn_gru_seq_fwd_seg": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "mmdfn_gru_seq_bwd_seg": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "mmdfn_gru_tab_reduce": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mmdfn_lstm_pointwise_fwd": [_P, _P, _P, _P, _L, _I, _P],
    "mmdfn_lstm_pointwise_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P],
    "mmdfn_gcnii_combine_fwd": [_P, _P, _P, _P, _P, _F, _F, _L, _I, _P],
    "mmdfn_gcnii_combine_bwd": [_P, _P, _P, _P, _P, _P, _F, _F, _L, _I, _P],
    "mmdfn_gcn_input_fwd": [_P] * 8 + [_I] * 4 + [_F, _P],
    "mmdfn_gcn_input_bwd": [_P] * 9 + [_I] * 4 + [_F, _P],
    "mmdfn_lstm_gate_fwd": [_P] * 10 + [_I] * 2 + [_P],
    "mmdfn_lstm_gate_fwd_ld": [_P] * 10 + [_I] * 3 + [_P],
    "mmdfn_lstm_gate_fwd_pre": [_P] * 10 + [_I] * 3 + [_P, _P],
    "mmdfn_lstm_gate_planes_workspace": [_I],
    "mmdfn_lstm_gate_cut_weights": [_P, _P, _P, _I, _P],
    "mmdfn_lstm_gate_takes_planes": [_I, _I],
    "mmdfn_lstm_gate_bwd": [_P] * 13 + [_I] * 4 + [_P],
    "mmdfn_gcnii_layer_fwd": [_P] * 7 + [_F, _F, _I, _I, _I, _F, _P],
    "mmdfn_prop_layer_fwd": [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _F, _P],
    "mmdfn_gcnii_layer_bwd": [_P] * 6 + [_F, _F, _I, _I, _I, _I, _P],
    "mmdfn_gcnii_layer_bwd_ld": [_P] * 6 + [_F, _F, _I, _I, _I, _I, _I, _P],
    "mmdfn_linear": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mmdfn_linear2": [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mmdfn_weight_planes_workspace": [_I, _I],
    "mmdfn_cut_weight_planes": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mmdfn_linear_planes": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mmdfn_linear_planes_group": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _F, _P],
    "mmdfn_linear_group_supported": [_I, _I, _I],
    "mmdfn_linear_group": [_I] + [_P] * 15 + [_I, _P],
    "mmdfn_linear_group_addend": [_I] + [_P] * 17 + [_I, _P],
    "mmdfn_softmax_scale_fwd": [_P, _P, _P, _P, _I, _I, _P],
    "mmdfn_softmax_scale_bwd": [_P, _P, _P, _P, _P, _I, _I, _P],
    "mmdfn_mfn_mem_fwd": [_P] * 6 + [_L, _P],
    "mmdfn_mfn_mem_bwd": [_P] * 7 + [_L, _P],
    "mmdfn_gated_pair_fwd": [_P] * 8 + [_I, _I, _I, _P],
    "mmdfn_gated_pair_bwd": [_P] * 12 + [_I, _I, _I, _P],
    "mmdfn_rowscale_colsum": [_P, _P, _P, _P, _I, _I, _P],
    "mmdfn_gemm_tn_splits": [_I, _I, _I],
    "mmdfn_gemm_tn": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mmdfn_gemm_tn_grouped_workspace": [_I, _P, _P, _P],
    "mmdfn_gemm_tn_grouped": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mmdfn_gemm_tn_batch_workspace": [_I, _P, _P, _I, _P, _P],
    "mmdfn_gemm_tn_batch": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mmdfn_wgrad_riders_stage": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mmdfn_wgrad_riders_staged": [],
    "mmdfn_wgrad_riders_flush": [_P],
    "mmdfn_wgrad_riders_drain": [_P, _I],
    "mmdfn_gru_seq_bwd_idle_cus": [_I, _P],
    "mmdfn_gru_seq_bwd_step_ns": [_I, _P],
    "mmdfn_gemm_tn_batch_ext": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P,
                                _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mmdfn_head_bwd_groups": [],
    "mmdfn_head_bwd_partial": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P],
    "mmdfn_colsum_partial": [_P, _L, _I, _I, _P, _P],
    "mmdfn_head_fwd": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _F, _P],
    "mmdfn_head_bwd_workspace": [_I, _I],
    "mmdfn_head_bwd": [_P] * 9 + [_L, _I, _I, _I, _I, _I, _F, _P],
    "mmdfn_focal_loss_fwd": [_P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "mmdfn_focal_loss_bwd": [_P, _P, _P, _P, _L, _I, _P],
    "mmdfn_focal_loss_fwd_grad": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "mmdfn_focal_loss_fwd_ignore": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _L, _P],
    "mmdfn_focal_loss_bwd_ignore": [_P, _P, _P, _P, _P, _L, _I, _L, _P],
    "mmdfn_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P],
    "mmdfn_party_gather": [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mmdfn_party_gather_bwd": [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mmdfn_party_gather_bwd_colsum": [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mmdfn_party_combine": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mmdfn_party_combine_bwd": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mmdfn_party_combine_bwd_dst": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mmdfn_mask_scale": [_I, _P, _P, _P, _P, _F, _P],
    "mmdfn_keep_flags": [_P, _L, _F, _P, _P],
    "mmdfn_keep_flags_stage": [_P, _L, ctypes.c_float, _P, _P],
    "mmdfn_keep_flags_flush": [_P],
    "mmdfn_gru_seq_fwd_takes_flags": [_I, _P],
    "mmdfn_colsum_workspace": [_I],
    "mmdfn_colsum": [_P, _L, _I, _I, _P, _P, _P],
}

ABI_VERSION = 17


class HipLibraryError(RuntimeError):
    pass


_libs = {}
_tuning = os.environ.get("MMDFN_TUNING_LIB", "0") == "1"


def set_tuning(flag):
    """Select the -DMMDFN_TUNING build (lib/libmmdfn_hip_tuning.so) for subsequent calls; returns the previous setting.
    Only tools/ and the variant-forcing kernel tests do this (also: MMDFN_TUNING_LIB=1 in the environment).  It is the
    one library that honours the MMDFN_* ablation / tiling switches; the production library has none compiled in."""
    global _tuning
    prev, _tuning = _tuning, bool(flag)
    return prev


def lib():
    """Load (once) and return the shared library; raises if unavailable."""
    handle = _libs.get(_tuning)
    if handle is not None:
        return handle
    tuning = _tuning
    path = _build.TUNING_LIBPATH if tuning else _build.LIBPATH
    if _build.is_stale(tuning):
        # missing or built from other sources (digest stamp): rebuild in place when a compiler is at hand (the GPU box
        # has hipcc).  A library that does not match the sources is never loaded.
        try:
            _build.build(verbose=False, tuning=tuning)
        except Exception as e:
            raise HipLibraryError(
                "%s is missing or stale (source digest mismatch) and could not be rebuilt (%s): run `python -m "
                "mm_dfn_amd.build` (needs hipcc); the MI355X path has no CPU fallback" % (path, e)) from e
    handle = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HipLibraryError("%s lacks symbol %s (stale build?)" % (os.path.basename(path), name)) from e
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name.endswith("_workspace") else ctypes.c_int
    if handle.mmdfn_abi_version() != ABI_VERSION:
        raise HipLibraryError("%s ABI version mismatch" % os.path.basename(path))
    _libs[tuning] = handle
    return handle


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise HipLibraryError("%s failed with code %d" % (what, rc))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipLibraryError("the MM-DFN HIP path needs tensors on an MI355X device (got %s); "
                                  "there is no CPU fallback" % t.device)


def ptr_array(tensors):
    """Host array of device pointers (one per group) for the grouped entry points."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def long_array(values):
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def float_array(values):
    return (ctypes.c_float * len(values))(*[float(v) for v in values])


def require_f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise HipLibraryError("the MM-DFN HIP kernels are fp32 (got %s)" % t.dtype)
