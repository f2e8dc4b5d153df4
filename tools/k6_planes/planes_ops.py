"""ctypes binding of the experiment library (tools/k6_planes/bin/libk6planes.so) next to mm_dfn_amd.ops."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from mm_dfn_amd import _hip, ops

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "bin", "libk6planes.so")
        if not os.path.exists(path):
            subprocess.check_call(["bash", os.path.join(HERE, "build.sh")])
        _lib = ctypes.CDLL(path)
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        _lib.mmdfn_cut_planes.argtypes = [P, P, L, I, I, P]
        _lib.mmdfn_propagate_planes.argtypes = [P] * 8 + [I] * 7 + [P]
    return _lib


def planes_rows(R):
    return (R + 31) // 32 * 32 + 32


def cut_planes(X, out=None):
    X = ops._rows_view(X, X.shape[0])
    R, d = X.shape
    dp = (d + 7) // 8 * 8
    if out is None:
        out = torch.empty(3, planes_rows(R), dp, dtype=torch.int16, device=X.device)
    _hip.check(lib().mmdfn_cut_planes(_hip.ptr(X), _hip.ptr(out), R, d, X.stride(0), _hip.stream()), "cut_planes")
    return out


def propagate_planes_raw(tiles, cross, H, planes, lay, out=None):
    H = ops._rows_view(H, lay.M * lay.N)
    d = H.shape[1]
    if out is None:
        out = torch.empty(H.shape[0], d, dtype=torch.float32, device=H.device)
    rc = lib().mmdfn_propagate_planes(_hip.ptr(tiles), _hip.ptr(cross), _hip.ptr(H), _hip.ptr(planes), _hip.ptr(out),
                                      *ops._lay_args(lay), lay.B, lay.M, lay.N, d, H.stride(0), out.stride(0), lay.max_len,
                                      _hip.stream())
    _hip.check(rc, "propagate_planes")
    return out
