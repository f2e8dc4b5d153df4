"""ctypes binding of the producer / consumer K6 experiment (tools/k6_pc/propagate_pc.hip; `bash tools/k6_pc/build.sh`)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402
from mm_dfn_amd import _hip  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "bin", "libk6pc.so")
        if not os.path.exists(path):
            raise RuntimeError("run `bash tools/k6_pc/build.sh` first")
        _lib = ctypes.CDLL(path)
        _lib.k6pc_propagate.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        _lib.k6pc_propagate.restype = ctypes.c_int
    return _lib


def propagate_pc(tiles, cross, H, lay, out=None):
    """out = A . H on the producer / consumer kernel (raises when the shape is not covered)."""
    d = H.shape[1]
    if out is None:
        out = torch.empty(H.shape[0], d, dtype=torch.float32, device=H.device)
    rc = lib().k6pc_propagate(_hip.ptr(tiles), _hip.ptr(cross), _hip.ptr(H), _hip.ptr(out), _hip.ptr(lay.dia_len),
                              _hip.ptr(lay.row_start), _hip.ptr(lay.tile_base), lay.B, lay.M, lay.N, d, H.stride(0),
                              out.stride(0), lay.max_len, _hip.stream())
    if rc != 0:
        raise RuntimeError("k6pc_propagate returned %d" % rc)
    return out
