"""The C-ABI library loads and exports every symbol include/mmdfn_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmdfn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(mmdfn_\w+)\s*\(", text)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "mmdfn_propagate" in syms and "mmdfn_adj_build" in syms and len(syms) >= 5


def test_library_exports_every_declared_symbol():
    from mm_dfn_amd import build
    if not os.path.exists(build.LIBPATH):
        pytest.skip("libmmdfn_hip.so not built here (hipcc unavailable?)")
    handle = ctypes.CDLL(build.LIBPATH)
    for s in declared_symbols():
        assert hasattr(handle, s), "missing export %s" % s
    from mm_dfn_amd import _hip
    assert handle.mmdfn_abi_version() == _hip.ABI_VERSION


def test_binding_table_matches_header():
    from mm_dfn_amd import _hip
    assert sorted(_hip.SIGNATURES.keys()) == declared_symbols()


def test_arg_counts_match_header():
    from mm_dfn_amd import _hip
    text = open(os.path.join(ROOT, "include", "mmdfn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in re.findall(r"\bint\s+(mmdfn_\w+)\s*\(([^)]*)\)", text):
        n = 0 if args.strip() in ("", "void") else len(args.split(","))
        assert n == len(_hip.SIGNATURES[name]), name


def test_cpu_tensors_fail_loudly():
    import torch
    from mm_dfn_amd import ops, _hip
    from mm_dfn_amd.layout import DialogueLayout
    lay = DialogueLayout([3, 2], 3, "cpu")
    with pytest.raises(_hip.HipLibraryError):
        ops.build_adjacency(torch.randn(3, 5, 8), [3, 2])
