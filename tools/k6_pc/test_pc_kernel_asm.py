"""The producer / consumer K6 kernel (tools/k6_pc/propagate_pc.hip, an experiment kept outside the product library) issues its producers' global loads from inline asm with
hand-counted waits; tools/k6_pc/verify_pc_asm.py checks on the ISA hipcc generates from its source that every wait
names the registers its load wrote and that nothing touches them in between (CPU test: hipcc cross-compiles without a
GPU)."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("defines", [(), ("MMDFN_TUNING",)], ids=["production", "tuning"])
def test_asm_loads_of_the_producer_consumer_kernel_are_retired_by_their_waits(defines):
    import verify_pc_asm
    path = verify_pc_asm.compile_listing(defines)
    try:
        assert verify_pc_asm.check_listing(path, verbose=False) == 0
    finally:
        os.unlink(path)


def test_the_verifier_catches_a_touched_register(tmp_path):
    import verify_pc_asm
    listing = """_ZN1_propagate_pc_kernelILi1EE:
.LBB0_1:                                ; =>This Loop Header: Depth=1
	s_waitcnt vmcnt(8) ; pc-wait v[10:13]
	v_add_u32_e32 v1, v10, v2
	global_load_dwordx4 v[10:13], v20, s[2:3] ; pc-load
	%s
	s_cbranch_scc1 .LBB0_1
.LBB0_2:
	s_endpgm
"""
    good = tmp_path / "good.s"
    good.write_text(listing % "v_add_u32_e32 v3, v4, v5")
    assert verify_pc_asm.check_listing(str(good), verbose=False) == 0
    bad = tmp_path / "bad.s"
    bad.write_text(listing % "v_mov_b32_e32 v30, v11")
    assert verify_pc_asm.check_listing(str(bad), verbose=False) > 0
