"""GPU: randomised ragged-shape stress of the bf16-piece kernels (tools/stress_split.py): bit-identical repeats,
NaN-poisoned tile padding, agreement with the exact-f32 MFMA kernels / fp64 to fp32 rounding."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_split_kernel_stress(seed):
    env = dict(os.environ)
    for k in ("MMDFN_PROP_CFG", "MMDFN_TILEDOT_SPLIT", "MMDFN_LIN_CFG"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_split.py"), "30", str(seed)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "stress ok: 30 cases" in out.stdout
