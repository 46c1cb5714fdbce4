"""-m gpu: run-to-run bitwise determinism of the MFMA kernels (tools/determinism_stress.py, 30 launches each): an intermittent
operand-register hazard (see coupling_step.hip) produces bf16-sized faults in a fraction of the launches, which one parity run can miss."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mfma_kernels_are_run_to_run_deterministic():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_stress.py"), "30"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TOTAL differing launches: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
