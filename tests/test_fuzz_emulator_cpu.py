"""CPU: a fixed seed of tools/fuzz_asm_emulator.py -- random launch geometries of the three generated kernel families through the lane-accurate emulator (keeps the campaign
tool runnable; the campaigns themselves: profiles/r05_fuzz_asm_emulator.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("family,cases", [("as", 3), ("os", 3), ("tn", 4)])
def test_emulator_campaign_fixed_seed(family, cases):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_asm_emulator.py"), "--family", family, "--seed", "5", "--cases", str(cases)],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"0 failing launch geometr(y/ies) of {cases}" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
