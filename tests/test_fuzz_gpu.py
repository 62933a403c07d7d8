"""Randomised parity campaign (tools/fuzz.py): LSB / MSB / segmented / 64-bit sorts of random sizes (dense around every
class, tile and chunk boundary), key types, bit ranges, directions and key distributions, checked on the device
against torch's stable sort -- an implementation independent of the library and of the oracle.  The model is the
reference's own randomised sweeps (test_device_radix_sort.cu:1034-1046 sizes shrinking at random,
msb/tests entropy levels)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [101, 102])
def test_random_campaign(cuda, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "700", str(seed)], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "ALL 700 CASES OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_campaign_notices_a_wrong_key(cuda):
    """negative control: one flipped bit in one output key must fail the campaign"""
    env = dict(os.environ, FUZZ_SELFTEST="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "60", "9"], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 1 and "MISMATCH" in out.stdout
