"""The product-form kernels with ONE workgroup (ZG_PSC_BLOCKS=1, read once per process, hence the child): tests/psc_single_block_check.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", ["1", "3"])
def test_in_loop_accumulator_flush_with_a_small_grid(blocks):
    env = dict(os.environ, ZG_PSC_BLOCKS=blocks, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "psc_single_block_check.py")], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0 and "single-block grid ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
