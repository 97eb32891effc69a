"""Run-to-run determinism of the dense kernels (tools/dense_determinism.py): the SparseImageCode rounds pass LDS images
between waves through barriers, counted vmcnt waits and LDS-DMA requests made by other waves -- a race there is a
run-to-run difference first.  The full-size soak is the tool itself; this is a short form of it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('key,n', [('c5', 40000), ('c3', 20000)])
def test_dense_kernels_are_deterministic_run_to_run(key, n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dense_determinism.py'), key, '4', '3', str(n)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and 'IDENTICAL' in text, text[-2000:]
