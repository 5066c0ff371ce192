"""pytest configuration: markers, import path, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure both shared libraries exist (no-op when they are up to date): libsga.so is
    # cross-compiled by hipcc without a GPU, the oracle by gcc
    import shutil
    import spin_glass_anneal_rl_amd as sg
    if not os.path.exists(sg._native.library_path()) and (
            shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        sg._native.build()
    import oracle
    oracle.build()


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden
