import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Stated fp64 tolerances of the build (DESIGN.md "Tolerances")
FLUX_ATOL = 1e-12
LNPROB_RTOL = 1e-10
LNPROB_ATOL = 1e-7
H_RTOL = 1e-12


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    """All lnprob-style fixtures (those that carry theta rows and lnprob)."""
    names = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if n in ("hgrid", "taps", "conv_semantics", "nan_semantics", "host_helpers"):
            continue
        names.append(n)
    return names


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
