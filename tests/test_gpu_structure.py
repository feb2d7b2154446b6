"""The launch structure chosen for a batch (csrc/capi.hip: walker kernel / tile geometry / final reduction / several
instruments in one launch -- crossovers measured on one box and coded with their measurements) is checked against the
alternatives ON THE BOX THE TESTS RUN ON: the automatic choice must be within 12 % of the fastest forced alternative.
(`scripts/structure_check.py` prints the full table for all configs.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))


@pytest.mark.parametrize("config,W,options", [
    ("C1", 128, ("walker", "geom", "finalize")),
    ("C1", 512, ("walker", "geom", "finalize")),
    ("C1", 2048, ("walker", "geom", "finalize")),
    ("C3", 64, ("geom", "finalize", "tile_multi")),
    ("C3", 512, ("geom", "finalize", "tile_multi")),
])
def test_automatic_launch_structure_is_near_the_fastest(config, W, options):
    torch = pytest.importorskip("torch")
    import structure_check as sc
    r = sc.check(config, W, None, options, npass=30 if config == "C1" else 10)
    table = {k: (round(1e3 * v[0], 1), v[1]) for k, v in r["alternatives"].items()}
    assert r["ratio"] <= 1.12, (f"{config} W={W}: automatic structure {r['auto'][1]} takes {1e3 * r['auto'][0]:.1f} us, "
                                f"alternatives {table}")
