"""GPU: a short fixed-seed run of the randomised parity soak (tests/fuzz_parity.py): random meshes (duplicate codes, degenerate /
huge / coplanar triangles, sizes around the 256-leaf block boundary), cameras, modes, stack depths, layouts, node encodings,
stripe shares and batch sizes - build and frames (per frame and batched, with and without converged hints) against the oracle,
bit for bit.  600 cases from other seeds were run while this was written; 40 keep the suite short."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_parity_soak():
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mean_hit = mod.run(40, 20250, log=lambda m: None)
    assert mean_hit > 0.05
