"""GPU: a short fixed-seed run of the randomised parity soak (tests/fuzz_parity.py): random meshes (duplicate codes, degenerate /
huge / coplanar triangles, sizes around the 256-leaf block boundary), cameras, modes (the two older fragment shaders included),
random textures / texcoords / material ids, stack depths, layouts, node encodings, stripe shares and batch sizes - build,
surface records and frames (per frame, through the entry point that allocates nothing, and batched; with and without converged
hints) against the oracle, bit for bit.  About 34 000 cases from other seeds were run in round 2 (all identical; 10 500 of them with batches of up to 32 frames, after the last kernel change); 60 keep the
suite short."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_parity_soak():
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mean_hit = mod.run(60, 20250, log=lambda m: None)
    assert mean_hit > 0.05
