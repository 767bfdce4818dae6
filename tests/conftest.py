import gzip
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


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/Makefile."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def rt():
    import raytracer_amd
    return raytracer_amd


@pytest.fixture(scope="session")
def plane_obj_path(tmp_path_factory):
    """web/plane.obj of the reference (data fixture, stored gzipped under tests/golden)."""
    d = tmp_path_factory.mktemp("plane")
    p = d / "plane.obj"
    with gzip.open(os.path.join(GOLDEN, "plane.obj.gz"), "rb") as f:
        p.write_bytes(f.read())
    return str(p)


@pytest.fixture(scope="session")
def plane_mesh(rt, plane_obj_path):
    return rt.normalize_mesh(rt.load_obj(plane_obj_path))


def random_soup(n_tri, seed, quantum=None, scale=1.0, size=0.02):
    """Seeded triangle soup; `quantum` snaps centroids onto a coarse grid to force duplicate Morton codes."""
    rng = np.random.default_rng(seed)
    c = rng.random((n_tri, 1, 3))
    if quantum:
        c = np.floor(c * quantum) / quantum
    p = (c + (rng.random((n_tri, 3, 3)) - 0.5) * size) * scale
    pos = p.reshape(-1, 3).astype(np.float32)
    nor = rng.standard_normal((n_tri * 3, 3)).astype(np.float32)
    nor /= np.linalg.norm(nor, axis=1, keepdims=True)
    idx = np.arange(n_tri * 3, dtype=np.uint32)
    import raytracer_amd
    return raytracer_amd.Mesh(pos, nor, idx, idx.copy())


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
