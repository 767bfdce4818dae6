"""CPU-only: the C-ABI library loads and exports every symbol include/rtcore.h declares; error
paths that need no GPU behave as documented (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_binding(rt):
    from raytracer_amd import _abi
    assert _header_symbols() == sorted(_abi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from raytracer_amd import _abi
    lib = C.CDLL(_abi.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"librtcore.so does not export {name}"
    assert lib.rt_abi_version() == 5


def test_no_torch_types_in_header():
    txt = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    assert "torch" not in txt and "at::" not in txt and "std::" not in txt


def test_product_never_touches_the_oracle():
    """The product package and its sources must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "rt_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
    assert "oracle" not in open(os.path.join(ROOT, "src", "main.cpp")).read().replace("CPU oracle", "")


def test_create_without_gpu_fails_loudly(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Context(0)
    assert e.value.code == -2          # RT_ERR_NO_DEVICE: there is no CPU fallback


def test_cpu_backend_is_refused(rt):
    from raytracer_amd import _abi
    lib = _abi.load()
    h = C.c_void_p()
    assert lib.rt_create(C.byref(h), b"cpu", None, 0) == -1
    assert b"hip" in lib.rt_last_error(None)


def test_packed_rows(rt):
    for h in (1, 15, 16, 17, 150, 1080, 2160):
        for world in (1, 2, 3, 8):
            rows = [rt.packed_rows(h, 16, r, world) for r in range(world)]
            assert sum(rows) == h
            assert rows[0] == max(rows)
            for r in range(world):
                assert rows[r] == sum(1 for y in range(h) if (y // 16) % world == r)


def test_render_opts_default(rt):
    from raytracer_amd import _abi
    o = _abi.RenderOpts()
    _abi.load().rt_render_opts_default(C.byref(o))
    # compute.wgsl:25-27,259
    assert (o.mode, o.gi_samples, o.gi_bounces, o.soft_samples, o.max_stack) == (2, 32, 3, 4, 12)
