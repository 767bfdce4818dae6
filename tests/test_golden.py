"""Committed oracle-generated fixtures (tests/golden/oracle_golden.npz, made by make_golden.py).
They pin the ORACLE's outputs (parity against the reference itself is unpinned);
the CPU test guards the oracle + host ingest, the GPU test checks the HIP path against the same
fixtures without running the oracle."""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN

PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))
KEYS = ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests", "hit_updates", "pixels", "stack_drops")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "oracle_golden.npz"))


def _check_build(g, codes, order, nodes, mesh):
    assert crc(mesh.positions) == g["plane_pos_crc"]
    assert crc(codes) == g["plane_codes_crc"] and crc(order) == g["plane_order_crc"]
    assert np.array_equal(codes[:64], g["plane_codes_head"]) and np.array_equal(order[:64], g["plane_order_head"])
    assert crc(nodes[:, [3, 7]]) == g["plane_topology_crc"]
    assert crc(nodes + np.float32(0.0)) == g["plane_nodes_crc"]
    assert np.array_equal(nodes[0], g["plane_root"])


def test_oracle_against_golden(gold, oracle, rt, plane_mesh):
    q = rt.generate_test_scene()
    qb = oracle.build_lbvh(q.positions, q.normals, q.vert_idx, q.norm_idx)
    assert np.array_equal(qb["nodes"], gold["quad_nodes"]) and np.array_equal(qb["codes"], gold["quad_codes"])
    b = oracle.build_lbvh(plane_mesh.positions, plane_mesh.normals, plane_mesh.vert_idx, plane_mesh.norm_idx)
    _check_build(gold, b["codes"], b["order"], b["nodes"], plane_mesh)
    U = rt.make_uniforms(48, 32, **PLANE_VIEW)
    assert np.array_equal(U.view(np.uint32), gold["uniforms_48x32"].view(np.uint32))
    assert np.array_equal(rt.make_uniforms(1920, 1080).view(np.uint32), gold["uniforms_default_1080p"].view(np.uint32))
    for name, mode, spp in (("primary", 0, 32), ("shadow", 1, 32), ("pt8", 2, 8)):
        out, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, gi_samples=spp)
        assert np.array_equal(out["rgba8"], gold[f"{name}_rgba8"]) and np.array_equal(out["hit_tri"], gold[f"{name}_hit_tri"])
        assert np.array_equal(out["hit_t"], gold[f"{name}_hit_t"])
        assert [st[k] for k in KEYS] == list(gold[f"{name}_counters"])


@pytest.mark.gpu
def test_gpu_against_golden(gold, rt, plane_mesh, gpu_ctx):
    gpu_ctx.set_mesh(plane_mesh)
    gpu_ctx.build_lbvh()
    _check_build(gold, gpu_ctx.morton(), gpu_ctx.tri_order(), gpu_ctx.flatten_for_ssbo()["bvhData"], plane_mesh)
    U = gold["uniforms_48x32"]
    for name, mode, spp in (("primary", 0, 32), ("shadow", 1, 32), ("pt8", 2, 8)):
        out, st = gpu_ctx.render(U, rt.render_opts(mode=mode, gi_samples=spp, flags=rt.FLAG_COUNT_WORK))
        assert np.array_equal(out["hit_tri"], gold[f"{name}_hit_tri"])
        hit = gold[f"{name}_hit_tri"] != 0xFFFFFFFF
        assert np.all(np.abs(out["hit_t"][hit] - gold[f"{name}_hit_t"][hit]) <= 1e-5 * np.abs(gold[f"{name}_hit_t"][hit]))
        assert np.abs(out["rgba8"].astype(int) - gold[f"{name}_rgba8"].astype(int)).max() <= 1
        assert [st[k] for k in KEYS] == list(gold[f"{name}_counters"])
