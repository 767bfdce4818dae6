#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.npz from the CPU ORACLE (not from the reference, which
cannot be run here: SURVEY.md §8c).  These are regression pins of the oracle's own outputs, so a
later edit of oracle/ or of the kernels cannot drift silently.  Re-run only on purpose:
    python tests/golden/make_golden.py
"""
import gzip
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

PLANE_VIEW = dict(cam_pos=(0.0, 0.0, 1.3), rot_deg=(-90.0, 0.0, 0.0), scale=(2.5, 2.5, 2.5))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main():
    import raytracer_amd as rt                     # host-side ingest only (no GPU needed)
    from oracle import binding as oracle
    oracle.build()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "plane.obj")
        open(p, "wb").write(gzip.open(os.path.join(HERE, "plane.obj.gz")).read())
        mesh = rt.normalize_mesh(rt.load_obj(p))
    g = {}
    q = rt.generate_test_scene()
    qb = oracle.build_lbvh(q.positions, q.normals, q.vert_idx, q.norm_idx)
    g["quad_nodes"], g["quad_codes"], g["quad_order"] = qb["nodes"], qb["codes"], qb["order"]
    b = oracle.build_lbvh(mesh.positions, mesh.normals, mesh.vert_idx, mesh.norm_idx)
    g["plane_pos_crc"] = crc(mesh.positions)
    g["plane_codes_crc"], g["plane_order_crc"] = crc(b["codes"]), crc(b["order"])
    g["plane_topology_crc"] = crc(b["nodes"][:, [3, 7]])
    g["plane_nodes_crc"] = crc(b["nodes"] + np.float32(0.0))          # +0.0 folds -0 into +0
    g["plane_codes_head"], g["plane_order_head"] = b["codes"][:64], b["order"][:64]
    g["plane_root"] = b["nodes"][0]
    U = rt.make_uniforms(48, 32, **PLANE_VIEW)
    g["uniforms_48x32"] = U
    g["uniforms_default_1080p"] = rt.make_uniforms(1920, 1080)
    for name, mode, spp in (("primary", 0, 32), ("shadow", 1, 32), ("pt8", 2, 8)):
        out, st = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=mode, gi_samples=spp)
        g[f"{name}_rgba8"], g[f"{name}_hit_tri"], g[f"{name}_hit_t"] = out["rgba8"], out["hit_tri"], out["hit_t"]
        g[f"{name}_counters"] = np.array([st[k] for k in ("rays_primary", "rays_shadow", "rays_bounce", "box_tests", "tri_tests",
                                                             "hit_updates", "pixels", "stack_drops")], np.uint64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **g)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), {k: getattr(v, "shape", None) for k, v in g.items()})


if __name__ == "__main__":
    main()
