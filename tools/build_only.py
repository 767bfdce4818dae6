#!/usr/bin/env python3
"""LBVH build only (for rocprofv3 --kernel-trace --stats): N builds of a stand-in scene, stage times from hipEvents."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import raytracer_amd as rt
from raytracer_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "s1m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
surface = (sys.argv[3] != "nosurface") if len(sys.argv) > 3 else True
m = scenes.s1m() if name == "s1m" else scenes.s70k() if name == "s70k" else scenes.lattice(rt.normalize_mesh(scenes.s1m()), int(name[7:]))
m = rt.normalize_mesh(m)
ctx = rt.Context(0)
if 2 * m.n_tri - 1 > (1 << 24):
    ctx.set_node_encoding(rt.NODES_V2)
ctx.set_mesh(m)
if surface:
    ctx.set_surface(m)
runs = [ctx.build_lbvh() for _ in range(n)]
ctx.set_build_stage_timing(True)      # the stage breakdown from builds of its own (the events between launches cost ~2 us each)
staged = [ctx.build_lbvh() for _ in range(max(4, n // 3))]
keys = [k for k in runs[0] if k.startswith("ms_")]
med = {k: float(np.median([r[k] for r in (runs if k in ("ms_total", "ms_pack") else staged)[2:]])) for k in keys}
med["ms_total_with_stage_events"] = float(np.median([r["ms_total"] for r in staged[2:]]))
print(name, m.n_tri, "triangles, median of", n - 2, "builds:", {k: round(v, 4) for k, v in med.items()},
      "frac", round(560.0 * m.n_tri / (med["ms_total"] * 1e-3) / 8e12, 4), flush=True)
ctx.close()
