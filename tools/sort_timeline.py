#!/usr/bin/env python3
"""Analysis aid (needs a library built with `make lib EXTRA=-DRT_SORT_TIMELINE`, e.g. into another file through
RTCORE_LIBRARY): where does a onesweep tile of the LBVH build spend its time?  Stage clocks per tile and pass."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd import _abi, scenes

ctx = rt.Context(0)
m = rt.normalize_mesh(scenes.s1m())
ctx.set_mesh(m)
ctx.set_build_stage_timing(True)
for _ in range(3):
    st = ctx.build_lbvh()
L = C.CDLL(_abi.LIB_PATH)
buf = np.zeros((3, 4096, 8), np.uint64)
rc = L.rt_debug_sort_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0
print("ms_sort", st["ms_sort"], "ms_morton", st["ms_morton"])
names = ["start->counts published", "->ranked+scans", "->LDS scatter", "->look-back done", "->stored"]
for p in range(3):
    b = buf[p].astype(np.int64)
    b = b[b[:, 5] > 0]
    t0 = b[:, 0].min()
    print(f"pass {p}: tiles {len(b)}  first start -> last end {(b[:, 5].max() - t0) * 1e-2:.1f} us; tile start spread {(b[:, 0].max() - t0) * 1e-2:.1f} us")
    for k in range(5):
        d = (b[:, k + 1] - b[:, k]) * 1e-2
        print(f"   {names[k]:24s} p10 {np.percentile(d, 10):6.2f}  p50 {np.percentile(d, 50):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
ctx.close()
