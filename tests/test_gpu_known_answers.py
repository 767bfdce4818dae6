"""GPU: the two pixels worked in float64 by tests/golden/derive_known_answers.py (soft shadows on the 6 x 6 grid; a path-traced
pixel with GI_SAMPLES = 2, GI_BOUNCES = 1), rendered by the HIP path from the same hand-written flat buffers."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _derivation():
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derive_known_answers.py")
    spec = importlib.util.spec_from_file_location("derive_known_answers", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("layout", ["wide64", "aos32"])
def test_worked_pixels_on_the_gpu(rt, oracle, layout):
    d = _derivation()
    lay = rt.LAYOUT_WIDE64 if layout == "wide64" else rt.LAYOUT_AOS32
    ctx = rt.Context(0)
    try:
        a = d.case_a()
        nodes, tp, tn = d.flat_buffers()
        ctx.set_bvh(nodes, tp, tn)
        U = d.uniforms(d.LIGHT_A)
        out, _ = ctx.render(U, rt.render_opts(mode=rt.MODE_SHADOW, gi_samples=32, soft_samples=4, layout=lay))
        x, y = a["pixel"]
        assert out["hit_tri"][y, x] == 1 and list(out["rgba8"][y, x]) == a["rgba8"] == [166, 154, 0, 255]
        assert abs(float(out["hit_t"][y, x]) - a["hit_t"]) < 2e-6 * a["hit_t"]
        ref, _ = oracle.render(nodes, tp, tn, U, mode=oracle.MODE_SHADOW, gi_samples=32, soft_samples=4)
        for k in ("rgba8", "hit_tri", "hit_t"):
            assert np.array_equal(out[k], ref[k])

        b = d.case_b()
        nodes, tp, tn = d.flat_buffers()
        ctx.set_bvh(nodes, tp, tn)
        U = d.uniforms(d.LIGHT_B)
        for split in ("1", "0"):                                   # sample-parallel PT pipeline and one lane per pixel
            os.environ["RTCORE_PT_SPLIT"] = split
            out, _ = ctx.render(U, rt.render_opts(mode=rt.MODE_PT, gi_samples=2, gi_bounces=1, soft_samples=4, layout=lay))
            assert list(out["rgba8"][y, x]) == b["rgba8"] == [198, 183, 0, 255]
        os.environ.pop("RTCORE_PT_SPLIT", None)
        ref, _ = oracle.render(nodes, tp, tn, U, mode=oracle.MODE_PT, gi_samples=2, gi_bounces=1, soft_samples=4)
        for k in ("rgba8", "hit_tri", "hit_t"):
            assert np.array_equal(out[k], ref[k])
    finally:
        os.environ.pop("RTCORE_PT_SPLIT", None)
        ctx.close()
