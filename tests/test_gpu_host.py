"""GPU: the repo's own C++ host (src/main.cpp -> raytracer_amd/bin/rt_host) drives the C ABI end to end:
OBJ + MTL + PPM textures -> normalise -> LBVH build -> one frame -> PPM.  Its image must equal the oracle's."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "raytracer_amd", "bin", "rt_host")


def _write_ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n# closed-form test texture\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb, np.uint8).tobytes())


def _read_ppm(path):
    raw = open(path, "rb").read()
    parts = raw.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = (int(x) for x in parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("textured", [False, True])
def test_cpp_host_frame_matches_oracle(oracle, rt, plane_obj_path, tmp_path, textured):
    from raytracer_amd import scenes
    assert os.path.exists(HOST), "raytracer_amd/bin/rt_host is not built (make host / __graft_entry__.build())"
    d = tmp_path
    shutil.copy(plane_obj_path, d / "plane.obj")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "plane_standin.mtl"), d / "plane_material.mtl")
    texs = [scenes.procedural_texture(128, 0), scenes.procedural_texture(32, 1)]
    _write_ppm(d / "standin_checker_0.rgb", texs[0])          # the names the MTL's map_Kd lines use
    _write_ppm(d / "standin_checker_1.rgb", texs[1])
    w, h = 160, 96
    cmd = [HOST, "--obj", str(d / "plane.obj"), "--width", str(w), "--height", str(h), "--mode", "shadow", "--cam", "0", "0", "0.6",
           "--out", str(d / "frame.ppm")] + (["--textured"] if textured else [])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raytracer_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["triangles"] == 6298 and info["nodes"] == 12595 and info["rays"] >= w * h
    img = _read_ppm(d / "frame.ppm")

    m = rt.normalize_mesh(rt.load_obj(str(d / "plane.obj")))
    b = oracle.build_lbvh(m.positions, m.normals, m.vert_idx, m.norm_idx)
    surf = None
    if textured:
        mats = [(tuple(x["diffuse"]), i) for i, x in enumerate(m.materials)]
        surf = oracle.Surface(oracle.gather_tri_uv(m.texcoords, m.tex_idx, m.material_ids, b["order"]), mats, texs)
    U = rt.make_uniforms(w, h, cam_pos=(0.0, 0.0, 0.6))
    ref, _ = oracle.render(b["nodes"], b["tri_pos"], b["tri_nor"], U, mode=1, n_threads=8, surface=surf)
    assert (ref["hit_tri"] != 0xFFFFFFFF).mean() > 0.02
    assert np.array_equal(img, ref["rgba8"][..., :3])


@pytest.mark.parametrize("textured", [False, True])
def test_cpp_host_rccl_world1_equals_single_process(plane_obj_path, tmp_path, textured):
    """rt_host --gpus 1: the multi-GPU program (fork before any HIP call, RCCL communicator, BVH blob export -> scatter /
    all-gather -> batched stripe render -> grouped send/recv gather -> de-interleave) with a world of one.  Its frame must
    be the single-process frame byte for byte, for 1 frame, for a full batch and for a batch with a ragged tail."""
    from raytracer_amd import scenes
    assert os.path.exists(HOST)
    d = tmp_path
    shutil.copy(plane_obj_path, d / "plane.obj")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "plane_standin.mtl"), d / "plane_material.mtl")
    _write_ppm(d / "standin_checker_0.rgb", scenes.procedural_texture(128, 0))
    _write_ppm(d / "standin_checker_1.rgb", scenes.procedural_texture(32, 1))
    w, h = 200, 120
    base = [HOST, "--obj", str(d / "plane.obj"), "--width", str(w), "--height", str(h), "--mode", "shadow", "--cam", "0", "0", "0.6"] \
        + (["--textured"] if textured else [])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "raytracer_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(base + ["--out", str(d / "single.ppm")], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    single = json.loads(res.stdout.strip().splitlines()[-1])
    ref = _read_ppm(d / "single.ppm")
    assert ref.reshape(-1, 3).max(axis=0).max() > 0
    for frames, batch in ((1, 8), (8, 4), (7, 3)):
        out = d / f"multi_{frames}_{batch}.ppm"
        res = subprocess.run(base + ["--gpus", "1", "--frames", str(frames), "--batch", str(batch), "--out", str(out)],
                             capture_output=True, text=True, timeout=180, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        info = json.loads(res.stdout.strip().splitlines()[-1])
        assert info["n_gpus"] == 1 and info["frames"] == frames and info["frames_per_pipeline"] == batch
        assert info["triangles"] == single["triangles"] and info["nodes"] == single["nodes"] and info["rays"] == single["rays"]
        assert np.array_equal(_read_ppm(out), ref), f"--gpus 1 --frames {frames} --batch {batch}"


def test_bench_runs_a_mesh_file_with_its_mtl_and_textures(tmp_path, plane_obj_path):
    """bench.py --scene obj:<file>: a mesh supplied out of band (the reference's web/plane.obj here) with its MTL and the images the
    MTL names beside it; the line must say which images were bound (`--no-texture`: the reference's constant albedo)."""
    import gzip
    import json
    import shutil
    import subprocess
    import sys

    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / "scene"
    d.mkdir()
    shutil.copy(plane_obj_path, d / "plane.obj")
    shutil.copy(os.path.join(root, "tests", "golden", "plane_standin.mtl"), d / "plane_material.mtl")     # the name web/plane.obj's mtllib line asks for
    rng = np.random.default_rng(5)
    for k in (0, 1):                                     # the two images the stand-in MTL names, as binary PPM
        img = rng.integers(0, 256, (32, 48, 3)).astype(np.uint8)
        (d / f"standin_checker_{k}.rgb").write_bytes(b"P6\n48 32\n255\n" + img.tobytes())
    lines = {}
    for name, extra in (("tex", []), ("plain", ["--no-texture"])):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scene", f"obj:{d / 'plane.obj'}", "--width", "320", "--height", "200",
                            "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-legs", "--no-fif2"] + extra,
                           capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        lines[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    w = lines["tex"]["config"]["workload"]
    assert "plane.obj" in w and "standin_checker_0.rgb" in w and "standin_checker_1.rgb" in w and "6298 triangles" in w
    assert "constant albedo" in lines["plain"]["config"]["workload"]
    assert lines["tex"]["value"] > 0 and lines["tex"]["config"]["rays_traced_per_frame"] == lines["plain"]["config"]["rays_traced_per_frame"]
