"""GPU: raytracer_amd.dist at world sizes 2 and 3 on ONE GPU (tests/world_on_one_gpu.py: N processes, each with its own rt
context on cuda:0, collectives over gloo through host memory) — BVH replication into adopted device buffers, every rank's own
stripe share, the exchange and the de-interleave with rank != 0 and world > 1, for v1 and v2 node records; and bench.py's N > 1
path end to end (`--rehearse-on-one-gpu`), whose line must report the assembled frame identical to rank 0's own render.
RCCL itself is exercised at world 1 (tests/test_gpu_dist.py, tests/test_gpu_host.py): it refuses two ranks on one device."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(n, script_args, timeout=300):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_dist_paths_with_several_ranks_on_one_gpu(world):
    r = _launch(world, [os.path.join(ROOT, "tests", "world_on_one_gpu.py")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):
        assert f"OK rank {rank} of {world}" in r.stdout


@pytest.mark.parametrize("extra", [[], ["--batch", "1"], ["--exchange", "all", "--batch", "4"]], ids=["batch16", "per_frame", "all_gather_batch4"])
def test_bench_n2_rehearsal_assembles_the_right_frame(extra):
    r = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--scene", "s70k", "--width", "640", "--height",
                    "360", "--steps", "7", "--warmup", "2"] + extra)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["assembled_frame_vs_single_gpu_render"] == "identical"
    assert "rehearsal" in line and "cpu_baseline" not in line and line["value"] > 0
