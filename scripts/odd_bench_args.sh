set -e
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2> gpurun_out/odd.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frames_per_launch'])"; }
run --steps 1 --warmup 0
run --steps 7 --warmup 1 --scene s70k --width 1280 --height 720
run --steps 50 --warmup 3 --no-legs
run --steps 5 --warmup 1 --batch 1 --no-legs
run --steps 3 --warmup 1 --mode pt --spp 4 --width 960 --height 540 --no-legs
run --steps 33 --warmup 2 --batch 32 --no-legs --scene s70k --width 640 --height 360
for k in 1 33; do
  echo "== N=2 rehearsal K=$k"
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29801 bench.py --gpus 2 --steps $k --warmup 1 --rehearse-on-one-gpu --scene s70k --width 640 --height 360 2> gpurun_out/odd2.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['assembled_frame_vs_single_gpu_render'])"
done
