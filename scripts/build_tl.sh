#!/bin/bash
# analysis build of the library with extra -D flags into raytracer_amd/lib/librtcore_tl.so (use through RTCORE_LIBRARY):  scripts/build_tl.sh -DRT_SORT_TIMELINE
set -e
cd "$(dirname "$0")/.."
mkdir -p build_tl
F="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero"
for f in rt_api rt_build rt_trace; do /opt/rocm/bin/hipcc $F "$@" -c raytracer_amd/csrc/$f.hip -o build_tl/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/lib/librtcore_tl.so build_tl/rt_api.o build_tl/rt_build.o build_tl/rt_trace.o build/rt_scene.o
