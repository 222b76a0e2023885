#!/bin/bash
# Builds an experimental variant of the library: tools/build_variant.sh NAME "-DPT_X=... -DPT_Y=..."
# -> pathtracer_amd/variants/NAME.so (load it with PT_AMD_LIB=...).  Not part of the product build: the kernels
# are compiled with -DPT_TUNING, which brings the PT_* environment knobs of kernels.hip back.
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
NAME=$1; EXTRA=$2
mkdir -p ../variants /tmp/ptvar
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DPT_TUNING $EXTRA -c kernels.hip -o /tmp/ptvar/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DPT_TUNING -x hip -c pt_api.cpp -o /tmp/ptvar/pt_api_tuning.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DPT_TUNING -x c++ -c bvh.cpp -o /tmp/ptvar/bvh_tuning.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$NAME.so obj_loader.o scene.o camera.o /tmp/ptvar/bvh_tuning.o bvh_device.o /tmp/ptvar/pt_api_tuning.o pt_objects.o pt_readback.o pt_comm.o /tmp/ptvar/$NAME.o -ldl
echo built ../variants/$NAME.so
