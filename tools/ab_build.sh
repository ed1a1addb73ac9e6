#!/bin/bash
# build a variant of libpapof.so with extra -D flags for sor.hip and kernels.hip into tools/ab/: tools/ab_build.sh <name> -DPAPOF_V_...=n ...
set -e
name=$1; shift
cd "$(dirname "$0")/../papteam_opticalflow_amd/csrc"
make -j6 > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value \
    -mllvm -structurizecfg-skip-uniform-regions=1 -I../../include "$@" -c sor.hip -o /tmp/sor_$name.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value \
    -I../../include "$@" -c kernels.hip -o /tmp/kernels_$name.o 2>/dev/null
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../tools/ab/libpapof_$name.so api.o /tmp/kernels_$name.o /tmp/sor_$name.o tiles.o -ldl
echo built tools/ab/libpapof_$name.so
