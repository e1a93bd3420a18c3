#!/bin/bash
# build an alternative libp2gpu with extra -D flags for one translation unit:
#   scratch/build_variant.sh <name> "<flags>" [unit]      (unit defaults to ntt)
set -e
cd "$(dirname "$0")/../acvm-backend-plonky2_amd/csrc"
U=${3:-ntt}
mkdir -p build_alt
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed $2 -c $U.hip -o build_alt/${U}_$1.o
objs=$(ls build/*.o | grep -v "/$U.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build_alt/libp2gpu_$1.so build_alt/${U}_$1.o $objs
echo built build_alt/libp2gpu_$1.so
