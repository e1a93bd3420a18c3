#!/bin/bash
# build an alternative libp2gpu with extra -D flags for ntt.hip: scratch/build_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../acvm-backend-plonky2_amd/csrc"
mkdir -p build_alt
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed $2 -c ntt.hip -o build_alt/ntt_$1.o
objs=$(ls build/*.o | grep -v "/ntt.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build_alt/libp2gpu_$1.so build_alt/ntt_$1.o $objs
echo built build_alt/libp2gpu_$1.so
