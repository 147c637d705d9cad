#!/bin/bash
# One translation unit rebuilt with extra -D flags and linked with the shipped objects into fft_amd/lib/libspectre_hip_<tag>.so
# (A/B of compile-time choices between processes: SPECTRE_HIP_LIB=<that file>).   tools/build_variant.sh <tag> <unit.hip> -DX=1 ...
set -e
tag=$1; unit=$2; shift 2
cd "$(dirname "$0")/.."
obj=/tmp/variant_${tag}_$(basename "$unit" .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function -Wno-inline-asm "$@" -c fft_amd/csrc/"$unit" -o "$obj"
others=$(ls fft_amd/lib/obj/*.o | grep -v "/$(basename "$unit" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fft_amd/lib/libspectre_hip_${tag}.so $others "$obj"
echo fft_amd/lib/libspectre_hip_${tag}.so
