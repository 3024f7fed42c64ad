#!/bin/bash
# build an A/B variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>  ->  csrc/variants/libfsq_<name>.so
# (only fsq_fit_rounds.hip is recompiled with the flags; the other objects are the production ones)
set -e
cd "$(dirname "$0")/../fluorosequencingimageanalysis_amd/csrc"
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c fsq_fit_rounds.hip -o variants/fsq_fit_rounds_$name.o 2>&1 | grep -v hip-link || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libfsq_$name.so variants/fsq_fit_rounds_$name.o $(ls *.o | grep -v '^fsq_fit_rounds.o$') -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo built variants/libfsq_$name.so
