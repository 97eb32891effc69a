#!/bin/bash
# Timing build of a ProductOfT translation unit with cycle stamps around the parts of a gradient evaluation
# (csrc/timing_variants.hpp: POT_STAMP): builds mjhmc_amd/lib/libpot_stamps.so from the product's other objects, with
# the Makefile's own compile flags (make print-flags).
#   tools/pot_stamps.sh [dense_pot | dense_pot64]        then, on the GPU box:
#   MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libpot_stamps.so python tools/pot_stamps.py        (dense_pot: float32 state)
#   MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libpot_stamps.so python tools/pot64_stamps.py      (dense_pot64: float64 state)
set -u
TU=${1:-dense_pot}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/mjhmc_amd/csrc" || exit 2
FLAGS=$(make -s print-flags) || exit 2
make -s -j8 all || exit 1
/opt/rocm/bin/hipcc $FLAGS -DPOT_STAMPS -c $TU.hip -o /tmp/pot_stamps_$TU.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpot_stamps.so $(ls build/*.o | grep -v "asan_\|hooks_\|/$TU.o") /tmp/pot_stamps_$TU.o -ldl && echo built libpot_stamps.so with stamps in $TU
