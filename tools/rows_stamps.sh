#!/bin/bash
# Timing build of the funnels' translation unit with cycle stamps at the phase boundaries of the relay kernel's sampling
# iterations (csrc/timing_variants.hpp: ROWS_STAMP): builds mjhmc_amd/lib/librows_stamps.so from the product's other objects
# with the Makefile's own flags.  Then, on the GPU box:
#   MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/librows_stamps.so python tools/rows_stamps.py [N] [L]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/mjhmc_amd/csrc" || exit 2
FLAGS=$(make -s print-flags) || exit 2
make -s -j8 all || exit 1
/opt/rocm/bin/hipcc $FLAGS -DROWS_STAMPS ${ROWS_EXTRA:-} -c energy_funnel.hip -o /tmp/rows_stamps_funnel.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librows_stamps.so $(ls build/*.o | grep -v "asan_\|hooks_\|/energy_funnel.o") /tmp/rows_stamps_funnel.o -ldl && echo built librows_stamps.so
