#!/bin/bash
# Timing build of the ProductOfT kernels with cycle stamps around the parts of a gradient evaluation (dense_pot.hip:
# POT_STAMP): builds mjhmc_amd/lib/libpot_stamps.so from the product's other objects.  On the GPU box:
#   MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libpot_stamps.so python tools/pot_stamps.py
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/mjhmc_amd/csrc" || exit 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm -DMJHMC_JUMP_WAVES=1 \
  -DPOT_STAMPS ${POTV:+-DPOTV=$POTV} -c dense_pot.hip -o /tmp/pot_stamps${POTV:-}.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpot_stamps${POTV:-}.so $(ls build/*.o | grep -v "asan_\|hooks_\|dense_pot.o") /tmp/pot_stamps${POTV:-}.o -ldl && echo built ${POTV:-}
