#!/bin/bash
# Timing builds of the SparseImageCode kernel with cycle stamps at position <k> (1..6) of every round of a leapfrog
# step's pass (csrc/timing_variants.hpp: SIC_STAMP): builds libsic_t<k>.so by recompiling dense_sic.hip only, with the
# Makefile's own flags (make print-flags), and linking it with the product's other objects.
# usage: tools/sic_variants.sh 1 2 3 4 5 6      then on the GPU box: tools/run_sic_variants.sh (tools/sic_leap_time.py)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/mjhmc_amd/csrc" || exit 2
FLAGS=$(make -s print-flags) || exit 2
make -s -j8 all || exit 1
for k in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DSIC_STAMPS=$k -c dense_sic.hip -o /tmp/sic_t$k.o 2>/tmp/sic_t$k.err && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsic_t$k.so $(ls build/*.o | grep -v "asan_\|hooks_\|dense_sic.o") /tmp/sic_t$k.o -ldl && echo built t$k || { echo "t$k FAILED:"; grep -m3 error /tmp/sic_t$k.err; } ) &
done
wait
