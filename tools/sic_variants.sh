#!/bin/bash
# Timing experiments on the SparseImageCode kernel: builds libsic_v<N>.so with -DSICV=<N> (a part of the round switched
# off: 1 no G1, 2 no G2, 3 no barriers, 4 no dictionary DMA, 5 no prior force) by recompiling dense_sic.hip only and
# linking it with the product's other objects
# (30t<k>: cycle stamps at position k of every round, see SIC_STAMP in dense_sic.hip).  Results of variants != 0 are garbage; only kernel times mean anything.
# usage: tools/sic_variants.sh 0 1 2 3 ...      then on the GPU box: tools/sic_leap_time.py
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/mjhmc_amd/csrc" || exit 2
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm -DMJHMC_JUMP_WAVES=1 \
      -DSICV=${v%%t*} $( [[ $v == *t* ]] && echo -DSICT=${v##*t} ) -c dense_sic.hip -o /tmp/sic_v$v.o 2>/tmp/sic_v$v.err && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsic_v$v.so $(ls build/*.o | grep -v "asan_\|hooks_\|dense_sic.o") /tmp/sic_v$v.o -ldl && echo built v$v || { echo "v$v FAILED:"; grep -m3 error /tmp/sic_v$v.err; } ) &
done
wait
