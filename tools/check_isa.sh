#!/bin/bash
# Compiles mjhmc_amd/csrc/dense_sic.hip to gfx950 assembly with the Makefile's flags and checks what the design of the
# SparseImageCode kernels promises about their leapfrog step: between the markers the kernel
# source places around one leapfrog step (MJHMC_LEAPFROG_STEP_BEGIN / _END) there must be
#   - no scratch_ instruction   (a spill store / reload: the reload is a counted load, its wait drains the dictionary ring)
#   - no flat_load              (a generic-pointer load: waited for with vmcnt(0) lgkmcnt(0))
#   - no s_waitcnt vmcnt(0)     (a full drain of the LDS-DMA ring)
# in the kernels the benchmark runs (n_coeffs 1024, patches in LDS: sic_jump_kernel<*,*,*,4,false>, sic_flf_kernel<*,4,false>),
# and reports the register / spill figures of every SparseImageCode kernel.
# usage: tools/check_isa.sh [out.s]        exit status 1 on a violation
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/mjhmc_dense_sic.s}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$ROOT/mjhmc_amd/csrc" || exit 2
[ -n "${SKIP_COMPILE:-}" ] || $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm -DMJHMC_JUMP_WAVES=1 \
  --cuda-device-only -S dense_sic.hip -o "$OUT" 2> /dev/null || { echo "compile failed"; exit 2; }
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', txt):
    meta[m.group(1)] = dict(scratch=int(m.group(2)), sgpr_spill=int(m.group(3)), vgpr=int(m.group(4)), vgpr_spill=int(m.group(5)))
for name, body in re.findall(r'^(_ZN5mjhmc\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, flags=re.S | re.M):
    if 'sic_' not in name:
        continue
    steps = re.findall(r'MJHMC_LEAPFROG_STEP_BEGIN.*?\n(.*?); MJHMC_LEAPFROG_STEP_END', body, flags=re.S)
    hot = ('sic_jump_kernel' in name or 'sic_flf_kernel' in name) and 'Li4ELb0EEE' in name
    n_scr = sum(len(re.findall(r'^\s*scratch_', s, flags=re.M)) for s in steps)
    n_flat = sum(len(re.findall(r'^\s*flat_load', s, flags=re.M)) for s in steps)
    n_drain = sum(len(re.findall(r'^\s*s_waitcnt[^\n]*vmcnt\(0\)', s, flags=re.M)) for s in steps)
    n_mfma = sum(len(re.findall(r'v_mfma', s)) for s in steps)
    n_glds = sum(len(re.findall(r'global_load_lds', s)) for s in steps)
    m = meta.get(name, {})
    tag = ''
    if hot and (n_scr or n_flat or n_drain):
        bad += 1
        tag = '   <-- VIOLATION'
    if steps:
        print('%-100s vgpr %3s spilled %3s scratch %4s B | step: scratch_ %d flat_load %d vmcnt(0) %d (mfma %d, lds-dma %d)%s'
              % (name[9:105], m.get('vgpr'), m.get('vgpr_spill'), m.get('scratch'), n_scr, n_flat, n_drain, n_mfma, n_glds, tag))
print('check_isa: %s' % ('FAILED: %d kernel(s)' % bad if bad else 'ok'))
sys.exit(1 if bad else 0)
PY
