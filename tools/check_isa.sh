#!/bin/bash
# Compiles mjhmc_amd/csrc/dense_sic.hip to gfx950 assembly with the Makefile's flags and checks what the design of the
# SparseImageCode kernels promises about their leapfrog step: between the markers the kernel
# source places around one leapfrog step (MJHMC_LEAPFROG_STEP_BEGIN / _END) there must be
#   - no scratch_ instruction   (a spill store / reload: the reload is a counted load, its wait drains the dictionary ring)
#   - no flat_load              (a generic-pointer load: waited for with vmcnt(0) lgkmcnt(0))
#   - at most TWO compiler-placed s_waitcnt vmcnt(0) per step, both at the head of the pass (the join of the two forms of
#     the residual's initialisation -- patches in LDS / in global memory -- before round 0), none inside the rounds (a drain of the LDS-DMA stream the design does not ask for; the owner's own
#                               wait for its block image is an inline-asm `s_waitcnt vmcnt(0)` between ;;#ASMSTART markers
#                               and is not counted)
# in the kernels the benchmark runs (n_coeffs 1024: sic_jump_kernel<*,*,*,4>, sic_flf_kernel<*,4>),
# and reports the register / spill figures of every SparseImageCode kernel.
# and, for the ProductOfT tile kernels (dense_pot.hip: float32 state; dense_pot64.hip: the reference's arithmetic), that the
# two GEMM loop bodies of every kernel (the basic blocks that hold the 128 MFMAs of two chunks) contain no scratch_
# instruction -- a spill reload inside the loop is a counted load whose wait drains the A-row prefetch; the epilogues may
# spill -- and, in dense_pot64.hip, that the per-step kick / drift pass (the block that streams the position through
# registers: >= 16 buffer loads and >= 16 buffer stores of the working copy) contains none either; with the VGPR / spill
# figures of every instance.
# Compile flags: the Makefile's own (make print-flags).
# usage: tools/check_isa.sh [out.s]        exit status 1 on a violation
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/mjhmc_dense_sic.s}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$ROOT/mjhmc_amd/csrc" || exit 2
FLAGS=$(make -s print-flags)
[ -n "${SKIP_COMPILE:-}" ] || $HIPCC $FLAGS \
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
    # one leapfrog step = the text between the markers.  hipcc may rotate the loop so that the END marker of an iteration
    # sits physically in front of the BEGIN marker: the step then runs from BEGIN to the branch back to END's block.
    steps = re.findall(r'MJHMC_LEAPFROG_STEP_BEGIN[^\n]*\n(.*?); MJHMC_LEAPFROG_STEP_END', body, flags=re.S)
    if not steps and 'MJHMC_LEAPFROG_STEP_BEGIN' in body and 'MJHMC_LEAPFROG_STEP_END' in body:
        lines = body.split('\n')
        b = next(i for i, l in enumerate(lines) if 'MJHMC_LEAPFROG_STEP_BEGIN' in l)
        e = next(i for i, l in enumerate(lines) if 'MJHMC_LEAPFROG_STEP_END' in l)
        if e < b:
            lab = None
            for i in range(e, -1, -1):
                m = re.match(r'^(\.LBB\w+):', lines[i])
                if m:
                    lab = m.group(1)
                    break
            stop = next((i for i in range(b, len(lines)) if lab and re.search(r's_c?branch\w*\s+' + re.escape(lab) + r'\b', lines[i])), None)
            if stop is not None:
                steps = ['\n'.join(lines[b + 1:stop + 1])]
    hot = ('sic_jump_kernel' in name or 'sic_flf_kernel' in name) and 'Li4EEE' in name
    n_scr = sum(len(re.findall(r'^\s*scratch_', s, flags=re.M)) for s in steps)
    n_flat = sum(len(re.findall(r'^\s*flat_load', s, flags=re.M)) for s in steps)
    steps_noasm = [re.sub(r';;#ASMSTART.*?;;#ASMEND', '', s, flags=re.S) for s in steps]
    n_drain = sum(len(re.findall(r'^\s*s_waitcnt[^\n]*vmcnt\(0\)', s, flags=re.M)) for s in steps_noasm)
    n_mfma = sum(len(re.findall(r'v_mfma', s)) for s in steps)
    n_glds = sum(len(re.findall(r'global_load_lds', s)) for s in steps)
    m = meta.get(name, {})
    tag = ''
    if hot and (n_scr or n_flat or n_drain > 2):
        bad += 1
        tag = '   <-- VIOLATION'
    if steps:
        print('%-100s vgpr %3s spilled %3s scratch %4s B | step: scratch_ %d flat_load %d vmcnt(0) %d (mfma %d, lds-dma %d)%s'
              % (name[9:105], m.get('vgpr'), m.get('vgpr_spill'), m.get('scratch'), n_scr, n_flat, n_drain, n_mfma, n_glds, tag))
print('check_isa: %s' % ('FAILED: %d kernel(s)' % bad if bad else 'ok'))
sys.exit(1 if bad else 0)
PY
RC=$?
for TU in dense_pot dense_pot64; do
  [ -n "${SKIP_COMPILE:-}" ] || $HIPCC $FLAGS --cuda-device-only -S $TU.hip -o /tmp/mjhmc_$TU.s 2> /dev/null || { echo "compile of $TU failed"; exit 2; }
  python3 - /tmp/mjhmc_$TU.s $TU <<'PY' || RC=1
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)', txt, flags=re.S):
    body = m.group(2)
    g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, body).group(1)) if re.search(r'\.%s:\s+(\d+)' % k, body) else None
    meta[m.group(1)] = dict(vgpr=g('vgpr_count'), agpr=g('agpr_count'), sgpr_spill=g('sgpr_spill_count'), scratch=g('private_segment_fixed_size'),
                            vgpr_spill=int(m.group(3)))
for name, body in re.findall(r'^(_ZN5mjhmc\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, flags=re.S | re.M):
    if 'pot' not in name:
        continue
    blocks, cur = [], []
    for l in body.split('\n'):
        if re.match(r'^\.LBB\w+:', l):
            blocks.append(cur)
            cur = []
        else:
            cur.append(l)
            if re.match(r'\s*s_c?branch', l):      # a loop body ends at its back edge: what follows falls through
                blocks.append(cur)
                cur = []
    blocks.append(cur)
    gemm = [b for b in blocks if sum('v_mfma' in x for x in b) >= 64]
    n_scr = sum(sum(re.match(r'\s*scratch_', x) is not None for x in b) for b in gemm)
    n_acc = sum(sum('v_accvgpr' in x for x in b) for b in gemm)
    # the float64-state kernel's per-step kick / drift passes: the blocks that stream the position through registers
    # (>= 16 buffer loads of the working copy and as many stores)
    # (the pass that stores the end point's position to the particle rows instead runs once per trajectory: not gated)
    passes = [b for b in blocks if sum('buffer_load' in x for x in b) >= 16 and sum('buffer_store' in x for x in b) >= 16
              and not any('v_mfma' in x for x in b)]
    p_scr = sum(sum(re.match(r'\s*scratch_', x) is not None for x in b) for b in passes)
    m = meta.get(name, {})
    hot = 'Li4E' in name and 'jump_kernel' in name
    tag = ''
    # (one or two folded reloads of a lane constant in a pass are tolerated: the allocator moves them around with every
    # change to the kernel; momentum elements reloaded inside the pass -- a handful of them, each behind a vmcnt(0) -- are not)
    if hot and (n_scr or p_scr > 2 or len(gemm) < 2 or ('pot64' in name and not passes)):
        bad += 1
        tag = '   <-- VIOLATION'
    if gemm:
        print('%-84s vgpr %3s spilled %3s scratch %4s B sgpr spills %3s | GEMM loop bodies %d: scratch_ %d (v_accvgpr %d)%s%s'
              % (name[9:93], m.get('vgpr'), m.get('vgpr_spill'), m.get('scratch'), m.get('sgpr_spill'), len(gemm), n_scr, n_acc,
                 (' | streamed passes %d: scratch_ %d' % (len(passes), p_scr)) if 'pot64' in name else '', tag))
print('check_isa %s: %s' % (sys.argv[2], 'FAILED: %d kernel(s)' % bad if bad else 'ok'))
sys.exit(1 if bad else 0)
PY
done
exit $RC
