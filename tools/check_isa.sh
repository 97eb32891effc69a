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
    if ('pot_fix_kernel' in name or 'pot64_decide_kernel' in name) and meta.get(name, {}).get('scratch'):
        print('%-84s scratch %s B   <-- VIOLATION (the fix / decide kernels hold no spills)' % (name[9:93], meta[name]['scratch']))
        bad += 1
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
    # scratch reloads tolerated in the streamed per-step pass, per instance (what each one has TODAY: a regression of a
    # clean instance fails; the allocator's one or two folded lane-constant reloads of the others are pinned at their count)
    allow = {'pot64_jump_kernelILi4ELb0ELi0EE': 1, 'pot64_jump_kernelILi4ELb1ELi0EE': 1}.get(re.sub(r'^_ZN5mjhmc\d+', '', name).split('Ev')[0], 0)
    new_kernels = any(t in name for t in ('pot_fix_kernel', 'pot64_decide_kernel'))
    if (hot and (n_scr or p_scr > allow or len(gemm) < 2 or ('pot64' in name and not passes))) or (new_kernels and m.get('scratch')):
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
# The row-form kernels of the funnels (elementwise.hpp): what their design promises --
#   mjhmc_fused_rows_kernel / mjhmc_fused_rows_relay_kernel (the product's form since round 6: four-wave workgroups, the cold
#   caches' inverse-L trajectories pooled): no scratch memory at all; the leapfrog-step loops -- the wave's own and the
#   pool's two-lanes-per-particle one -- are straight vector code (<= 150 / 170 instructions for the 32-coordinate row,
#   nothing from memory);
#   mjhmc_traj_rows_kernel (full rows): step loops the same; between the first row store and the end of the forward path no
#   wait that drains the loads/stores counter (a vmcnt(0) there is a wait for the previous store's round trip to memory).
[ -n "${SKIP_COMPILE:-}" ] || $HIPCC $FLAGS --cuda-device-only -S energy_funnel.hip -o /tmp/mjhmc_energy_funnel.s 2> /dev/null || { echo "compile of energy_funnel failed"; exit 2; }
python3 - /tmp/mjhmc_energy_funnel.s <<'PY' || RC=1
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)', txt, flags=re.S):
    body = m.group(2)
    g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, body).group(1)) if re.search(r'\.%s:\s+(\d+)' % k, body) else None
    meta[m.group(1)] = dict(vgpr=g('vgpr_count'), scratch=g('private_segment_fixed_size'), vgpr_spill=int(m.group(3)))
for name, body in re.findall(r'^(_ZN5mjhmc\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, flags=re.S | re.M):
    if ('rows_kernel' not in name and 'rows_relay_kernel' not in name) or 'IdEE' not in name:
        continue
    lines = body.split('\n')
    pos = {}
    for k, l in enumerate(lines):
        mm = re.match(r'^(\.LBB\w+):', l)
        if mm:
            pos[mm.group(1)] = k
    instr = lambda l: bool(re.match(r'^\s+[a-z]', l)) and not l.strip().startswith(';')
    steps = []                                   # innermost loops made of float64 multiply-adds: the leapfrog steps
    for k, l in enumerate(lines):
        mm = re.search(r's_cbranch_scc1\s+(\.LBB\w+)', l)
        if mm and pos.get(mm.group(1), 1 << 30) < k:
            seg = [x.strip() for x in lines[pos[mm.group(1)]:k + 1] if instr(x)]
            if sum(x.startswith(('v_fma_f64', 'v_fmac_f64')) for x in seg) >= 40 and len(seg) < 400:
                steps.append(seg)
    n_mem = sum(sum(x.startswith(('scratch_', 'global_', 'buffer_', 'ds_', 'flat_')) for x in seg) for seg in steps)
    longest = max([len(seg) for seg in steps] or [0])
    m = meta.get(name, {})
    fused = 'fused_rows' in name
    relay = 'rows_relay_kernel' in name      # (its own-trajectory step carries the pool's bookkeeping registers: <= 170)
    full4 = 'Li8ELi2ELb1E' in name
    drains = None
    if not fused and 'Lb1E' in name:             # full rows: the forward path's stores
        first = next((k for k, l in enumerate(lines) if 'global_store_dwordx4' in l), None)
        end = next((k for k in range(first or 0, len(lines)) if 's_endpgm' in lines[k] or 's_branch' in lines[k]), len(lines))
        drains = sum(bool(re.search(r's_waitcnt[^\n]*vmcnt\(0\)', l)) for l in lines[first:end]) if first is not None else None
    tag = ''
    if (not steps or n_mem or (full4 and longest > (170 if relay else 150)) or (fused and m.get('scratch')) or (drains not in (None, 0))):
        bad += 1
        tag = '   <-- VIOLATION'
    print('%-92s vgpr %3s spilled %3s scratch %4s B | step loops %d (longest %d instructions, memory operations %d)%s%s'
          % (name[9:101], m.get('vgpr'), m.get('vgpr_spill'), m.get('scratch'), len(steps), longest, n_mem,
             '' if drains is None else ' | vmcnt(0) after the first row store: %d' % drains, tag))
print('check_isa energy_funnel (row form): %s' % ('FAILED: %d kernel(s)' % bad if bad else 'ok'))
sys.exit(1 if bad else 0)
PY
exit $RC
