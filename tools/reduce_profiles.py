"""Reduce a tools/profile_round.sh collection (rocprofv3 CSVs under gpurun_out/<dir>) to what is committed under
profiles/<round>/ and to profiles/hbm_traffic.json (the `roofline.traffic` source of bench.py).

usage: python tools/reduce_profiles.py gpurun_out/r2prof profiles/r02

HBM bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB and come from separate
passes; on gfx950 FETCH_SIZE reports half of wide streaming reads (calibrated in round 1 on a kernel reading a known
409.6 MB: factor 1.9996) -> bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024.  "Per launch" is per timed unit of the
bench: one fused launch for C2, one sampling iteration (jump kernel + its compacted / inverse-L passes) otherwise.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = {'c2': 'mjhmc_jump_kernel', 'c2nofuse': 'mjhmc_jump_kernel', 'c3': 'pot_jump_kernel', 'c3f64': 'pot64_jump_kernel',
        'c4': 'mjhmc_fused_rows_relay_kernel', 'c4nofuse': 'mjhmc_traj_rows_kernel', 'c5': 'sic_jump_kernel', 'c5bf16': 'sic_jump_kernel'}
OURS = ('mjhmc', 'pot_', 'pot64_', 'sic_', 'compact_list', 'cold_list')


def counters(path):
    """{counter: {kernel short name: [values per dispatch]}}"""
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            out.setdefault(row['Counter_Name'], {}).setdefault(row['Kernel_Name'], []).append(float(row['Counter_Value']))
    return out


def launches_per_iteration(path, kernel):
    """Big dense batches run as two halves on two streams (api.hip: half_args): the main kernel is then dispatched once per
    half and iteration, from two queues."""
    queues = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row['Kernel_Name']:
                queues.add(row.get('Queue_Id'))
    return max(1, len(queues))


def find(d, suffix):
    for dirpath, _, files in os.walk(d):
        for fn in files:
            if fn.endswith(suffix):
                return os.path.join(dirpath, fn)
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    traffic_path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    notes = []
    for w in ('c2', 'c2nofuse', 'c3', 'c3f64', 'c4', 'c4nofuse', 'c5', 'c5bf16'):
        ks = find(os.path.join(src, 'kt_' + w), 'kernel_stats.csv')
        if ks:
            shutil.copy(ks, os.path.join(dst, w + '_kernel_stats.csv'))
        b = os.path.join(src, 'bench_%s.json' % w)
        if os.path.exists(b) and os.path.getsize(b):
            shutil.copy(b, os.path.join(dst, 'bench_%s_under_rocprof.json' % w))
        per = {}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            p = find(os.path.join(src, 'pmc_%s_%s' % (w, c)), 'counter_collection.csv')
            if not p:
                continue
            cs = counters(p).get(c, {})
            main = [v for k, vals in cs.items() if MAIN[w] in k for v in vals]
            ours = [v for k, vals in cs.items() if any(t in k for t in OURS) for v in vals]
            if not main:
                continue
            fused = w == 'c2'
            if fused:      # fused launches only (the bench also runs a few one-iteration launches for the secondary figure)
                big = [v for k, vals in cs.items() if MAIN[w] in k and k.rstrip().endswith('true>(mjhmc::JumpArgs<double>, mjhmc::IsoGaussF<double> const)') for v in vals]
                n_units = max(len(big), 1) / float(launches_per_iteration(p, 'true>(mjhmc::JumpArgs<double>, mjhmc::IsoGaussF<double> const)'))
                per[c] = sum(big) / n_units            # a fused launch runs as several parts on as many streams
            elif w == 'c4':
                # fused launches in row form (since round 5), one queue; the bench's few one-iteration launches (trajectory +
                # jump-process kernels) are other kernels
                per[c] = sum(main) / float(len(main))
            elif w == 'c4nofuse':
                # `bench.py --workload c4 --steps 1`: one trajectory launch (row form) + one jump-process launch per iteration;
                # the warm-up's fused call is another kernel
                single = [v for k, vals in cs.items() if ('mjhmc_traj_rows_kernel' in k or 'mjhmc_step_kernel' in k) for v in vals]
                per[c] = sum(single) / float(len(main))
            elif w == 'c2nofuse':
                # `bench.py --workload c2 --steps 1`: the timed calls are single iterations (never fused); its warm-up is one
                # fused call (three parts on three streams) -- only the one-iteration launches count here
                fused_tag = 'true>(mjhmc::JumpArgs<double>, mjhmc::IsoGaussF<double> const)'
                single = [v for k, vals in cs.items() if MAIN[w] in k and not k.rstrip().endswith(fused_tag) for v in vals]
                per[c] = sum(single) / max(len(single), 1)
            else:
                n_units = len(main) / float(launches_per_iteration(p, MAIN[w]))
                per[c] = sum(ours) / n_units            # every kernel of an iteration, per iteration
            # keep a trimmed copy of the pass: our kernels only
            with open(p) as f, open(os.path.join(dst, '%s_pmc_%s.csv' % (w, c.lower())), 'w') as g:
                rd = csv.DictReader(f)
                wr = csv.writer(g)
                wr.writerow(['Kernel_Name', 'Counter_Name', 'Counter_Value', 'Grid_Size', 'VGPR_Count', 'Scratch_Size', 'LDS_Block_Size'])
                kept = 0
                for row in rd:
                    if any(t in row['Kernel_Name'] for t in OURS) and kept < 400:
                        wr.writerow([row['Kernel_Name'][:110], row['Counter_Name'], row['Counter_Value'], row['Grid_Size'],
                                     row['VGPR_Count'], row['Scratch_Size'], row['LDS_Block_Size']])
                        kept += 1
        if len(per) == 2:
            key = {'c2nofuse': 'c2_one_iteration_per_launch', 'c4nofuse': 'c4_one_iteration_per_launch'}.get(w, w)
            total = per['FETCH_SIZE'] * 1024 * 2.0 + per['WRITE_SIZE'] * 1024
            traffic[key] = {'bytes_per_launch': total, 'iterations_per_launch': 1, 'fused': w in ('c2', 'c4'),
                            'FETCH_SIZE_KiB': per['FETCH_SIZE'], 'WRITE_SIZE_KiB': per['WRITE_SIZE'], 'read_correction': 2.0,
                            'what': 'one fused launch (any number of fused iterations: the state crosses HBM once)' if w in ('c2', 'c4')
                                    else 'all kernels of one sampling iteration',
                            'sources': ['%s/%s_pmc_fetch_size.csv' % (os.path.basename(dst), w),
                                        '%s/%s_pmc_write_size.csv' % (os.path.basename(dst), w)]}
            notes.append('%s: HBM bytes per launch %.4g (read %.4g, write %.4g)' % (key, total, per['FETCH_SIZE'] * 2048, per['WRITE_SIZE'] * 1024))
        # SQ passes
        sq = {}
        for tag in ('sq1', 'sq2'):
            p = find(os.path.join(src, '%s_%s' % (tag, w)), 'counter_collection.csv')
            if p:
                for cname, per_k in counters(p).items():
                    vals = [v for k, vs in per_k.items() if MAIN[w] in k and (w != 'c2' or 'true>(' in k) for v in vs]
                    if vals:
                        parts = launches_per_iteration(p, 'true>(' if w == 'c2' else MAIN[w])
                        sq[cname] = (sum(vals) / (len(vals) / float(parts)), int(len(vals) / parts))
        if sq:
            with open(os.path.join(dst, '%s_pmc_sq.txt' % w), 'w') as g:
                g.write('rocprofv3 --pmc passes (tools/profile_round.sh), dominant kernel %s, average per launch (all parts / halves of a split launch together)\n' % MAIN[w])
                for k in sorted(sq):
                    g.write('%s %.6g  (over %d launches)\n' % (k, sq[k][0], sq[k][1]))
                if 'SQ_INSTS_VALU' in sq and 'GRBM_GUI_ACTIVE' in sq:
                    cyc = sq['GRBM_GUI_ACTIVE'][0] / 8.0
                    g.write('GRBM_GUI_ACTIVE / 8 XCDs = %.4g cycles per launch; VALU issue cycles = SQ_INSTS_VALU * 4 / 1024 SIMDs '
                            '= %.4g -> vector pipe busy %.0f %%\n' % (cyc, sq['SQ_INSTS_VALU'][0] * 4 / 1024.0,
                                                                     100.0 * sq['SQ_INSTS_VALU'][0] * 4 / 1024.0 / cyc))
            notes.append('%s: %s' % (w, {k: '%.4g' % v[0] for k, v in sq.items()}))
    traffic['_provenance'] = {'calibration_known_read_over_FETCH_SIZE': 1.9996481868971179,
                              'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --workload X '
                                      '--no-cpu-baseline` (tools/profile_round.sh); bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE '
                                      'reports half of wide streaming reads; calibrated in round 1 on to_particle_major reading a known 409.6 MB, '
                                      'profiles/r01/c2_pmc_fetch_size.csv); reduced by tools/reduce_profiles.py'}
    json.dump(traffic, open(traffic_path, 'w'), indent=1)
    print('\n'.join(notes))


if __name__ == '__main__':
    main()
