"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) to HBM
bytes per launch of the jump kernel and store them in profiles/hbm_traffic.json.

usage: python tools/reduce_pmc.py WORKLOAD FETCH.csv WRITE.csv ITERATIONS_PER_LAUNCH

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of wide streaming reads, calibrated in round 1 on a kernel reading a known 409.6 MB
(profiles/r01/c2_pmc_fetch_size.csv, factor 1.9996) -> reads = FETCH_SIZE * 1024 * 2."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, counter, kernel='mjhmc_jump_kernel'):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row['Kernel_Name'] and row['Counter_Name'] == counter:
                vals.append(float(row['Counter_Value']))
    if not vals:
        raise SystemExit('no %s rows for %s in %s' % (counter, kernel, path))
    return vals


def main():
    workload, fetch_csv, write_csv, ipl = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    kernel = sys.argv[5] if len(sys.argv) > 5 else 'mjhmc_jump_kernel'
    fetch = per_launch(fetch_csv, 'FETCH_SIZE', kernel)
    write = per_launch(write_csv, 'WRITE_SIZE', kernel)
    # the last launch of the run is the timed one (warm-up launches precede it)
    f_kib, w_kib = fetch[-1], write[-1]
    total = f_kib * 1024 * 2.0 + w_kib * 1024
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[workload] = {'bytes_per_launch': total, 'iterations_per_launch': ipl,
                      'FETCH_SIZE_KiB': f_kib, 'WRITE_SIZE_KiB': w_kib, 'read_correction': 2.0,
                      'launches_seen': len(fetch), 'sources': [os.path.basename(fetch_csv), os.path.basename(write_csv)]}
    json.dump(data, open(path, 'w'), indent=1)
    print(workload, 'HBM bytes per launch: %.4g (read %.4g, write %.4g), %d iterations per launch'
          % (total, f_kib * 2048, w_kib * 1024, ipl))


if __name__ == '__main__':
    main()
