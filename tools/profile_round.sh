#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline fields on the GPU box (run through gpurun):
#   kernel-trace statistics of every workload, and the HBM / SQ counter passes -- each in its OWN run, counters never
#   combined with tracing domains other than --kernel-trace (MI355X_MICROARCH.md, HBM / rocprofv3 section).
# usage: tools/profile_round.sh OUTDIR [workloads...]      (OUTDIR under gpurun_out/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1; shift
WL=${*:-c2 c3 c3f64 c4 c5 c5bf16}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  args="--workload $w --no-cpu-baseline --shard-of 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$w" -o $w -- python3 "$ROOT/bench.py" $args > "$OUT/bench_$w.json" 2> "$OUT/kt_$w.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${w}_$c" -o $w -- python3 "$ROOT/bench.py" $args > /dev/null 2> "$OUT/pmc_${w}_$c.err"
  done
done
for w in c2 c3f64 c4 c5 c5bf16; do
  case " $WL " in *" $w "*) ;; *) continue;; esac
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/sq1_$w" -o $w -- python3 "$ROOT/bench.py" --workload $w --no-cpu-baseline --shard-of 1 > /dev/null 2> "$OUT/sq1_$w.err"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2_$w" -o $w -- python3 "$ROOT/bench.py" --workload $w --no-cpu-baseline --shard-of 1 > /dev/null 2> "$OUT/sq2_$w.err"
done
# the one-iteration-per-launch form of C2 (the HBM-bound kernel the fused launch is measured against)
case " $WL " in *" c2 "*)
  # --steps 1: a call of ONE iteration is never fused (what every sampling_iteration() caller gets)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_c2nofuse" -o c2nofuse -- python3 "$ROOT/bench.py" --workload c2 --steps 1 --no-cpu-baseline --shard-of 1 > "$OUT/bench_c2nofuse.json" 2> "$OUT/kt_c2nofuse.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_c2nofuse_$c" -o c2nofuse -- python3 "$ROOT/bench.py" --workload c2 --steps 1 --no-cpu-baseline --shard-of 1 > /dev/null 2> "$OUT/pmc_c2nofuse_$c.err"
  done;;
esac
# ... and of C4 (trajectory launch in row form + jump-process launch)
case " $WL " in *" c4 "*)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_c4nofuse" -o c4nofuse -- python3 "$ROOT/bench.py" --workload c4 --steps 1 --no-cpu-baseline --shard-of 1 > "$OUT/bench_c4nofuse.json" 2> "$OUT/kt_c4nofuse.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_c4nofuse_$c" -o c4nofuse -- python3 "$ROOT/bench.py" --workload c4 --steps 1 --no-cpu-baseline --shard-of 1 > /dev/null 2> "$OUT/pmc_c4nofuse_$c.err"
  done;;
esac
# keep what is small: drop raw per-dispatch traces over 8 MB
find "$OUT" -name "*.csv" -size +8M -delete
du -sh "$OUT"
