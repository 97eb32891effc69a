cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sicv
for v in ${VARIANTS:-t1 t2 t3 t4 t5 t6}; do
  export MJHMC_HIP_LIB=$R/mjhmc_amd/lib/libsic_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sicv/v$v -o v$v -- python3 $R/tools/sic_leap_time.py 4 25 > $R/gpurun_out/sicv/v$v.log 2>&1
  f=$(find $R/gpurun_out/sicv/v$v -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "v$v: $(grep sic_leap "$f" | cut -d, -f2-4)"; else echo "v$v: no kernel stats (see v$v.log)"; fi
done
