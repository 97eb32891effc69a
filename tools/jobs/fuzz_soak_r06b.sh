mkdir -p gpurun_out/r06b
( timeout 400 python tools/fuzz_rows.py 300 7 ) > gpurun_out/r06b/fuzz_rows.txt 2>&1
tail -2 gpurun_out/r06b/fuzz_rows.txt
( timeout 1200 python tools/soak_all.py 30 ) > gpurun_out/r06b/soak_all.txt 2>&1
tail -8 gpurun_out/r06b/soak_all.txt
( timeout 400 python tools/fuzz_dense.py 120 4 ) > gpurun_out/r06b/fuzz_dense.txt 2>&1
tail -2 gpurun_out/r06b/fuzz_dense.txt
