mkdir -p gpurun_out/r06d
( timeout 1500 python tools/soak_all.py 120 ) > gpurun_out/r06d/soak_all_120s.txt 2>&1
tail -8 gpurun_out/r06d/soak_all_120s.txt
( timeout 700 python tools/fuzz_rows.py 600 11 ) > gpurun_out/r06d/fuzz_rows_600s.txt 2>&1
tail -1 gpurun_out/r06d/fuzz_rows_600s.txt
( timeout 400 python tools/fuzz_dense.py 300 12 ) > gpurun_out/r06d/fuzz_dense_300s.txt 2>&1
tail -1 gpurun_out/r06d/fuzz_dense_300s.txt
