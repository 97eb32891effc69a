mkdir -p gpurun_out/r06
( timeout 1200 python tools/soak_all.py 30 ) > gpurun_out/r06/soak_all.txt 2>&1
tail -12 gpurun_out/r06/soak_all.txt
( timeout 400 python tools/fuzz_dense.py 120 3 ) > gpurun_out/r06/fuzz_dense.txt 2>&1
tail -3 gpurun_out/r06/fuzz_dense.txt
