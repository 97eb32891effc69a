timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
( timeout 300 python tools/fuzz_rows.py 150 21 ) 2>&1 | tail -1
( MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_FORCE_RELAY=1 timeout 300 python tools/fuzz_rows.py 150 22 ) 2>&1 | tail -1
echo "MM one-iteration calls: rows / group form"; export MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so
timeout 100 python tools/c4_iter.py 1000000 1 15 0.1 mm; MJHMC_NO_ROWS=1 timeout 100 python tools/c4_iter.py 1000000 1 15 0.1 mm
echo "MM L=5 20/launch: product choice / forced relay / group form"; timeout 100 python tools/c4_iter.py 1000000 20 5 0.1 mm; MJHMC_FORCE_RELAY=1 timeout 100 python tools/c4_iter.py 1000000 20 5 0.1 mm;  MJHMC_NO_ROWS=1 timeout 100 python tools/c4_iter.py 1000000 20 5 0.1 mm
echo "funnel L=5: product choice / forced relay"; timeout 100 python tools/c4_iter.py 1000000 20 5; MJHMC_FORCE_RELAY=1 timeout 100 python tools/c4_iter.py 1000000 20 5
