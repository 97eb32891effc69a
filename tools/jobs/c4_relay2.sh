timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -5
echo RELAY; timeout 200 python tools/sweep_L.py 32 1000000 funnel
for i in 1 2; do timeout 120 python tools/c4_iter.py 1000000 20 15; done
echo NO_RELAY; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 20 15
echo shard8; timeout 120 python tools/c4_iter.py 125000 20 15
MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 125000 20 15
