set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15
for i in 1 2; do timeout 120 python tools/c4_iter.py 1000000 20 15; done
MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 20 15
MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so timeout 120 python tools/c4_iter.py 1000000 20 15
timeout 120 python tools/c4_iter.py 125000 20 15
MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 125000 20 15
