export MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so
echo "one iteration per call, compacted path"; for i in 1 2; do timeout 120 python tools/c4_iter.py 1000000 1 15; done
echo "one iteration per call, fused relay"; for i in 1 2; do MJHMC_FUSE_ONE=1 timeout 120 python tools/c4_iter.py 1000000 1 15; done
echo "N/8"; timeout 120 python tools/c4_iter.py 125000 1 15; MJHMC_FUSE_ONE=1 timeout 120 python tools/c4_iter.py 125000 1 15
