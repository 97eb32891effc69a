T=$PWD/mjhmc_amd/lib/libmjhmc_hip_test.so
for rep in 1 2 3; do
MJHMC_HIP_LIB=$T timeout 300 python tools/call_overhead.py 2>&1 | grep "wall per call" | sed 's/^/events    /'
MJHMC_HIP_LIB=$T MJHMC_NO_EVENTS=1 timeout 300 python tools/call_overhead.py 2>&1 | grep "wall per call" | sed 's/^/no events /'
done
