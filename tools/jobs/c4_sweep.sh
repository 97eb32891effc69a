echo RELAY; timeout 200 python tools/sweep_L.py 32 1000000 funnel
echo NO_RELAY; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 200 python tools/sweep_L.py 32 1000000 funnel
