for n in 2 5 10 20 40 64; do echo -n "n=$n "; timeout 120 python tools/c4_iter.py 1000000 $n 15 | tail -1; done
echo NO_RELAY; for n in 5 20 64; do echo -n "n=$n "; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 $n 15 | tail -1; done
