MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/librows_stamps.so timeout 200 python tools/rows_stamps.py 1000000 15 20 | tail -16
