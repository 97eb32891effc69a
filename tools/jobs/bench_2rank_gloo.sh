mkdir -p gpurun_out/r06
( time MJHMC_BENCH_BACKEND=gloo MJHMC_BENCH_ONE_GPU=1 timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 --workload c4 --no-cpu-baseline --detail gpurun_out/r06/bench_2rank_detail.json ) > gpurun_out/r06/bench_2rank_gloo_c4.json 2> gpurun_out/r06/bench_2rank_gloo_c4.err
tail -5 gpurun_out/r06/bench_2rank_gloo_c4.err; head -c 1500 gpurun_out/r06/bench_2rank_gloo_c4.json
