mkdir -p gpurun_out/r06c
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06c/bench_detail_final.json ) > gpurun_out/r06c/bench_steps20_warmup5.json 2> gpurun_out/r06c/bench_final.err
tail -3 gpurun_out/r06c/bench_final.err; wc -c gpurun_out/r06c/bench_steps20_warmup5.json
