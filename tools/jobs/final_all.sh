mkdir -p gpurun_out/r06
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/pytest_gpu.log 2>&1
tail -4 gpurun_out/r06/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06/bench_detail_final.json ) > gpurun_out/r06/bench_steps20_warmup5.json 2> gpurun_out/r06/bench_final.err
tail -4 gpurun_out/r06/bench_final.err; wc -c gpurun_out/r06/bench_steps20_warmup5.json
bash tools/profile_round.sh gpurun_out/r06prof_c4 c4 > gpurun_out/r06prof_c4.log 2>&1
tail -2 gpurun_out/r06prof_c4.log
