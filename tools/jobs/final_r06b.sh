# end of round 6 (second session): the GPU suite, smoke, the driver's bench line, then the rocprofv3 evidence of every workload
mkdir -p gpurun_out/r06b
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06b/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/r06b/pytest_gpu.log | tail -2
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06b/bench_detail_final.json ) > gpurun_out/r06b/bench_steps20_warmup5.json 2> gpurun_out/r06b/bench_final.err
tail -3 gpurun_out/r06b/bench_final.err; wc -c gpurun_out/r06b/bench_steps20_warmup5.json
bash tools/profile_round.sh gpurun_out/r06b_prof > gpurun_out/r06b_prof.log 2>&1
tail -2 gpurun_out/r06b_prof.log
