mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_dense_parity.py -x -q -k "streamed_download or split_launches_failure" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_state_ops.py tests/test_gpu_fused.py -x -q 2>&1 | tail -5
( time python bench.py --steps 20 --warmup 5 --detail gpurun_out/r06/bench_detail_a.json ) > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err
tail -3 gpurun_out/r06/bench_a.err
wc -c gpurun_out/r06/bench_a.json
