mkdir -p gpurun_out/r06
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/pytest_gpu.log 2>&1
tail -15 gpurun_out/r06/pytest_gpu.log
