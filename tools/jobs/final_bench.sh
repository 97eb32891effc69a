mkdir -p gpurun_out/r06
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06/bench_detail_final.json ) > gpurun_out/r06/bench_steps20_warmup5.json 2> gpurun_out/r06/bench_final.err
tail -4 gpurun_out/r06/bench_final.err; wc -c gpurun_out/r06/bench_steps20_warmup5.json
bash tools/profile_round.sh gpurun_out/r06prof_c4 c4 > gpurun_out/r06prof_c4.log 2>&1
tail -2 gpurun_out/r06prof_c4.log
( timeout 300 python tools/sweep_L.py 32 1000000 funnel ) > gpurun_out/r06/sweep_L_c4.txt 2>&1
( timeout 300 python tools/sweep_L.py 512 100000 iso ) > gpurun_out/r06/sweep_L_c2.txt 2>&1
cat gpurun_out/r06/sweep_L_c4.txt gpurun_out/r06/sweep_L_c2.txt
