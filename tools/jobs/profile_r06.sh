bash tools/profile_round.sh gpurun_out/r06prof > gpurun_out/r06prof.log 2>&1
tail -3 gpurun_out/r06prof.log
ls gpurun_out/r06prof | head -50
