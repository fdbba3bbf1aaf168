bash tools/gpu_verify_round.sh r06b
bash tools/gpu_pmc.sh > gpurun_out/r06b_pmc.log 2>&1
tail -5 gpurun_out/r06b_pmc.log
