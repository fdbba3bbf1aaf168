# round-final measurement set on one MI355X box (gpurun -- 'bash tools/gpu_final.sh'); copy what should be judged into profiles/
mkdir -p gpurun_out
R=$PWD
# PMC passes first: bench.py reports roofline.traffic from profiles/r02_pmc_summary.json
bash tools/gpu_pmc.sh > gpurun_out/final_pmc.log 2>&1
cp gpurun_out/pmc_summary_final.json profiles/r02_pmc_summary.json
bash tools/gpu_train_prof.sh final > gpurun_out/final_train_prof.log 2>&1
cp gpurun_out/final_train_groups.json profiles/r02_train_step_kernel_groups.json
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; head -c 250 gpurun_out/final_bench.json; echo
timeout 900 python bench.py --no-overlap --no-cpu-baseline --no-train > gpurun_out/final_bench_sequential.json 2> gpurun_out/final_bench_seq.err; head -c 250 gpurun_out/final_bench_sequential.json; echo
timeout 900 python bench.py --no-cpu-baseline --no-train --profile-layers > gpurun_out/final_layers.log 2>&1
timeout 600 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; head -c 200 gpurun_out/bench_c4.json; echo
timeout 600 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; head -c 200 gpurun_out/bench_c5.json; echo
timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 > gpurun_out/final_train.json 2> gpurun_out/final_train.err; head -c 200 gpurun_out/final_train.json; echo
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/final_train_phases.json 2>&1; tail -1 gpurun_out/final_train_phases.json
timeout 300 python tools/train_layers.py > gpurun_out/final_train_layers.txt 2>&1
