# round-final measurement set on one MI355X box (gpurun -- 'bash tools/gpu_final.sh [tag]'); copy what should be judged into profiles/
TAG=${1:-r04}
mkdir -p gpurun_out
R=$PWD
bash tools/gpu_train_prof.sh $TAG > gpurun_out/${TAG}_train_prof.log 2>&1
cp gpurun_out/${TAG}_train_groups.json profiles/${TAG}_train_step_kernel_groups.json   # bench.py attaches this file to its `train` object
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; head -c 250 gpurun_out/${TAG}_bench_default.json; echo
timeout 600 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_config4_spp.json 2> gpurun_out/bench_c4.err; head -c 200 gpurun_out/${TAG}_bench_config4_spp.json; echo
timeout 600 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_config5_1280_nc365_bf16.json 2> gpurun_out/bench_c5.err; head -c 200 gpurun_out/${TAG}_bench_config5_1280_nc365_bf16.json; echo
# configs[4], the second reading of "batch 8 over 8 GPUs" (SURVEY 8 header): one image per rank
timeout 600 python bench.py --imgsz 1280 --batch 1 --dtype bf16 --nc 365 --steps 20 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_config5_1280_nc365_bf16_batch1.json 2> gpurun_out/bench_c5b.err; head -c 200 gpurun_out/${TAG}_bench_config5_1280_nc365_bf16_batch1.json; echo
timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 > gpurun_out/${TAG}_bench_train_bs64.json 2> gpurun_out/final_train.err; head -c 200 gpurun_out/${TAG}_bench_train_bs64.json; echo
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/${TAG}_train_phases_bs64.json 2>&1; tail -1 gpurun_out/${TAG}_train_phases_bs64.json
timeout 300 python tools/train_layers.py > gpurun_out/${TAG}_train_layers_bs64.txt 2>&1
timeout 300 python bench.py --profile-layers --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > gpurun_out/${TAG}_layer_table_bs32.txt 2>&1; tail -3 gpurun_out/${TAG}_layer_table_bs32.txt | head -c 300; echo
bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc_run.log 2>&1
cp gpurun_out/pmc_summary_final.json gpurun_out/${TAG}_pmc_summary.json; cp gpurun_out/bench_kernel_stats_final.md gpurun_out/${TAG}_bench_kernel_stats.md
timeout 300 python tools/conv_lab.py --rounds 5 --reps 20 --batch 32 --only "L6.cv2,L8.cv2,L10.cv2,L13" --arms "conv_v10=0;v10_half=0;v10_half=1;v10_half=2" > gpurun_out/${TAG}_conv_lab_v10.txt 2>&1
timeout 300 python tools/conv_lab.py --rounds 5 --reps 10 --batch 64 --noact --only "L6.cv2,L8.cv2,L10.cv2" --arms "conv_v10=0;v10_half=0;v10_half=1;v10_half=2" >> gpurun_out/${TAG}_conv_lab_v10.txt 2>&1
