# round-5 measurement set on one MI355X box (gpurun -- 'bash tools/gpu_r05_final.sh'); copy what should be judged into profiles/
TAG=r05
mkdir -p gpurun_out
R=$PWD
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "default exit $?"; head -c 200 gpurun_out/${TAG}_bench_default.json; echo
bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc_run.log 2>&1
cp gpurun_out/pmc_summary_final.json gpurun_out/${TAG}_pmc_summary.json; cp gpurun_out/bench_kernel_stats_final.md gpurun_out/${TAG}_bench_kernel_stats.md
head -14 gpurun_out/${TAG}_bench_kernel_stats.md | cut -c1-160
bash tools/gpu_train_prof.sh $TAG > gpurun_out/${TAG}_train_prof.log 2>&1
cp gpurun_out/${TAG}_train_groups.json gpurun_out/${TAG}_train_step_kernel_groups.json; cp gpurun_out/${TAG}_train_kstats.md gpurun_out/${TAG}_train_step_bs64_kernel_stats.md
head -12 gpurun_out/${TAG}_train_step_bs64_kernel_stats.md | cut -c1-160
timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 3 > gpurun_out/${TAG}_bench_train_bs64.json 2> gpurun_out/final_train.err; head -c 200 gpurun_out/${TAG}_bench_train_bs64.json; echo
timeout 300 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_config4_spp.json 2> gpurun_out/bench_c4.err; head -c 200 gpurun_out/${TAG}_bench_config4_spp.json; echo
timeout 300 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_config5_1280_nc365_bf16.json 2> gpurun_out/bench_c5.err; head -c 200 gpurun_out/${TAG}_bench_config5_1280_nc365_bf16.json; echo
timeout 300 python bench.py --imgsz 1280 --batch 1 --dtype bf16 --nc 365 --steps 20 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_config5_1280_nc365_bf16_batch1.json 2> gpurun_out/bench_c5b.err; head -c 200 gpurun_out/${TAG}_bench_config5_1280_nc365_bf16_batch1.json; echo
timeout 200 python tools/train_layers.py > gpurun_out/${TAG}_train_layers_bs64.txt 2>&1
timeout 200 python bench.py --profile-layers --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > gpurun_out/${TAG}_layer_table_bs32.txt 2>&1
# HBM-side traffic of the train step per kernel (two PMC passes)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trpmc; mkdir -p $R/gpurun_out/trpmc
for c in FETCH_SIZE WRITE_SIZE; do
  Y3_NO_EXCHANGE_LEG=1 timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/trpmc/pmc_$c -o pmc -- python $R/bench.py --mode train --batch 64 --steps 2 --warmup 1 > $R/gpurun_out/trpmc_$c.log 2>&1; echo "exit $?" >> $R/gpurun_out/trpmc_$c.log
done
cd $R && python tools/pmc_traffic.py gpurun_out/trpmc 4 > gpurun_out/${TAG}_train_step_traffic_by_kernel.txt 2> gpurun_out/train_pmc_traffic.err; head -12 gpurun_out/${TAG}_train_step_traffic_by_kernel.txt | cut -c1-200
rm -rf gpurun_out/trpmc
