# round-final measurement set: PMC passes -> profiles json (so bench reports traffic), default bench (with cpu baseline), train bench, kernel stats
mkdir -p gpurun_out
R=$PWD
bash tools/gpu_pmc.sh > gpurun_out/final_pmc.log 2>&1
cp gpurun_out/pmc_summary_final.json profiles/r01_pmc_summary.json
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 600 gpurun_out/final_bench.json
timeout 300 python bench.py --mode train --batch 64 --steps 5 --warmup 2 > gpurun_out/final_train.json 2> gpurun_out/final_train.err; tail -c 400 gpurun_out/final_train.json
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/final_train_phases.json 2>&1; tail -1 gpurun_out/final_train_phases.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof_train -o t -- python $R/tools/train_bench.py --batch 64 --steps 2 --fused > $R/gpurun_out/final_prof_train.log 2>&1
cd $R
python tools/kstats.py gpurun_out/final_prof_train "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 2 --fused (3 steps incl. warm-up), yolov3 640x640 autocast fp16" > gpurun_out/final_train_kstats.md 2>&1
rm -rf gpurun_out/final_prof_train
head -8 gpurun_out/final_train_kstats.md | cut -c1-140
