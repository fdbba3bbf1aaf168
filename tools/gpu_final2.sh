mkdir -p gpurun_out
R=$PWD
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "detect_batches or autoshape or end_to_end" 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 500 gpurun_out/final_bench.json
timeout 900 python bench.py --no-overlap --no-cpu-baseline > gpurun_out/final_bench_sequential.json 2> gpurun_out/final_bench_seq.err; head -c 300 gpurun_out/final_bench_sequential.json
timeout 300 python bench.py --mode train --batch 64 --steps 5 --warmup 2 > gpurun_out/final_train.json 2> gpurun_out/final_train.err; head -c 200 gpurun_out/final_train.json
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/final_train_phases.json 2>&1; tail -1 gpurun_out/final_train_phases.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof_train -o t -- python $R/tools/train_bench.py --batch 64 --steps 2 --fused > $R/gpurun_out/final_prof_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof_bench -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final_prof_bench.log 2>&1
cd $R
python tools/kstats.py gpurun_out/final_prof_train "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 2 --fused (3 steps incl. warm-up), yolov3 640x640 autocast fp16" > gpurun_out/final_train_kstats.md 2>&1
python tools/kstats.py gpurun_out/final_prof_bench "rocprofv3 --kernel-trace --stats : python bench.py --steps 5 --warmup 2 --no-cpu-baseline (MI355X)" > gpurun_out/bench_kernel_stats_final.md 2>&1
rm -rf gpurun_out/final_prof_train gpurun_out/final_prof_bench
head -8 gpurun_out/bench_kernel_stats_final.md | cut -c1-140
