# round-final measurement set on one MI355X box (gpurun -- 'bash tools/gpu_final.sh'); copy what should be judged into profiles/
mkdir -p gpurun_out
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_full.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_full.log
tail -5 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
# PMC passes first: bench.py reports roofline.traffic from profiles/r01_pmc_summary.json
bash tools/gpu_pmc.sh > gpurun_out/final_pmc.log 2>&1
cp gpurun_out/pmc_summary_final.json profiles/r01_pmc_summary.json
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; head -c 250 gpurun_out/final_bench.json; echo
timeout 900 python bench.py --no-overlap --no-cpu-baseline > gpurun_out/final_bench_sequential.json 2> gpurun_out/final_bench_seq.err; head -c 250 gpurun_out/final_bench_sequential.json; echo
timeout 900 python bench.py --no-cpu-baseline --profile-layers > gpurun_out/final_layers.log 2>&1
timeout 600 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; head -c 200 gpurun_out/bench_c4.json; echo
timeout 600 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; head -c 200 gpurun_out/bench_c5.json; echo
# training step (configs[2] per-GPU shape)
timeout 300 python bench.py --mode train --batch 64 --steps 5 --warmup 2 > gpurun_out/final_train.json 2> gpurun_out/final_train.err; head -c 200 gpurun_out/final_train.json; echo
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/final_train_phases.json 2>&1; tail -1 gpurun_out/final_train_phases.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof_train -o t -- python $R/tools/train_bench.py --batch 64 --steps 2 --fused > $R/gpurun_out/final_prof_train.log 2>&1
cd $R
python tools/kstats.py gpurun_out/final_prof_train "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 2 --fused (3 steps incl. warm-up), yolov3 640x640 autocast fp16" > gpurun_out/final_train_kstats.md 2>&1
rm -rf gpurun_out/final_prof_train
