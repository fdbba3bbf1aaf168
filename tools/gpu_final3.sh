mkdir -p gpurun_out
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_full.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_full.log
tail -5 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/gpu_pmc.sh > gpurun_out/final_pmc.log 2>&1
cp gpurun_out/pmc_summary_final.json profiles/r01_pmc_summary.json
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; head -c 250 gpurun_out/final_bench.json; echo
timeout 900 python bench.py --no-overlap --no-cpu-baseline > gpurun_out/final_bench_sequential.json 2> gpurun_out/final_bench_seq.err; head -c 250 gpurun_out/final_bench_sequential.json; echo
timeout 900 python bench.py --no-cpu-baseline --profile-layers > gpurun_out/final_layers.log 2>&1
timeout 600 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; head -c 200 gpurun_out/bench_c4.json; echo
timeout 600 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; head -c 200 gpurun_out/bench_c5.json; echo
