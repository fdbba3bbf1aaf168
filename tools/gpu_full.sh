mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_full.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_full.log
tail -8 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; tail -c 1500 gpurun_out/bench_c5.json
timeout 600 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; tail -c 900 gpurun_out/bench_c4.json
