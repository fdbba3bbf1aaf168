# last check of a round: full GPU suite + smoke + the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_full.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_full.log
grep -a "passed\|failed\|exit" gpurun_out/pytest_gpu_full.log | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; head -c 300 gpurun_out/final_bench.json; echo
