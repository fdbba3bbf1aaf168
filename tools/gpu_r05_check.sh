# round 5, first GPU call: the full -m gpu suite, smoke, the default bench line (own-output NMS pipeline, in-run train families), the 2-rank self-launch smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_gpu.log
grep -a "passed\|failed\|error" gpurun_out/r05_pytest_gpu.log | tail -5
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r05_bench_default_a.json 2> gpurun_out/r05_bench_default_a.err; echo "bench exit $?"; head -c 400 gpurun_out/r05_bench_default_a.json; echo
tail -5 gpurun_out/r05_bench_default_a.err
bash tools/gpu_dist_smoke.sh r05
