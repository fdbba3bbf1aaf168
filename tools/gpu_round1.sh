mkdir -p gpurun_out
for k in conv layout decode nms model end_to_end; do
  echo "=== pytest -k $k" >> gpurun_out/pytest_gpu.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "$k" >> gpurun_out/pytest_gpu.log 2>&1
  echo "exit $?" >> gpurun_out/pytest_gpu.log
done
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 2 --profile-layers > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
grep -E "passed|failed|exit|error" gpurun_out/pytest_gpu.log | tail -30
tail -5 gpurun_out/smoke.log
tail -c 3000 gpurun_out/bench.log
