mkdir -p gpurun_out
LOG=gpurun_out/pytest_gpu4.log; : > $LOG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "train or loss_vs_oracle" >> $LOG 2>&1; echo "exit $?" >> $LOG
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -k "conv or model" >> $LOG 2>&1; echo "exit $?" >> $LOG
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_auto.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_auto.log
grep -E "passed|failed|exit|^E  |FAILED" $LOG | cut -c1-400 | tail -40
tail -c 1500 gpurun_out/bench_auto.log
