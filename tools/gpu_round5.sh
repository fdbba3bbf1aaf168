mkdir -p gpurun_out
LOG=gpurun_out/pytest_gpu5.log; : > $LOG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "wgrad or train" >> $LOG 2>&1; echo "exit $?" >> $LOG
timeout 900 python tools/train_bench.py --batch 16 --steps 3 > gpurun_out/train_bench.log 2>&1; echo "exit $?" >> gpurun_out/train_bench.log
Y3_WGRAD=direct timeout 900 python tools/train_bench.py --batch 4 --steps 1 > gpurun_out/train_bench_direct.log 2>&1; echo "exit $?" >> gpurun_out/train_bench_direct.log
grep -E "passed|failed|exit|^E  |FAILED" $LOG | cut -c1-300 | tail -30
tail -3 gpurun_out/train_bench.log; tail -3 gpurun_out/train_bench_direct.log
