mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "two_ranks or train_step_640 or wgrad_benchmark" > gpurun_out/r3k_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3k_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3k_pytest.log | tail -5
timeout 300 python tools/prof_train_ops.py > gpurun_out/r3k_train_ops.txt 2>&1; head -45 gpurun_out/r3k_train_ops.txt
timeout 300 python tools/train_bench.py --batch 64 --steps 4 --fused 2>&1 | tail -1
