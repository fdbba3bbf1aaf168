mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "wgrad or train_step or gradients_vs" > gpurun_out/r3n_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3n_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3n_pytest.log | tail -5
bash tools/gpu_ab_train.sh
