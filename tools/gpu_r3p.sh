mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "loss or outstanding or plans or rect_batches or train_step_autocast or map_parity" > gpurun_out/r3p_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3p_pytest.log
grep -a "passed\|failed\|exit\|Error\|assert" gpurun_out/r3p_pytest.log | tail -8
