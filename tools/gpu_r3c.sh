mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -k "wgrad_benchmark_shapes" > gpurun_out/r3c_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3c_pytest.log
grep -a "passed\|failed\|exit\|Error\|fault\|^\[wgrad" gpurun_out/r3c_pytest.log | tail -12
if grep -aq "fault" gpurun_out/r3c_pytest.log; then
  AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=0 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -k "wgrad_benchmark_shapes and L6cv2" > gpurun_out/r3c_pytest_serial.log 2>&1
  grep -a "passed\|failed\|fault\|line" gpurun_out/r3c_pytest_serial.log | head -8
fi
timeout 600 python tools/v7_ablate.py > gpurun_out/r3c_ablate.txt 2>&1; cat gpurun_out/r3c_ablate.txt | tail -24
