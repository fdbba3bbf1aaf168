# round 3: the new bench-shape training tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -k "bn_passes or train_step_640 or lost_handoff" > gpurun_out/r3b_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3b_pytest.log
grep -a "passed\|failed\|exit\|Error\|^\[wgrad\|^\[train" gpurun_out/r3b_pytest.log | tail -30
for args in "4 80 128 256 2" "16 80 128 256 2" "32 80 128 256 2" "64 80 128 256 2" "64 80 128 256 0" "64 40 256 512 2" "64 20 512 1024 2"; do
  timeout 120 python tools/lab/wgrad_fault.py $args 2>&1 | grep -a "plan\|ok\|fault" | tr '\n' ' '; echo
done
