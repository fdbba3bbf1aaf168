mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "conv_v9" > gpurun_out/r3e_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3e_pytest.log
grep -a "passed\|failed\|exit\|Error\|outside\|assert" gpurun_out/r3e_pytest.log | tail -12
timeout 600 python tools/conv_lab.py --rounds 5 --reps 10 --only "cv2" > gpurun_out/r3e_lab.txt 2>&1; cat gpurun_out/r3e_lab.txt | tail -20
