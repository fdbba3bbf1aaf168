timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "side_stream or train_step or multi_scale or gradient_exchange or map_parity" > gpurun_out/run_b.log 2>&1
grep -a "passed\|failed\|Error\|^E " gpurun_out/run_b.log | tail -15
