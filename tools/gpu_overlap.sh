# filter gradients on a side stream (order / lag / priority / CU mask): A/B of the batch-64 train step (one process, interleaved), the CU-mask probe, then the
# training parity tests in the mode named by MODE_ENV
mkdir -p gpurun_out
timeout 300 python tools/wgrad_overlap_ab.py ${ARMS:+--arms "$ARMS"} --rounds 2 > gpurun_out/wgrad_overlap_ab.txt 2>&1
grep -a "round\|best" gpurun_out/wgrad_overlap_ab.txt | tail -24
timeout 120 python tools/cu_mask_probe.py > gpurun_out/cu_mask_probe.txt 2>&1; tail -14 gpurun_out/cu_mask_probe.txt
env ${MODE_ENV:-Y3_WGRAD_STREAM=1 Y3_WGRAD_PRIO=1 Y3_WGRAD_ORDER=1 Y3_WGRAD_LAG=4} timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "train_step_gradients_vs_oracle or train_step_autocast or rccl_one_rank or two_outstanding or train_step_640 or exchange_two_ranks" > gpurun_out/overlap_pytest.log 2>&1; echo "exit $?" >> gpurun_out/overlap_pytest.log
grep -a "passed\|failed\|exit" gpurun_out/overlap_pytest.log | tail -3
