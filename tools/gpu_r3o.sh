mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "conv_mfma or conv_baseline or wgrad_and_dgrad or model_fp32_vs or model_half or train_step_autocast" > gpurun_out/r3o_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3o_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3o_pytest.log | tail -5
bash tools/gpu_ab.sh yolov3_amd/lib/libyolov3_hip_old.so 2>&1 | cut -c1-330
bash tools/gpu_ab_train.sh
