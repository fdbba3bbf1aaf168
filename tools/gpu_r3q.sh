mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "bneck or model_fp32_vs or model_half or stem_pair" > gpurun_out/r3q_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3q_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3q_pytest.log | tail -5
bash tools/gpu_ab.sh yolov3_amd/lib/libyolov3_hip_old.so 2>&1 | cut -c1-260
