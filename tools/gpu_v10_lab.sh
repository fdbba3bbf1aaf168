# conv_v10.h A/B on the benchmark shapes + the conv parity groups (one gpurun call)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv and not wgrad and not train_launches" > gpurun_out/v10_pytest.log 2>&1; tail -4 gpurun_out/v10_pytest.log
timeout 300 python tools/conv_lab.py --rounds 5 --reps 20 --batch 32 --only "L6.cv2,L8.cv2,L10.cv2,L13" --arms "conv_v10=0;conv_v10=1" > gpurun_out/v10_lab_bs32.txt 2>&1; grep -v amdgpu gpurun_out/v10_lab_bs32.txt
timeout 300 python tools/conv_lab.py --rounds 5 --reps 10 --batch 64 --noact --only "L6.cv2,L8.cv2,L10.cv2" --arms "conv_v10=0;conv_v10=1" > gpurun_out/v10_lab_bs64.txt 2>&1; grep -v amdgpu gpurun_out/v10_lab_bs64.txt
