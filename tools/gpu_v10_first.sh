# first GPU contact of conv_v10.h: parity cases, then the A/B against v9 / the v3 tiles on the benchmark shapes (batch 32 inference, batch 64 training forward)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "v10 or fragment_copy" > gpurun_out/v10_pytest.log 2>&1; tail -15 gpurun_out/v10_pytest.log
timeout 300 python tools/conv_lab.py --rounds 5 --reps 20 --batch 32 --only "L6.cv2,L8.cv2,L10.cv2,L13" --arms "conv_v10=0;conv_v10=1;conv_v10=1,v10_mp=7;conv_v10=1,v10_mp=6" > gpurun_out/v10_lab_bs32.txt 2>&1; cat gpurun_out/v10_lab_bs32.txt
timeout 300 python tools/conv_lab.py --rounds 5 --reps 10 --batch 64 --noact --only "L6.cv2,L8.cv2,L10.cv2" --arms "conv_v10=0;conv_v10=1" > gpurun_out/v10_lab_bs64.txt 2>&1; cat gpurun_out/v10_lab_bs64.txt
