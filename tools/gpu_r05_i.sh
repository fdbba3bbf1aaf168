# conv_1x1s.h with several filter tiles per pixel range (Cin = 512 / Cout = 512): parity, then A/B against v6
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "s1x1 or baseline_shapes" > gpurun_out/r05_pytest_s1x1_b.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_s1x1_b.log
grep -a "passed\|failed\|Error\|assert" gpurun_out/r05_pytest_s1x1_b.log | tail -8
timeout 200 python tools/conv_lab.py --rounds 3 --reps 20 --batch 32 --only "L8.cv1 512" --arms "conv_1x1s=0;conv_1x1s=1" > gpurun_out/r05_conv_lab_s1x1_c.txt 2>&1
timeout 200 python tools/conv_lab.py --rounds 3 --reps 10 --batch 64 --noact --only "L8.cv1 512,T L8.cv1" --arms "conv_1x1s=0;conv_1x1s=1" >> gpurun_out/r05_conv_lab_s1x1_c.txt 2>&1
cut -c1-150 gpurun_out/r05_conv_lab_s1x1_c.txt
