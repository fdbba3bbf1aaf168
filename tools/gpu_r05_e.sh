# round 5: conv_1x1s.h -- parity tests of every instantiation, then same-box A/B against the tile kernels at the benchmark's shapes
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "s1x1" > gpurun_out/r05_pytest_s1x1.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_s1x1.log
grep -a "passed\|failed\|Error\|assert" gpurun_out/r05_pytest_s1x1.log | tail -12
timeout 200 python tools/conv_lab.py --rounds 3 --reps 20 --batch 32 --only "L6.cv1 256,L26.cv1,L4.cv1 128" --arms "conv_1x1s=0;conv_1x1s=1" > gpurun_out/r05_conv_lab_s1x1.txt 2>&1
timeout 200 python tools/conv_lab.py --rounds 3 --reps 10 --batch 64 --noact --only "L6.cv1 256,L26.cv1,L4.cv1 128,T L6.cv1,T L4.cv1" --arms "conv_1x1s=0;conv_1x1s=1" >> gpurun_out/r05_conv_lab_s1x1.txt 2>&1
cut -c1-150 gpurun_out/r05_conv_lab_s1x1.txt
