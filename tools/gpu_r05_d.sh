# round 5, fourth GPU call: the one-barrier block reduction of channel_reduce (BatchNorm tests + train-step tests + the train bench), tile-variant sweep of the HBM-bound 1x1 launches
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "bn_ or batchnorm or train_step or sync_bn or stem_bn" > gpurun_out/r05_pytest_bn.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_bn.log
grep -a "passed\|failed" gpurun_out/r05_pytest_bn.log | tail -3
timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 3 > gpurun_out/r05_bench_train_d.json 2> gpurun_out/r05_bench_train_d.err; echo "train exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_train_d.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step')}); print(d['roofline']['kernel_groups'])
PY
timeout 300 python tools/conv_lab.py --rounds 3 --reps 20 --batch 32 --only "L6.cv1,L26.cv1,L4.cv1,L8.cv1,L10.cv1" --arms "conv=0;conv=4;conv=5;conv=6;conv=15" > gpurun_out/r05_conv_lab_1x1.txt 2>&1; cat gpurun_out/r05_conv_lab_1x1.txt | cut -c1-150
