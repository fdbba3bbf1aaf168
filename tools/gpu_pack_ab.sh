# source-indexed filter packing: its test, the train-step tests, then the packing time inside the train step (in-run family)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "pack_filter or train_step or multi_scale or two_outstanding or gradient_exchange_over" > gpurun_out/r05_pytest_pack.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_pack.log
grep -a "passed\|failed\|Error\|assert" gpurun_out/r05_pytest_pack.log | tail -6
timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['kernel_groups']; print('train:', d['value'], 'img/s ', d['ms_per_step'], 'ms/step  packing', g['filter packing'], ' final loss', d['final_loss'])"
