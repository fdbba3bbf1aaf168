# the "direct" form of the gradient exchange (parallel.GradBuckets: all-to-all + owner's sum + all-gather) on the 1-GPU box: the two-rank parity tests of both
# forms, then the world-2 train leg of bench.py (two ranks on one GPU over gloo) with each form -> gpurun_out/r05_direct_exchange.txt
mkdir -p gpurun_out
out=gpurun_out/r05_direct_exchange.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gradient_exchange" 2>&1 | grep -E "passed|failed|error" > $out
for form in all_reduce direct; do
  Y3_GRAD_EXCHANGE=$form timeout 150 python3 bench.py --gpus 2 --mode train --steps 3 --warmup 1 --batch 8 --dist-backend gloo 2> gpurun_out/direct_exchange_$form.err | python3 -c "
import json,sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$form', 'world', d['process_group']['world_size'], d['value'], d['unit'], d['ms_per_step'], 'ms', 'loss', d['final_loss'], d['config'].get('gradient_exchange'))
except Exception as e: print('$form NO JSON', e); print(open('gpurun_out/direct_exchange_$form.err').read()[-1500:])" >> $out
done
cat $out
