# batch-64 train step per v10_group setting (interleaved, one box)
mkdir -p gpurun_out
for rep in 1 2; do for v in 0 1; do
  Y3_NO_EXCHANGE_LEG=1 Y3_TUNE=v10_group=$v timeout 200 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v10_group=$v rep=$rep', d['value'], 'img/s', d['ms_per_step'], 'ms/step, loss', d['final_loss'])"
done; done | tee gpurun_out/v10_group_train_ab.txt
