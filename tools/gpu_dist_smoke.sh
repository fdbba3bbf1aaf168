# two ranks on ONE GPU over gloo (RCCL refuses two ranks per device): the multi-process plumbing of `bench.py --gpus N` -- bench.py starting its own ranks
# (no torch.distributed.run wrapper: self_launch), rendezvous on 127.0.0.1, rank / device mapping, GradBuckets at world 2 (SUM + divide path on device
# tensors), barrier + max-over-ranks timing, the rank-0 JSON line with the appended train leg, the watchdog -- before the driver's first 8-GPU run.
# Outputs: gpurun_out/${TAG}_dist_smoke_{infer,train}.json (copy into profiles/).
TAG=${1:-r06}
mkdir -p gpurun_out
timeout 200 python3 bench.py --gpus 2 --train-timeout 90 --steps 3 --warmup 1 --batch 4 --train-batch 4 --train-steps 2 --dist-backend gloo --no-cpu-baseline --no-clocks \
  > gpurun_out/${TAG}_dist_smoke_infer.json 2> gpurun_out/dist_smoke_infer.err; echo "infer exit $?"
tail -c 600 gpurun_out/${TAG}_dist_smoke_infer.json | cut -c1-600; echo
timeout 150 python3 bench.py --gpus 2 --mode train --steps 2 --warmup 1 --batch 4 --dist-backend gloo > gpurun_out/${TAG}_dist_smoke_train.json 2> gpurun_out/dist_smoke_train.err; echo "train exit $?"
python3 -c "
import json
for f in ('gpurun_out/${TAG}_dist_smoke_infer.json', 'gpurun_out/${TAG}_dist_smoke_train.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, 'n_gpus', d['n_gpus'], 'world', d['process_group']['world_size'], d['value'], d['unit'], (d.get('train') or {}).get('value'), (d.get('train') or {}).get('error'))
    except Exception as e: print(f, 'NO JSON', e); print(open('gpurun_out/dist_smoke_' + ('infer' if 'infer' in f else 'train') + '.err').read()[-1500:])
"
