timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "maxpool_backward or (train and spp) or (train and tiny) or autograd_vs" > gpurun_out/run_b.log 2>&1
grep -a "passed\|failed\|Error\|^E " gpurun_out/run_b.log | tail -15
run() { timeout 300 python bench.py --mode train --model $2 --batch 64 --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step']); g=d['roofline']['kernel_groups']; print({k.split(' (')[0]: v['ms_per_step'] for k,v in g.items()})"; }
export Y3_NO_EXCHANGE_LEG=1
run "spp" yolov3-spp
run "tiny" yolov3-tiny
