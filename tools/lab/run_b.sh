timeout 600 python tools/wgrad_lab.py --arms "wgrad_blocks=1024;wgrad_blocks=512;wgrad_blocks=256;wgrad_blocks=384;wgrad_blocks=640" --shapes L2cv1,L4cv1,L6cv1 --rounds 4 --reps 10 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
for v in 1024 512 256 384; do Y3_NO_EXCHANGE_LEG=1 Y3_TUNE=wgrad_blocks=$v timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_blocks=$v', d['value'], d['ms_per_step'])"; done
done
Y3_WGRAD_STREAM=0 timeout 600 python tools/train_layers.py --batch 64 --top 80 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_train_layers_bs64.txt
tail -5 gpurun_out/r06_train_layers_bs64.txt
