run() { Y3_NO_EXCHANGE_LEG=1 Y3_TUNE=$1 timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
for a in wgrad_blocks=512 bn_nt_bytes=33554432 bn_nt_bytes=67108864 bn_nt_bytes=268435456 bn_nt_bytes=4611686018427387904 v10_half=0 v10_half=1 wgrad_xcd=0 wgrad_xcd=1 wgrad_xcd=3 conv_ahead=2 v10_group=0; do run $a; done
done
