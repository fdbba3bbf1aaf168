run() { Y3_TUNE=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['legs_ms']['forward+decode'])"; }
for i in 1 2; do
for a in nms_sort=1 v10_half=0 v10_half=1 v10_group=0 tile_xcd=0 conv_ahead=2 v10_defer=1; do run $a; done
done
