timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nms" 2>&1 | tail -5
for i in 1 2 3; do
for v in 0 1; do Y3_TUNE=nms_sort=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nms_sort=$v', d['value'], d['ms_per_step'], d['legs_ms'], d['sequential_images_per_sec_per_gpu'], d.get('sustained_images_per_sec'))"; done
done
