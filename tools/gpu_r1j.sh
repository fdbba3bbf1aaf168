mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "spp or layout or config5" 2>&1 | tail -8
for v in lds direct; do Y3_SPP=$v timeout 300 python bench.py --model yolov3-spp --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['legs_ms'], d['roofline']['whole_forward']['by_kernel'].get('pools'))"; done
