timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode or golden or detect or autoshape or map_parity or model" 2>&1 | tail -4
run() { timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['legs_ms']['forward+decode'])"; }
for i in 1 2 3; do
  Y3_LIB=$PWD/yolov3_amd/lib/libyolov3_hip_base.so run "A (old)"
  run "B (new)"
done
