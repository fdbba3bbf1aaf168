# A/B two builds of the library on ONE box, train step: Y3_LIB=<A> vs default, interleaved (boxes of the pool differ by several %)
mkdir -p gpurun_out
A=${1:-yolov3_amd/lib/libyolov3_hip_old.so}
run() { timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  Y3_NO_EXCHANGE_LEG=1 Y3_LIB=$PWD/$A run "A (old)"
  Y3_NO_EXCHANGE_LEG=1 run "B (new)"
done
