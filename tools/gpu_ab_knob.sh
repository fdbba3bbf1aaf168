# A/B one run-time knob on ONE box, train step: Y3_TUNE=<A> vs the defaults, interleaved (boxes of the pool differ by several %)
#   bash tools/gpu_ab_knob.sh wgrad_strip=0
mkdir -p gpurun_out
A=${1:?knob=value of arm A}
run() { timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  Y3_NO_EXCHANGE_LEG=1 Y3_TUNE=$A run "A ($A)"
  Y3_NO_EXCHANGE_LEG=1 run "B (defaults)"
done
