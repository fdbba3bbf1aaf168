# A/B one environment switch of the Python engine on ONE box, train step: <VAR=value> vs the defaults, interleaved
#   bash tools/gpu_ab_env.sh Y3_DEFER_SHORTCUT=0
mkdir -p gpurun_out
A=${1:?VAR=value of arm A}
run() { timeout 300 python bench.py --mode train --batch 64 --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  env Y3_NO_EXCHANGE_LEG=1 $A bash -c "$(declare -f run); run 'A ($A)'"
  Y3_NO_EXCHANGE_LEG=1 run "B (defaults)"
done
