run() { timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
export Y3_NO_EXCHANGE_LEG=1
for i in 1 2; do
  run "defaults"
  Y3_WGRAD_STREAM=0 run "one stream"
  Y3_TUNE=wgrad_patch=0 run "wgrad_big"
  Y3_WGRAD_STREAM=0 Y3_TUNE=wgrad_patch=0 run "one stream + wgrad_big"
  (cd _prevtree && run "r05 tree")
done
