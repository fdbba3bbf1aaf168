# threshold of the non-temporal BatchNorm passes (knob bn_nt_bytes), A/B of the batch-64 train step on one box
mkdir -p gpurun_out
run() { timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
  Y3_TUNE=bn_nt_bytes=4611686018427387904 run "plain forms only "
  Y3_TUNE=bn_nt_bytes=134217728 run "nt >= 128 MB     "
  Y3_TUNE=bn_nt_bytes=314572800 run "nt >= 300 MB     "
  Y3_TUNE=bn_nt_bytes=629145600 run "nt >= 600 MB     "
  Y3_TUNE=bn_nt_bytes=67108864 run "nt >= 64 MB      "
done
