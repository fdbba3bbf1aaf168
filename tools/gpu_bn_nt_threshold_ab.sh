# threshold of the non-temporal BatchNorm forms against the 256 MB Infinity Cache: same-box train-step A/B
mkdir -p gpurun_out
out=gpurun_out/r05_bn_nt_threshold_ab.txt
echo "# same box, interleaved: Y3_TUNE=bn_nt_bytes=<bytes> python bench.py --mode train --batch 64 --steps 10 --warmup 4 (default 134217728)" > $out
for r in 1 2; do
  for v in 134217728 67108864 268435456 536870912 4000000000; do
    Y3_TUNE=bn_nt_bytes=$v timeout 200 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['kernel_groups']; print('round $r bn_nt_bytes $v :', d['value'], 'img/s ', d['ms_per_step'], 'ms/step  bn family', g['bn / activation passes']['ms_per_step'])" >> $out
  done
done
cat $out
