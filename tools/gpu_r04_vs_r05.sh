# same box: the round-4 tree (git archive 89af238 in _r04tree/, its own library) against this tree -- two interleaved rounds of the inference line (r04's `value` is the
# synthetic-tensor NMS form: compared with r05's `synthetic_nms_tensor.images_per_sec`; r05's own-output `value` beside it) and of the batch-64 train step
# prepare once, here (the tree is git-ignored): mkdir _r04tree && git archive 89af238 | tar -x -C _r04tree && (cd _r04tree && python -c 'import __graft_entry__ as g; g.build()')
mkdir -p gpurun_out
[ -d _r04tree ] || { echo '_r04tree/ missing: see the line above'; exit 1; }
out=gpurun_out/r05_same_box_r04_vs_r05.txt
echo "# same box, interleaved; infer: python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks; train: python bench.py --mode train --batch 64 --steps 10 --warmup 4" > $out
R=$PWD
for r in 1 2; do
  (cd _r04tree && timeout 300 python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r r04 infer: value (NMS on the synthetic tensor)', d['value'], 'img/s  forward+decode', d['legs_ms']['forward+decode'], 'ms  dominant-group frac', d['roofline']['frac'])" >> $out
  timeout 300 python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r r05 infer: value (NMS on the model output)', d['value'], 'img/s  synthetic-tensor form', d['synthetic_nms_tensor']['images_per_sec'], 'img/s  forward+decode', d['legs_ms']['forward+decode'], 'ms  dominant-group frac', d['roofline']['frac'])" >> $out
  (cd _r04tree && timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r r04 train:', d['value'], 'img/s ', d['ms_per_step'], 'ms/step')" >> $out
  timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['kernel_groups']; print('round $r r05 train:', d['value'], 'img/s ', d['ms_per_step'], 'ms/step   families (in-run):', {k.split(' (')[0]: v['ms_per_step'] for k, v in g.items()})" >> $out
done
cat $out
