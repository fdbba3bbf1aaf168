# same-box A/B of conv_1x1s.h inside the whole forward and the whole train step (Y3_TUNE is read once per process): two interleaved rounds
mkdir -p gpurun_out
out=gpurun_out/r05_s1x1_whole_step_ab.txt
echo "# same box, interleaved: python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks   and   python bench.py --mode train --batch 64 --steps 10 --warmup 4" > $out
for r in 1 2; do
  for arm in "conv_1x1s=0" "conv_1x1s=1"; do
    Y3_TUNE=$arm timeout 300 python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['roofline']['whole_forward']
print('round $r $arm infer value', d['value'], 'img/s  forward+decode', d['legs_ms']['forward+decode'], 'ms  kernel_ms', w['kernel_ms'], ' 1x1 groups', {k:v['ms'] for k,v in w['by_kernel'].items() if '1x1' in k})" >> $out
    Y3_TUNE=$arm timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['kernel_groups']
print('round $r $arm train value', d['value'], 'img/s ', d['ms_per_step'], 'ms/step  conv fwd+dgrad', g['conv forward + data gradient (implicit GEMM)']['ms_per_step'], ' bn', g['bn / activation passes']['ms_per_step'], ' wgrad', g['wgrad (filter gradients)']['ms_per_step'])" >> $out
  done
done
cat $out
