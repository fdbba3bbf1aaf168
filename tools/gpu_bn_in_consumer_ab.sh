# consumer-side BatchNorm on the Bottleneck.cv1 layers (conv_1x1s.h IN form): kernel parity, the train-step tests, same-box A/B of the whole step
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "bn_in_consumer or s1x1 or train_step or train_forward or multi_scale or two_outstanding or map_parity" > gpurun_out/r05_pytest_bnin.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_bnin.log
grep -a "passed\|failed\|Error\|assert" gpurun_out/r05_pytest_bnin.log | tail -8
out=gpurun_out/r05_bn_in_consumer_ab.txt
echo "# same box, interleaved: Y3_BN_IN_CONSUMER=<0|1> python bench.py --mode train --batch 64 --steps 10 --warmup 4" > $out
for r in 1 2; do
  for v in 0 1; do
    Y3_BN_IN_CONSUMER=$v timeout 200 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline']['kernel_groups']; print('round $r Y3_BN_IN_CONSUMER=$v :', d['value'], 'img/s ', d['ms_per_step'], 'ms/step  final loss', d['final_loss'], ' conv fwd+dgrad', g['conv forward + data gradient (implicit GEMM)']['ms_per_step'], ' bn', g['bn / activation passes']['ms_per_step'], g['bn / activation passes']['calls_per_step'], 'calls')" >> $out
  done
done
cat $out
