mkdir -p gpurun_out
LOG=gpurun_out/pytest_gpu3.log; : > $LOG
for v in v3a v3b v3c; do
  echo "=== Y3_CONV=$v pytest -k conv" >> $LOG
  Y3_CONV=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -k "conv_mfma" >> $LOG 2>&1
  echo "exit $?" >> $LOG
done
echo "=== default pytest -k model,loss" >> $LOG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "model or loss or end_to_end" >> $LOG 2>&1; echo "exit $?" >> $LOG
for v in v2 v3a v3b v3c; do
  Y3_CONV=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-layers > gpurun_out/bench_$v.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$v.log
done
grep -E "passed|failed|exit|===" $LOG | tail -14
for v in v2 v3a v3b v3c; do echo "--- $v"; tail -2 gpurun_out/bench_$v.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], d['ms_per_step'], d['legs_ms'], r['kernel'], r['achieved'], r['frac'], r['whole_forward']['by_kernel_ms'])
    else: print(l.strip())
"; done
