mkdir -p gpurun_out
LOG=gpurun_out/pytest_gpu16.log; : > $LOG
for v in v5a v5b v5c; do
  echo "=== Y3_CONV=$v pytest -k conv" >> $LOG
  Y3_CONV=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -k "conv_mfma or model_half" >> $LOG 2>&1
  echo "exit $?" >> $LOG
done
grep -E "passed|failed|exit|===" $LOG | tail -12
for v in auto v5a v5b v5c; do
  Y3_CONV=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-layers > gpurun_out/bench_$v.log 2>&1
done
for v in auto v5a v5b v5c; do echo "--- $v"; grep -E "^\{" gpurun_out/bench_$v.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['value'], d['ms_per_step'], d['legs_ms']['forward+decode'])
"; done
for l in "L5 " "L6.0.cv2" "L7 " "L8.0.cv2" "L9 " "L10.0.cv2" "L13 " "L19.cv2" "L22 " "L26.cv2" "L8.0.cv1" "L10.0.cv1" "L12 " "L19.cv1"; do echo -n "$l: "; for v in auto v5a v5b v5c; do grep -E "^ +$l" gpurun_out/bench_$v.log | head -1 | awk '{printf "%s ms %s TF | ", $2, $3}'; done; echo; done
