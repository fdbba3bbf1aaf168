mkdir -p gpurun_out
Y3_CONV_SMALL=v3 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -k "conv_mfma or model_half" 2>&1 | tail -3
for v in base v3; do
  Y3_CONV_SMALL=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-layers > gpurun_out/bench_small_$v.log 2>&1
done
for v in base v3; do echo "--- $v"; grep -E "^\{" gpurun_out/bench_small_$v.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['legs_ms']['forward+decode'])
"; done
for l in "L0 " "L1 " "L2.cv1" "L2.cv2" "L3 " "L4.0.cv1" "L4.0.cv2" "L7 " "L13 "; do echo -n "$l: "; for v in base v3; do grep -E "^ +$l" gpurun_out/bench_small_$v.log | head -1 | awk '{printf "%s ms %s TF %s GB/s | ", $2, $3, $4}'; done; echo; done
