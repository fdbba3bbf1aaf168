mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r3h_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3h_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3h_pytest.log | tail -8
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --train-steps 4 > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r3h_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['legs_ms'], d.get('train',{}).get('value'), d['roofline']['kernel'], d['roofline']['frac'], {k: v['ms'] for k, v in d['roofline']['whole_forward']['by_kernel'].items()})"
