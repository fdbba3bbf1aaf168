# round 3, first GPU pass: full suite on the pruned library + A/B of the request depth (knob conv_ahead 3 vs 2) on one box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r3a_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3a_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3a_pytest.log | tail -8
for i in ; do
  for a in 2 3; do
    Y3_TUNE=conv_ahead=$a timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-clocks --train-steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead $a', d['value'], d['legs_ms'], d.get('train',{}).get('value'), {k: v['ms'] for k, v in d['roofline']['whole_forward']['by_kernel'].items()})"
  done
done
