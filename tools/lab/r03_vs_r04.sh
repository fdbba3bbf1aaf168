# same-box comparison of the round-3 tree (checked out under tools/lab/_r03 by hand, lab only) and this tree: inference line + train leg, alternating
R=$PWD
for i in 1 2; do
  for t in r03 r04; do
    if [ $t = r03 ]; then cd $R/tools/lab/_r03; else cd $R; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-clocks --train-steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$t', 'infer', d['value'], 'seq', d.get('sequential_images_per_sec_per_gpu'), 'fwd ms', d['legs_ms']['forward+decode'], 'dominant', r['kernel'], r['achieved'], 'kernel_ms', r['whole_forward']['kernel_ms'], '| train', d['train']['value'], d['train']['ms_per_step'])"
  done
done
