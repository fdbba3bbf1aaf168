# small-batch inference (where the K-split form runs): base library (Y3_LIB) vs this tree, alternating
A=${1:-yolov3_amd/lib/libyolov3_hip_base.so}
for b in ${BATCHES:-1 4}; do for i in 1 2; do
  Y3_LIB=$PWD/$A python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A bs$b', d['value'], d['legs_ms']['forward+decode'], {k: v['ms'] for k, v in list(d['roofline']['whole_forward']['by_kernel'].items())[:4]})"
  python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B bs$b', d['value'], d['legs_ms']['forward+decode'], {k: v['ms'] for k, v in list(d['roofline']['whole_forward']['by_kernel'].items())[:4]})"
done; done
