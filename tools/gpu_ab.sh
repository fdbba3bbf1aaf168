# A/B two builds of the library on ONE box (boxes differ by several %): Y3_LIB=<A> vs default, interleaved
mkdir -p gpurun_out
A=${1:-yolov3_amd/lib/libyolov3_hip_old.so}
for i in 1 2 3; do
  Y3_LIB=$PWD/$A python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A  ', d['value'], d['legs_ms'], {k: v['ms'] for k, v in d['roofline']['whole_forward']['by_kernel'].items()})"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B  ', d['value'], d['legs_ms'], {k: v['ms'] for k, v in d['roofline']['whole_forward']['by_kernel'].items()})"
done
