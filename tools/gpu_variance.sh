# run-to-run spread of the default inference line on ONE box (two-stream schedule), with the HIP hardware-queue count varied
mkdir -p gpurun_out
for i in 1 2 3 4; do
  for q in default 2 8; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 120 python bench.py --no-train --no-cpu-baseline --no-clocks 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues=$q run=$i', d['value'], d['ms_per_step'], 'seq', d['sequential_images_per_sec_per_gpu'], 'fwd', d['legs_ms']['forward+decode'], 'frac', d['roofline']['frac'])"
  done
done | tee gpurun_out/bench_variance.txt
unset GPU_MAX_HW_QUEUES
for i in 1 2; do timeout 120 python bench.py --no-train --no-cpu-baseline --no-clocks --no-overlap 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-overlap run=$i', d['value'], d['ms_per_step'])"; done | tee -a gpurun_out/bench_variance.txt
