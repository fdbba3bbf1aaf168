# round 5: conv_1x1s.h dispatched by default -- the whole -m gpu suite, A/B of the added shapes, the default bench line
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r05_pytest_gpu_f.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_gpu_f.log
grep -a "passed\|failed" gpurun_out/r05_pytest_gpu_f.log | tail -3
timeout 200 python tools/conv_lab.py --rounds 3 --reps 10 --batch 64 --noact --only "T L2.cv1,head 256,T L26.cv1" --arms "conv_1x1s=0;conv_1x1s=1" > gpurun_out/r05_conv_lab_s1x1_b.txt 2>&1; cut -c1-150 gpurun_out/r05_conv_lab_s1x1_b.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05_bench_default_f.json 2> gpurun_out/r05_bench_default_f.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_default_f.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'sequential_images_per_sec_per_gpu', 'legs_ms', 'synthetic_nms_tensor')})
print({k: v for k, v in d['roofline']['whole_forward']['by_kernel'].items()})
print('train', {k: d['train'].get(k) for k in ('value', 'ms_per_step')}, d['train']['roofline']['kernel_groups'])
PY
