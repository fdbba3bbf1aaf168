# full GPU suite + smoke + the default bench line (the driver's round-end sequence): bash tools/gpu_verify_round.sh [tag, default r06]
T=${1:-r06}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_pytest_gpu_full.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/${T}_pytest_gpu_full.log
grep -a "passed\|failed" gpurun_out/${T}_pytest_gpu_full.log | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_style.json 2> gpurun_out/${T}_bench_driver_style.err; echo "bench exit $?"
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_driver_style.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'sustained_images_per_sec', 'legs_ms', 'synthetic_nms_tensor')}); print('roofline frac', d['roofline']['frac'], d['roofline'].get('mfma_busy_frac_pmc'), d['roofline']['traffic_source'])
print('train', {k: d['train'].get(k) for k in ('value', 'ms_per_step')}, d['train'].get('roofline', {}).get('kernel_groups'))
PY
