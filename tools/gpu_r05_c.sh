# round 5, third GPU call: the blocked greedy NMS (its tests + the default line), filter gradients on a side stream A/B (3 interleaved rounds)
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "nms or e2e or detect_batches or pipeline" > gpurun_out/r05_pytest_nms.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_nms.log
grep -a "passed\|failed" gpurun_out/r05_pytest_nms.log | tail -3
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05_bench_default_c.json 2> gpurun_out/r05_bench_default_c.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_default_c.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'sequential_images_per_sec_per_gpu', 'legs_ms', 'nms_candidates_per_image', 'synthetic_nms_tensor')})
print('train', {k: d['train'].get(k) for k in ('value', 'ms_per_step')})
PY
timeout 400 python tools/wgrad_overlap_ab.py --arms base,s --rounds 3 --steps 8 > gpurun_out/r05_wgrad_stream_ab.txt 2>&1; tail -8 gpurun_out/r05_wgrad_stream_ab.txt | cut -c1-300
