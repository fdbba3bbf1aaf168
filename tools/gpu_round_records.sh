# the records of a round on its final tree, after tools/gpu_verify_round.sh: the other BASELINE configs (inference lines of configs[3] / [4], train steps of the other
# models), the two-rank plumbing smoke, the PMC summary behind bench.py's `traffic` / `mfma_busy_frac_pmc`.  bash tools/gpu_round_records.sh [tag, default r06]
T=${1:-r06}
mkdir -p gpurun_out
timeout 600 python bench.py --model yolov3-spp --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/${T}_bench_config4_spp.json 2>/dev/null; echo "config4 exit $?"
timeout 600 python bench.py --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/${T}_bench_config5_1280.json 2>/dev/null; echo "config5 exit $?"
for m in yolov3-spp yolov3-tiny; do
  Y3_NO_EXCHANGE_LEG=1 timeout 600 python bench.py --mode train --batch 64 --model $m --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${T}_train_$m.json
done
Y3_NO_EXCHANGE_LEG=1 timeout 600 python bench.py --mode train --imgsz 1280 --batch 8 --dtype bf16 --nc 365 --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${T}_train_config5_1280.json
python - <<PY
import json
for f in ("bench_config4_spp", "bench_config5_1280", "train_yolov3-spp", "train_yolov3-tiny", "train_config5_1280"):
    try:
        d = json.loads(open("gpurun_out/${T}_%s.json" % f).read().strip().splitlines()[-1]); print(f, d["value"], d["unit"], d["ms_per_step"])
    except Exception as e: print(f, "NO JSON", e)
PY
bash tools/gpu_dist_smoke.sh $T
bash tools/gpu_pmc.sh
