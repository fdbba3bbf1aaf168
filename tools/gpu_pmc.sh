# PMC passes (one counter group per run, --kernel-trace only: the pool refuses pmc + sys traces) + a kernel-stats run of the same command
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/pmc_$tag.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_$tag.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/rocprof_stats.log 2>&1
cd $R
grep -a '^{"metric"' gpurun_out/rocprof_stats.log | tail -1 > gpurun_out/pmc_bench_line.json
python tools/pmc_summary.py gpurun_out gpurun_out/pmc_summary_final.json gpurun_out/pmc_bench_line.json
python tools/kstats.py gpurun_out/pmc_stats "rocprofv3 --kernel-trace --stats : python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train (MI355X)" > gpurun_out/bench_kernel_stats_final.md
head -12 gpurun_out/bench_kernel_stats_final.md
rm -rf gpurun_out/pmc_*/ gpurun_out/pmc_stats
# ---- the same passes over the batch-64 train step (the filter-gradient and BatchNorm kernels): folded under "train_step" of the same summary
mkdir -p gpurun_out/train
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  Y3_NO_EXCHANGE_LEG=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/train/pmc_$tag -o pmc -- python $R/bench.py --mode train --batch 64 --steps 3 --warmup 2 > $R/gpurun_out/train/pmc_$tag.log 2>&1; echo "exit $?" >> $R/gpurun_out/train/pmc_$tag.log
done
Y3_NO_EXCHANGE_LEG=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/train/pmc_stats -o bench -- python $R/bench.py --mode train --batch 64 --steps 3 --warmup 2 > $R/gpurun_out/train/rocprof_stats.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/train gpurun_out/pmc_summary_train.json
python tools/kstats.py gpurun_out/train/pmc_stats "rocprofv3 --kernel-trace --stats : python bench.py --mode train --batch 64 --steps 3 --warmup 2 (MI355X)" 40 > gpurun_out/train_kernel_stats_final.md
python - <<'PY'
import json
a = json.load(open("gpurun_out/pmc_summary_final.json"))
a["train_step"] = json.load(open("gpurun_out/pmc_summary_train.json"))
json.dump(a, open("gpurun_out/pmc_summary_final.json", "w"), indent=1)
PY
rm -rf gpurun_out/train/pmc_*/ gpurun_out/train/pmc_stats
