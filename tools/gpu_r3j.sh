# round 3: rocprofv3 kernel stats + PMC passes of the default bench with v9 dispatched; full GPU suite not repeated here
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/pmc_$tag.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_$tag.log
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/rocprof_stats.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out gpurun_out/r03_pmc_summary.json
python tools/kstats.py gpurun_out/pmc_stats "rocprofv3 --kernel-trace --stats : python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train (MI355X, round 3)" > gpurun_out/r03_bench_kernel_stats.md
head -14 gpurun_out/r03_bench_kernel_stats.md
rm -rf gpurun_out/pmc_*/
timeout 300 python bench.py --profile-layers --no-cpu-baseline --no-train > gpurun_out/r03_layer_table_bs32.txt 2>&1; tail -3 gpurun_out/r03_layer_table_bs32.txt | cut -c1-300
