# round 5, second GPU call: the whole -m gpu suite (no -x), rocprofv3 --kernel-trace --stats of the default inference line (every kernel: the NMS chain on the model's own output), 2-rank smoke
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r05_pytest_gpu_b.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_gpu_b.log
grep -a "passed\|failed" gpurun_out/r05_pytest_gpu_b.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/r05_rocprof_stats.log 2>&1
cd $R
python tools/kstats.py gpurun_out/r05_stats "rocprofv3 --kernel-trace --stats : python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-clocks (MI355X; NMS on the model's own output)" 70 > gpurun_out/r05_bench_kernel_stats_b.md
rm -rf gpurun_out/r05_stats
head -50 gpurun_out/r05_bench_kernel_stats_b.md | cut -c1-170
bash tools/gpu_dist_smoke.sh r05
