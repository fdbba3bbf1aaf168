mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/f_prof -o t -- python $R/tools/train_bench.py --batch 64 --steps 1 --fused > $R/gpurun_out/f_prof.log 2>&1
cd $R
python tools/ktrace.py gpurun_out/f_prof > gpurun_out/f_trace.txt 2>&1
python tools/kstats.py gpurun_out/f_prof "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 1 --fused (2 steps incl. warm-up)" > gpurun_out/f_kstats.md 2>&1
rm -rf gpurun_out/f_prof
head -30 gpurun_out/f_kstats.md | cut -c1-150
