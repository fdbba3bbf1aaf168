mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/d_prof -o t -- python $R/tools/train_bench.py --batch 64 --steps 1 --fused > $R/gpurun_out/d_prof.log 2>&1
cd $R
python tools/ktrace.py gpurun_out/d_prof > gpurun_out/d_trace.txt 2>&1
rm -rf gpurun_out/d_prof
wc -l gpurun_out/d_trace.txt
