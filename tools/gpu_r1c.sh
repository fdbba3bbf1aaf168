# current-state measurement: per-layer table, inference + train bench, training-step kernel stats
mkdir -p gpurun_out
R=$PWD
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-layers > gpurun_out/c_infer.log 2>&1; echo "exit $?" >> gpurun_out/c_infer.log
timeout 300 python bench.py --mode train --batch 64 --steps 5 --warmup 2 > gpurun_out/c_train.log 2>&1; echo "exit $?" >> gpurun_out/c_train.log
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/c_train_phases.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c_prof_train -o t -- python $R/tools/train_bench.py --batch 64 --steps 2 --fused > $R/gpurun_out/c_prof_train.log 2>&1
cd $R
python tools/kstats.py gpurun_out/c_prof_train "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 2 --fused (3 steps incl. warm-up)" > gpurun_out/c_train_kstats.md 2>&1
rm -rf gpurun_out/c_prof_train
tail -3 gpurun_out/c_infer.log | cut -c1-1500; tail -2 gpurun_out/c_train.log; tail -1 gpurun_out/c_train_phases.log; head -14 gpurun_out/c_train_kstats.md
