mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "process_batch or scale_boxes or val_edge or train or wgrad" > gpurun_out/h_pytest.log 2>&1; echo "exit $?" >> gpurun_out/h_pytest.log
tail -12 gpurun_out/h_pytest.log
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/h_train_phases.log 2>&1; tail -1 gpurun_out/h_train_phases.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/h_prof -o t -- python $R/tools/train_bench.py --batch 64 --steps 1 --fused > $R/gpurun_out/h_prof.log 2>&1
cd $R
python tools/kstats.py gpurun_out/h_prof "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 1 --fused (2 steps incl. warm-up)" > gpurun_out/h_kstats.md 2>&1
rm -rf gpurun_out/h_prof
head -18 gpurun_out/h_kstats.md | cut -c1-150
