mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "epilogue_bn or train" > gpurun_out/l_pytest.log 2>&1; echo "exit $?" >> gpurun_out/l_pytest.log
tail -8 gpurun_out/l_pytest.log
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/l_train_phases.log 2>&1; tail -1 gpurun_out/l_train_phases.log
Y3_BN_EPILOGUE=0 timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/l_train_phases_noepi.log 2>&1; tail -1 gpurun_out/l_train_phases_noepi.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/l_prof -o t -- python $R/tools/train_bench.py --batch 64 --steps 1 --fused > $R/gpurun_out/l_prof.log 2>&1
cd $R
python tools/ktrace.py gpurun_out/l_prof > gpurun_out/l_trace.txt 2>&1
python tools/kstats.py gpurun_out/l_prof "train step with epilogue stats" > gpurun_out/l_kstats.md 2>&1
rm -rf gpurun_out/l_prof
head -24 gpurun_out/l_kstats.md | cut -c1-150
