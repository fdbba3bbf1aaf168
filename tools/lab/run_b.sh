timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "layer0" > gpurun_out/run_b.log 2>&1
grep -a "passed\|failed\|Error\|^E " gpurun_out/run_b.log | tail -15
R=$PWD; cd /tmp && export TMPDIR=/tmp
for arm in 1; do
  Y3_STEM_RECOMPUTE=$arm Y3_NO_EXCHANGE_LEG=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$arm -o kt -- python $R/bench.py --mode train --batch 64 --steps 3 --warmup 2 > /tmp/kt_$arm.log 2>&1
  cd $R; python tools/kstats.py /tmp/kt_$arm "arm $arm" 60 | grep -i "stem\|kernel |" ; cd /tmp
done
cd $R; bash tools/gpu_ab_env.sh Y3_STEM_RECOMPUTE=0
