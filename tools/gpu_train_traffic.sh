# (1) the conv / v10 tests on the final library, (2) HBM-side traffic of the batch-64 train step per kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
mkdir -p gpurun_out
R=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "v10 or baseline_shapes or train_launches or golden" > gpurun_out/v10_refactor_pytest.log 2>&1; echo "exit $?" >> gpurun_out/v10_refactor_pytest.log
grep -a "passed\|failed\|exit" gpurun_out/v10_refactor_pytest.log | tail -3
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trpmc; mkdir -p $R/gpurun_out/trpmc
for c in FETCH_SIZE WRITE_SIZE; do
  Y3_NO_EXCHANGE_LEG=1 timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/trpmc/pmc_$c -o pmc -- python $R/bench.py --mode train --batch 64 --steps 2 --warmup 1 > $R/gpurun_out/trpmc_$c.log 2>&1; echo "exit $?" >> $R/gpurun_out/trpmc_$c.log
done
cd $R && python tools/pmc_traffic.py gpurun_out/trpmc 3 > gpurun_out/train_pmc_traffic.txt 2> gpurun_out/train_pmc_traffic.err; head -30 gpurun_out/train_pmc_traffic.txt; tail -2 gpurun_out/train_pmc_traffic.err
rm -rf gpurun_out/trpmc
