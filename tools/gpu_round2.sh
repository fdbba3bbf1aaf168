mkdir -p gpurun_out
LOG=gpurun_out/pytest_gpu2.log; : > $LOG
for k in conv nms loss model; do
  echo "=== pytest -k $k" >> $LOG
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "$k" >> $LOG 2>&1
  echo "exit $?" >> $LOG
done
nproc > gpurun_out/host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/host.txt 2>&1; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> gpurun_out/host.txt
timeout 900 python bench.py --steps 10 --warmup 2 --profile-layers > gpurun_out/bench_v2.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_v2.log
Y3_CONV_V1=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v1.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_v1.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
cd $R
find gpurun_out/prof_r1 -name "*kernel_trace*" -size +20M -delete
ls -la gpurun_out/prof_r1/* | head -20
grep -E "passed|failed|exit" $LOG | tail -12
tail -c 1500 gpurun_out/bench_v2.log; tail -c 900 gpurun_out/bench_v1.log
