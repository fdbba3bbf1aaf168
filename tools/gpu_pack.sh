mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "pack_filter_jobs or autocast or gradients_vs_oracle or two_outstanding or rccl_one_rank or train_forward" > gpurun_out/pk_pytest.log 2>&1; echo "exit $?" >> gpurun_out/pk_pytest.log
tail -4 gpurun_out/pk_pytest.log
run() { timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['final_loss'])"; }
for i in 1 2 3; do
  Y3_PACK_JOBS=0 run "one pack launch per layer "
  run "one launch for all layers "
done
