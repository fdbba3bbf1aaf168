# BatchNorm-backward statistics in the data-gradient epilogue: parity tests, then the batch-64 train step with and without (A/B on one box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "bn_backward or autocast or gradients_vs_oracle or two_outstanding or rccl_one_rank" > gpurun_out/bnb_pytest.log 2>&1; echo "exit $?" >> gpurun_out/bnb_pytest.log
tail -15 gpurun_out/bnb_pytest.log
for i in 1 2; do
  Y3_BNB_EPILOGUE=0 timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('separate', d['value'], d['ms_per_step'], d['final_loss'])"
  timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('epilogue', d['value'], d['ms_per_step'], d['final_loss'])"
done
timeout 300 python tools/train_layers.py > gpurun_out/bnb_train_layers.txt 2>&1
