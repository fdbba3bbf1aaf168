# elementwise BatchNorm passes: non-temporal forms + uncapped grid (Y3_BN_STREAM) and the data-gradient epilogue statistics (Y3_BNB_EPILOGUE), A/B on one box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "bn_backward or autocast or gradients_vs_oracle or stem_bn_bwd or epilogue_bn or train_forward" > gpurun_out/bns_pytest.log 2>&1; echo "exit $?" >> gpurun_out/bns_pytest.log
tail -6 gpurun_out/bns_pytest.log
run() { timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
  Y3_BN_STREAM=0 Y3_BNB_EPILOGUE=0 run "round1-form          "
  Y3_BN_STREAM=0 run "epilogue-stats only  "
  Y3_BNB_EPILOGUE=0 run "stream forms only    "
  run "both (default)       "
done
timeout 300 python tools/train_layers.py > gpurun_out/bns_train_layers.txt 2>&1
