mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "wgrad or loss" > gpurun_out/wx_pytest.log 2>&1; echo "exit $?" >> gpurun_out/wx_pytest.log
tail -4 gpurun_out/wx_pytest.log
run() { timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
  Y3_TUNE=wgrad_xcd=0 run "dispatch order          "
  Y3_TUNE=wgrad_xcd=1 run "xcd groups (128 tiles)  "
  Y3_TUNE=wgrad_xcd=2 run "xcd groups (all)        "
done
Y3_TUNE=wgrad_xcd=0 timeout 300 python tools/train_layers.py > gpurun_out/wx_layers_0.txt 2>&1
Y3_TUNE=wgrad_xcd=1 timeout 300 python tools/train_layers.py > gpurun_out/wx_layers_1.txt 2>&1
