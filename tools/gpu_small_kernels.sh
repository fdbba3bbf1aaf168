# float4 wgrad reduce, vectorised loss_obj backward / detect_raw_bwd: parity tests + train-step A/B on one box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "loss or wgrad or autocast or gradients_vs_oracle or train_forward" > gpurun_out/sk_pytest.log 2>&1; echo "exit $?" >> gpurun_out/sk_pytest.log
tail -5 gpurun_out/sk_pytest.log
run() { timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
  Y3_WGRAD_REDUCE=1 run "one-element wgrad reduce "
  run "default                  "
  Y3_WGRAD_STREAM=1 run "wgrad on a second stream "
done
bash tools/gpu_train_prof.sh sk > gpurun_out/sk_prof_head.log 2>&1; head -45 gpurun_out/sk_train_kstats.md | cut -c1-170
