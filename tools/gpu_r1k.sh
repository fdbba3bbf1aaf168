mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "epilogue_bn or train or wgrad or conv_mfma" > gpurun_out/k_pytest.log 2>&1; echo "exit $?" >> gpurun_out/k_pytest.log
tail -15 gpurun_out/k_pytest.log
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/k_train_phases.log 2>&1; tail -1 gpurun_out/k_train_phases.log
Y3_BN_EPILOGUE=0 timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/k_train_phases_noepi.log 2>&1; tail -1 gpurun_out/k_train_phases_noepi.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer', d['value'], d['legs_ms'])"
