mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "wgrad or train_step" > gpurun_out/r3m_pytest.log 2>&1; echo "exit $?" >> gpurun_out/r3m_pytest.log
grep -a "passed\|failed\|exit\|Error" gpurun_out/r3m_pytest.log | tail -5
for i in 1 2; do timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'])"; done
timeout 300 python tools/train_layers.py > gpurun_out/r3m_train_layers.txt 2>&1; head -12 gpurun_out/r3m_train_layers.txt
