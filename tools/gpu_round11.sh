mkdir -p gpurun_out
LOG=gpurun_out/pytest_gpu11.log; : > $LOG
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "fused_sgd or train_step" >> $LOG 2>&1; echo "exit $?" >> $LOG
grep -E "passed|failed|exit|^E  |FAILED" $LOG | cut -c1-300 | tail -20
for b in 16 64; do timeout 900 python tools/train_bench.py --batch $b --steps 3 --fused 2>&1 | grep -E "^\{|Error|error" ; done
timeout 900 python bench.py --mode train --batch 64 --steps 5 --warmup 2 2>&1 | grep -E "^\{|Error|error"
timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | grep -E "^\{|Error|error" | cut -c1-1800
