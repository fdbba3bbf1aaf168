mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "train" > gpurun_out/n_pytest.log 2>&1; echo "exit $?" >> gpurun_out/n_pytest.log
tail -6 gpurun_out/n_pytest.log
for i in 1 2; do
timeout 300 python tools/train_bench.py --batch 64 --steps 4 --fused 2>&1 | tail -1 | cut -c100-330
Y3_WGRAD_STREAM=0 timeout 300 python tools/train_bench.py --batch 64 --steps 4 --fused 2>&1 | tail -1 | cut -c100-330
done
timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | cut -c1-160
