mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "wgrad or train or sgd" > gpurun_out/e_pytest.log 2>&1; echo "exit $?" >> gpurun_out/e_pytest.log
tail -15 gpurun_out/e_pytest.log
timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/e_train_phases.log 2>&1; tail -1 gpurun_out/e_train_phases.log
Y3_WGRAD=dma timeout 300 python tools/train_bench.py --batch 64 --steps 3 --fused > gpurun_out/e_train_phases_dma.log 2>&1; tail -1 gpurun_out/e_train_phases_dma.log
timeout 300 python bench.py --mode train --batch 64 --steps 5 --warmup 2 > gpurun_out/e_train.log 2>&1; tail -1 gpurun_out/e_train.log
