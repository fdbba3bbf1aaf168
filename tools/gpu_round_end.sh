# full GPU test suite + smoke, then the round's measurement set
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_full.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_full.log
tail -6 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/gpu_final.sh
