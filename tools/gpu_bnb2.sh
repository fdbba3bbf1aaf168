mkdir -p gpurun_out
timeout 120 tools/lab/bn_lab > gpurun_out/bn_lab.txt 2>&1; echo "lab exit $?"
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "bn_backward or rccl_one_rank or two_outstanding" > gpurun_out/bnb_pytest.log 2>&1; echo "exit $?" >> gpurun_out/bnb_pytest.log
tail -12 gpurun_out/bnb_pytest.log
