mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "letterbox or autoshape or stem or end_to_end or model_half" > gpurun_out/i_pytest.log 2>&1; echo "exit $?" >> gpurun_out/i_pytest.log
tail -25 gpurun_out/i_pytest.log
