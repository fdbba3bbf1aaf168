# XCD-grouped tiles in the training stem kernels (stem_conv with statistics, stem_bn_bwd_wgrad): tests, then the per-layer table of the batch-64 train step per knob value
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "stem or bneck or golden or train_step or train_forward or autocast" > gpurun_out/tile_xcd_pytest2.log 2>&1; echo "exit $?" >> gpurun_out/tile_xcd_pytest2.log
grep -a "passed\|failed\|exit" gpurun_out/tile_xcd_pytest2.log | tail -3
for rep in 1 2; do for v in 0 1; do
  Y3_TUNE=tile_xcd=$v timeout 200 python tools/train_layers.py --top 4 > gpurun_out/train_layers_xcd${v}_$rep.txt 2>&1
  echo "tile_xcd=$v rep=$rep $(grep -a '^units\|^L0 \|^L1 ' gpurun_out/train_layers_xcd${v}_$rep.txt | tr '\n' ';')"
done; done | tee gpurun_out/tile_xcd_train_ab.txt
