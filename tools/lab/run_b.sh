timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nms or val or detect or autoshape" 2>&1 | tail -3
for i in 1 2; do
for v in 0 1; do Y3_TUNE=nms_sort=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nms_sort=$v', d['value'], d['ms_per_step'], d['legs_ms'], d['sequential_images_per_sec_per_gpu'], d.get('sustained_images_per_sec'))"; done
done
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/nmsprof -o nms -- python $R/tools/lab/nms_probe.py 2>&1 | grep "ms per call"
cd $R
python tools/kstats.py gpurun_out/nmsprof "rocprofv3 --kernel-trace --stats : python tools/lab/nms_probe.py (23 calls of y3_nms, bs 32, 25200 x 85 fp16, conf 0.001 iou 0.6 multi_label)" 16 > gpurun_out/r06_nms_kernel_stats.md
cat gpurun_out/r06_nms_kernel_stats.md | cut -c1-160
rm -rf gpurun_out/nmsprof
