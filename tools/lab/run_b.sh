timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wgrad_patch or wgrad_benchmark_shapes" 2>&1 | tail -3
for r in 1 2; do
  for L in base new; do
    if [ $L = base ]; then export Y3_LIB=$PWD/yolov3_amd/lib/libyolov3_hip_base.so; else unset Y3_LIB; fi
    echo "== $L"; timeout 600 python tools/wgrad_lab.py --arms "wgrad_patch=1" --shapes L6cv2,L8cv2,L10cv2 --rounds 4 --reps 10 2>&1 | grep -v "amdgpu.ids\|^shape"
  done
done
unset Y3_LIB
Y3_LIB=$PWD/yolov3_amd/lib/libyolov3_hip_wpabl.so timeout 900 python tools/wgrad_patch_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_wgrad_patch_ablate.txt
