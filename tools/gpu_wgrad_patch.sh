# round 6: the padded-position filter-gradient kernel (csrc/wgrad_patch.h) -- parity tests, the ablation table (lab library: python tools/wgrad_patch_ablate.py --build
# first; delete yolov3_amd/lib/libyolov3_hip_wpabl.so afterwards), per-launch A/B against wgrad_big on the benchmark's shapes
#   gpurun -- bash tools/gpu_wgrad_patch.sh
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wgrad_patch or wgrad_benchmark_shapes or wgrad_256_tile" 2>&1 | tail -3
if [ -f yolov3_amd/lib/libyolov3_hip_wpabl.so ]; then
  Y3_LIB=$PWD/yolov3_amd/lib/libyolov3_hip_wpabl.so timeout 900 python tools/wgrad_patch_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_wgrad_patch_ablate.txt
fi
timeout 600 python tools/wgrad_lab.py --arms "wgrad_patch=0;wgrad_patch=1" --shapes L6cv2,L8cv2,L10cv2 --rounds 5 --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_wgrad_patch_lab.txt
