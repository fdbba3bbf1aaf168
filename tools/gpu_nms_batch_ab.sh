# nms_candidates_kernel with batched row loads: full GPU suite + smoke on the new library, then the same box's inference line with the previous NMS object
# (yolov3_amd/lib/libyolov3_hip_prev.so = HEAD's detect_nms.hip linked with the same other objects, via Y3_LIB) and the new one, interleaved
# prepare here: git show <previous commit>:yolov3_amd/csrc/detect_nms.hip compiled with build()'s flags (-I yolov3_amd/csrc -I include) to an object, linked with the
# other objects of yolov3_amd/build/ into yolov3_amd/lib/libyolov3_hip_prev.so (git-ignored, travels with the snapshot); delete it afterwards
mkdir -p gpurun_out
out=gpurun_out/r05_nms_row_batch_ab.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r05_pytest_gpu_full.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/r05_pytest_gpu_full.log
echo "# $(grep -a 'passed\|failed' gpurun_out/r05_pytest_gpu_full.log | tail -1)" > $out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | sed 's/^/# /' >> $out
echo "# same box, interleaved: python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks" >> $out
for r in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export Y3_LIB=$PWD/yolov3_amd/lib/libyolov3_hip_prev.so; else unset Y3_LIB; fi
    timeout 200 python bench.py --steps 30 --warmup 5 --no-train --no-cpu-baseline --no-clocks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r $lib: value', d['value'], 'img/s  sequential', d['sequential_images_per_sec_per_gpu'], ' legs', d['legs_ms'], ' synthetic-tensor form', d['synthetic_nms_tensor']['images_per_sec'])" >> $out
  done
done
unset Y3_LIB
cat $out
