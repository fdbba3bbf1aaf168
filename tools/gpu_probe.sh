mkdir -p gpurun_out
for pr in 0 1 2; do
 for v in v3a v3b; do
  Y3_CONV=$v Y3_CONV_PROBE=$pr timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-layers > gpurun_out/probe_${v}_$pr.log 2>&1
 done
done
for l in "L6.0.cv2" "L8.0.cv2" "L10.0.cv2" "L13 " "L6.0.cv1" "L8.0.cv1"; do echo -n "$l: "; for v in v3a v3b; do for pr in 0 1 2; do grep -E "^ +$l" gpurun_out/probe_${v}_$pr.log | head -1 | awk -v t="$v/$pr" '{printf "%s %s ms | ", t, $2}'; done; done; echo; done
