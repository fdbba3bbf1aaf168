# conv_v10.h A/B on the benchmark shapes + its parity group (one gpurun call)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "v10" > gpurun_out/v10_pytest.log 2>&1; tail -4 gpurun_out/v10_pytest.log
ARMS=${1:-"conv_v10=1;conv_v10=1,v10_half=1"}
timeout 300 python tools/conv_lab.py --rounds 5 --reps 20 --batch 32 --only "L6.cv2,L8.cv2,L10.cv2,L13" --arms "$ARMS" > gpurun_out/v10_lab_bs32.txt 2>&1; grep -v amdgpu gpurun_out/v10_lab_bs32.txt
timeout 300 python tools/conv_lab.py --rounds 5 --reps 10 --batch 64 --noact --only "L6.cv2,L8.cv2,L10.cv2" --arms "$ARMS" > gpurun_out/v10_lab_bs64.txt 2>&1; grep -v amdgpu gpurun_out/v10_lab_bs64.txt
for t in "" "v10_half=1"; do Y3_TUNE=$t python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUNE=$t', d['value'], d['legs_ms'], {k: v['ms'] for k, v in d['roofline']['whole_forward']['by_kernel'].items()})"; done
for t in "" "v10_half=1"; do Y3_TUNE=$t python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUNE=$t', d['value'], d['legs_ms'], {k: v['ms'] for k, v in d['roofline']['whole_forward']['by_kernel'].items()})"; done
