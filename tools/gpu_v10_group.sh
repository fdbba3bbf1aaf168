# conv_v10: tiles of a filter tile's pixel range taken round-robin by the blocks of one XCD (knob v10_group) -- tests, single-kernel A/B, the default line per setting, FETCH_SIZE per setting
mkdir -p gpurun_out
R=$PWD
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "v10 or baseline_shapes or train_launches or epilogue_bn or golden" > gpurun_out/v10_group_pytest.log 2>&1; echo "exit $?" >> gpurun_out/v10_group_pytest.log
grep -a "passed\|failed\|exit" gpurun_out/v10_group_pytest.log | tail -3
timeout 200 python tools/conv_lab.py --rounds 4 --reps 20 --batch 32 --only "L6.cv2,L8.cv2,L10.cv2,L13" --arms "v10_group=0;v10_group=1" > gpurun_out/v10_group_lab.txt 2>&1; tail -12 gpurun_out/v10_group_lab.txt
for rep in 1 2; do for v in 0 1; do
  Y3_TUNE=v10_group=$v timeout 120 python bench.py --no-cpu-baseline --no-train --no-clocks 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('v10_group=$v rep=$rep value', d['value'], 'seq', d['sequential_images_per_sec_per_gpu'], 'fwd', d['legs_ms']['forward+decode'], 'frac', r['frac'], {k:(v['ms'],v['tflops']) for k,v in r['forms'].items()})"
done; done | tee gpurun_out/v10_group_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/vg$v; mkdir -p $R/gpurun_out/vg$v
  Y3_TUNE=v10_group=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/vg$v/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/vg$v.log 2>&1
  Y3_TUNE=v10_group=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/vg$v/pmc_TCC -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks >> $R/gpurun_out/vg$v.log 2>&1
  (cd $R && python tools/pmc_summary.py gpurun_out/vg$v gpurun_out/pmc_vg$v.json > /dev/null 2>&1; python - <<P
import json
d=json.load(open("gpurun_out/pmc_vg$v.json"))
for k in ("conv_igemm_v10","stem_pair","bneck_pair"):
    r=d.get(k,{}); f=r.get("FETCH_SIZE",{})
    print("v10_group=$v", k, "read MB", round(f.get("avg",0)*2048/1e6,1), "l2 hit", round(r.get("l2_hit_rate",0),3), "us", f.get("pass_avg_us"))
P
  )
  rm -rf $R/gpurun_out/vg$v
done | tee -a $R/gpurun_out/v10_group_ab.txt
