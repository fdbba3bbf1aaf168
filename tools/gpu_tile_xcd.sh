# XCD-grouped tile order of stem_pair / bneck_pair (knob tile_xcd): kernel tests, per-layer A/B (interleaved, twice), FETCH_SIZE pass per setting
mkdir -p gpurun_out
R=$PWD
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "stem or bneck or golden or fused" > gpurun_out/tile_xcd_pytest.log 2>&1; echo "exit $?" >> gpurun_out/tile_xcd_pytest.log
grep -a "passed\|failed\|exit" gpurun_out/tile_xcd_pytest.log | tail -3
for rep in 1 2; do for v in 0 1; do
  Y3_TUNE=tile_xcd=$v timeout 120 python bench.py --profile-layers --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-clocks > gpurun_out/layers_xcd${v}_$rep.txt 2>&1
  echo "tile_xcd=$v rep=$rep $(grep -a 'L0+L1\|^ *L2 \|L4.0 \|L4.1 \|total' gpurun_out/layers_xcd${v}_$rep.txt | awk '{printf "%s %s | ", $1, $2}') $(grep -a '^{"metric"' gpurun_out/layers_xcd${v}_$rep.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'fwd', d['legs_ms']['forward+decode'])")"
done; done | tee gpurun_out/tile_xcd_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/xcd$v; mkdir -p $R/gpurun_out/xcd$v
  Y3_TUNE=tile_xcd=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/xcd$v/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks > $R/gpurun_out/xcd$v.log 2>&1
  Y3_TUNE=tile_xcd=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/xcd$v/pmc_TCC -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-clocks >> $R/gpurun_out/xcd$v.log 2>&1
  (cd $R && python tools/pmc_summary.py gpurun_out/xcd$v gpurun_out/pmc_xcd$v.json > /dev/null 2>&1; python - <<P
import json
d=json.load(open("gpurun_out/pmc_xcd$v.json"))
for k in ("stem_pair","bneck_pair","conv_igemm_v10"):
    r=d.get(k,{}); print("tile_xcd=$v", k, "read MB", round(r.get("hbm_read_bytes_per_launch",0)/1e6,1), "l2 hit", r.get("l2_hit_rate"), "us", (r.get("FETCH_SIZE") or {}).get("pass_avg_us"))
P
  )
  rm -rf $R/gpurun_out/xcd$v
done | tee -a $R/gpurun_out/tile_xcd_ab.txt
