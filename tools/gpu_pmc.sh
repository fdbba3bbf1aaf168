mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_$tag.log
done
cd $R
python - <<'PY'
import sqlite3, glob, re, json
out = {}
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    dbs = glob.glob(d + '*.db')
    if not dbs: continue
    db = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        rows = db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    except Exception as e:
        print(d, "schema:", tabs[-12:], e); continue
    for k, c, n, s, a in rows:
        k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", k)[:70]
        if 'conv_igemm' in k or 'decode' in k or 'nms' in k or 'nchw' in k:
            out.setdefault(k, {})[c] = {"dispatches": n, "sum": s, "avg": a}
json.dump(out, open('gpurun_out/pmc_summary.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {c: (x["dispatches"], round(x["avg"], 1)) for c, x in v.items()})
PY
rm -rf gpurun_out/pmc_*/
