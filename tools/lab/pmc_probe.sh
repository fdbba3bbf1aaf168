mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_probe -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-clocks > /tmp/pmc_probe.log 2>&1
cd $R
python - <<'PY' > gpurun_out/pmc_schema.txt 2>&1
import glob, sqlite3
dbs = glob.glob('/tmp/pmc_probe/**/*.db', recursive=True)
print(dbs)
db = sqlite3.connect(dbs[0])
for name, typ, sql in db.execute("select name, type, sql from sqlite_master"):
    print(typ, name)
    if typ in ('table','view') and sql: print('   ', sql[:1500].replace('\n',' '))
for t in ['counters_collection', 'kernels', 'top_kernels']:
    try:
        cur = db.execute(f'select * from {t} limit 3')
        print(t, [d[0] for d in cur.description])
        for r in cur: print('   ', r)
    except Exception as e: print(t, 'ERR', e)
PY
head -c 20000 gpurun_out/pmc_schema.txt
