mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/tools/train_bench.py --batch 16 --steps 3 > $R/gpurun_out/train_prof.log 2>&1; echo "exit $?" >> $R/gpurun_out/train_prof.log
cd $R
python - <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob('gpurun_out/prof_train/*.db')[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel ms", tot/1e6)
for n,c,s,a in rows[:25]:
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:90]
    print(f"{s/1e6:9.3f} ms {c:6d} calls {a/1e3:9.2f} us  {n}")
PY
grep -E "^\{" gpurun_out/train_prof.log
rm -rf gpurun_out/prof_train
