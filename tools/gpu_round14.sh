mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/tools/train_bench.py --batch 64 --steps 2 --fused > $R/gpurun_out/train_prof64.log 2>&1; echo "exit $?" >> $R/gpurun_out/train_prof64.log
cd $R
python - <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob('gpurun_out/prof_train/*.db')[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = ["# rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 2 --fused (3 steps incl. warm-up), yolov3 640x640 autocast fp16", f"total kernel ms {tot/1e6:.1f} (3 steps)", "| ms | calls | avg us | kernel |", "|---|---|---|---|"]
for n,c,s,a in rows[:24]:
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:90]
    out.append(f"| {s/1e6:.3f} | {c} | {a/1e3:.2f} | {n} |")
open('gpurun_out/train64_kernel_stats.md','w').write("\n".join(out)+"\n")
print("\n".join(out))
PY
grep -E "^\{" gpurun_out/train_prof64.log
rm -rf gpurun_out/prof_train
