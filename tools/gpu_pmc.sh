mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_$tag.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc_stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob, re, json
out = {}
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    dbs = glob.glob(d + '*.db')
    if not dbs: continue
    db = sqlite3.connect(dbs[0])
    if 'pmc_stats' in d:
        rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        lines = ["# rocprofv3 --kernel-trace --stats : python bench.py --steps 5 --warmup 2 --no-cpu-baseline (round 1 final, MI355X)", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
        for n,c,s,a,mn,mx in rows[:30]:
            n = re.sub(r"\(anonymous namespace\)::", "", n)[:110]
            lines.append(f"| {n} | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")
        open('gpurun_out/bench_kernel_stats_final.md','w').write("\n".join(lines)+"\n")
        continue
    rows = db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    for k, c, n, s, a in rows:
        k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", k)[:80]
        if 'conv_igemm' in k or 'decode' in k or 'nms' in k or 'nchw' in k:
            out.setdefault(k, {})[c] = {"dispatches": n, "sum": s, "avg": a}
json.dump(out, open('gpurun_out/pmc_summary_final.json', 'w'), indent=1)
for k, v in out.items():
    if 'conv_igemm' in k: print(k, {c: (x["dispatches"], round(x["avg"], 1)) for c, x in v.items()})
PY
head -12 gpurun_out/bench_kernel_stats_final.md
rm -rf gpurun_out/pmc_*/
