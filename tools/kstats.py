"""Per-kernel summary of a rocprofv3 --kernel-trace database: python tools/kstats.py <dir> [title] [rows] -> markdown table on stdout."""
import glob
import re
import sqlite3
import sys

d = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else d
db = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tables else next(t for t in tables if "kernel" in t.lower())
rows = db.execute(f"select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from {kt} group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# {title}\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 32
for n, c, s, a, mn, mx in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)[:120]
    print(f"| {n} | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / tot:.1f} |")
