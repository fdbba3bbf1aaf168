"""Per-kernel HBM-side traffic of a command from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), any kernels (tools/pmc_summary.py knows the inference symbols only):
python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <steps the command ran> > table.  Units per MI355X_MICROARCH.md: both counters in KiB, FETCH_SIZE
counts 128-byte requests as 64 B on gfx950 (read bytes = FETCH_SIZE x 1024 x 2)."""
import glob
import re
import sqlite3
import sys


def load(root, counter):
    out = {}
    for d in glob.glob(f"{root}/pmc_{counter}/**/*.db", recursive=True):
        db = sqlite3.connect(d)
        for k, n, s in db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
            a = out.setdefault(k, [0, 0.0])
            a[0] += n
            a[1] += s
        dur = {k: (n, t) for k, n, t in db.execute("select name, count(*), sum(duration) from kernels group by name")}
        return out, dur
    return out, {}


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    return (k[:96] + "...") if len(k) > 99 else k


root, steps = sys.argv[1], float(sys.argv[2])
rd, dur = load(root, "FETCH_SIZE")
wr, _ = load(root, "WRITE_SIZE")
rows = []
for k in set(rd) | set(wr):
    n = (rd.get(k) or wr.get(k))[0]
    r = rd.get(k, [0, 0.0])[1] * 1024 * 2
    w = wr.get(k, [0, 0.0])[1] * 1024
    t = dur.get(k, (0, 0))[1] / 1e6   # ms (of the FETCH_SIZE pass)
    rows.append((r + w, k, n, r, w, t))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# total {tot / steps / 1e9:.2f} GB per step over {steps:g} steps ({sum(r[3] for r in rows) / steps / 1e9:.2f} read + {sum(r[4] for r in rows) / steps / 1e9:.2f} written)")
print(f"{'launches/step':>13s} {'read MB/launch':>15s} {'written MB/launch':>18s} {'GB/step':>8s} {'share':>6s} {'ms/step':>8s} {'TB/s':>6s}  kernel")
for b, k, n, r, w, t in rows[:40]:
    print(f"{n / steps:13.1f} {r / n / 1e6:15.1f} {w / n / 1e6:18.1f} {b / steps / 1e9:8.2f} {b / tot:6.3f} {t / steps:8.3f} {(b / 1e12) / (t / 1e3) if t else 0:6.2f}  {short(k)}")
