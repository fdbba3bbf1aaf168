"""Per-dispatch dump of a rocprofv3 --kernel-trace database (last step only): python tools/ktrace.py <dir> [name-substring ...]
Prints dispatch order, kernel (shortened), grid, workgroup, duration us -- used to map wgrad / dgrad / BN launches to layers."""
import glob
import re
import sqlite3
import sys

d = sys.argv[1]
pats = sys.argv[2:]
db = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x")]
gy = [c for c in cols if c.lower() in ("grid_y", "grid_size_y")]
wx = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x")]
sel = "name, start, end" + "".join(", " + c[0] for c in (gx, gy, wx) if c)
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
# keep the last third (train_bench runs warm-up + steps): find the last nchw_to_nhwc launch
last = max((i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[0]), default=0)
for i, r in enumerate(rows[last:]):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r[0])
    if pats and not any(p in n for p in pats):
        continue
    print(i, n[:70].ljust(70), " ".join(str(v) for v in r[3:]), f"{(r[2] - r[1]) / 1e3:9.2f}")
